"""Writes a tiny training recipe to disk in the formats the reference's command lines read:
OpenFst `vector` FSTs (den.fst, HCLG.fst), Kaldi's tree, text transition models (0.trans_mdl, final.mdl), final.occs,
words.txt, phones/silence.csl, a zip of 16 kHz PCM wavs, label text files, and the two YAML files.  Test infrastructure:
the bytes are assembled from the published formats ([upstream knowledge]: no Kaldi / OpenFst here to write them), the
content comes from pykaldi2_amd.synth -- the same arrays the `-synthetic` paths are fed."""
import io
import os
import struct
import wave
import zipfile

import numpy as np
import yaml

from pykaldi2_amd import synth


def write_fst_vector(path, num_states, start, src, dst, ilabel, olabel, weight, final):
    """OpenFst binary StdVectorFst: header, then per state final weight, arc count and {ilabel, olabel, weight, next}."""
    src = np.asarray(src)
    order = np.argsort(src, kind="stable")
    src, dst, il, ol, w = src[order], np.asarray(dst)[order], np.asarray(ilabel)[order], np.asarray(olabel)[order], np.asarray(weight)[order]
    counts = np.bincount(src, minlength=num_states)
    s = lambda x: struct.pack("<i", len(x)) + x
    out = [struct.pack("<i", 2125659606), s(b"vector"), s(b"standard"), struct.pack("<iiQqqq", 2, 0, 0, int(start), int(num_states), len(src))]
    a = 0
    for st in range(num_states):
        out.append(struct.pack("<fq", float(final[st]), int(counts[st])))
        for _ in range(counts[st]):
            out.append(struct.pack("<iifi", int(il[a]), int(ol[a]), float(w[a]), int(dst[a])))
            a += 1
    with open(path, "wb") as f:
        f.write(b"".join(out))


def write_transition_model_text(path, phone2entry, entries, tuples):
    """Kaldi's text transition model (`copy-transition-model --binary=false`): topology entries shared by phones,
    <Triples> (one pdf per HMM state) or <Tuples> (forward and self-loop pdf), dummy log-probabilities."""
    by_entry = {}
    for ph, e in sorted(phone2entry.items()):
        by_entry.setdefault(e, []).append(ph)
    two = any(f != l for st in entries for f, l, _ in st)
    lines = ["<TransitionModel>", "<Topology>"]
    for e, phones in sorted(by_entry.items()):
        lines += ["<TopologyEntry>", "<ForPhones>", " ".join(str(p) for p in phones), "</ForPhones>"]
        for hs, (f, l, dsts) in enumerate(entries[e]):
            if not dsts:
                lines.append("<State> %d </State>" % hs)
                continue
            cls = ("<ForwardPdfClass> %d <SelfLoopPdfClass> %d" % (f, l)) if two else ("<PdfClass> %d" % f)
            tr = " ".join("<Transition> %d %.4f" % (d, 1.0 / len(dsts)) for d in dsts)
            lines.append("<State> %d %s %s </State>" % (hs, cls, tr))
        lines.append("</TopologyEntry>")
    lines.append("</Topology>")
    tag = "Tuples" if two else "Triples"
    lines.append("<%s> %d" % (tag, len(tuples)))
    n_tids = 0
    for ph, hs, fwd, loop in tuples:
        lines.append(("%d %d %d %d" % (ph, hs, fwd, loop)) if two else ("%d %d %d" % (ph, hs, fwd)))
        n_tids += len(entries[phone2entry[int(ph)]][int(hs)][2])
    lines += ["</%s>" % tag, "<LogProbs>", " [ " + " ".join(["0"] + ["-0.69"] * n_tids) + " ]", "</LogProbs>", "</TransitionModel>"]
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def wav_bytes(x):
    pcm = np.clip(np.round(np.asarray(x) * 32768.0), -32768, 32767).astype("<i2")
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(pcm.tobytes())
    return buf.getvalue()


def write_wav_zip(path, wavs):
    with zipfile.ZipFile(path, "w") as z:
        for utt, x in wavs.items():
            z.writestr("corpus/%s.wav" % utt, wav_bytes(x))


def write_labels(path, labels):
    with open(path, "w") as f:
        for utt, ids in labels.items():
            f.write(utt + " " + " ".join(str(int(v)) for v in ids) + "\n")


def model_yaml(path, label_size, hidden=64, layers=2, decoder=None):
    cfg = dict(data_config=dict(frame_len=400, frame_shift=160, seg_len=80, seg_shift=80, sequence_mode=True,
                                load_label=True, use_cmn=True, simulation_prob=0),
               model_config=dict(feat_dim=80, hidden_size=hidden, dropout=0.0, num_layers=layers, label_size=label_size))
    if decoder:
        cfg["decoder_config"] = decoder
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    return str(path)


def chain_recipe(root, P=120, den_states=300, den_arcs=5000, durations=(1.6, 2.3), seed=0):
    """<root>/chain: den.fst, 0.trans_mdl, tree; <root>/ali: final.mdl, tree; <root>/lang: L.fst, phones/disambig.int;
    <root>/data: train.zip, tid.txt (transition-id alignments), data.yaml.  Returns the arrays behind the files."""
    rng = np.random.default_rng(seed)
    for d in ("chain", "ali", "lang/phones", "data"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    g = synth.den_graph_arcs(den_states, den_arcs, P, seed=seed, loop_pdf_differs=True)
    # den.fst: ilabel = olabel = pdf + 1, weight = -log(prob), every state final (bin/train_chain.py:167,202)
    write_fst_vector(os.path.join(root, "chain", "den.fst"), g["num_states"], 0, g["src"], g["dst"], g["pdf"] + 1, g["pdf"] + 1,
                     -np.log(g["prob"]), np.zeros(g["num_states"]))
    tree, tm = synth.chain_model(P, seed=seed)
    for d in ("chain", "ali"):
        tree.write(os.path.join(root, d, "tree"))
    tuples = [tuple(int(v) for v in t) for t in tm.tuples]
    write_transition_model_text(os.path.join(root, "chain", "0.trans_mdl"), tm.phone2entry, tm.entries, tuples)
    write_transition_model_text(os.path.join(root, "ali", "final.mdl"), tm.phone2entry, tm.entries, tuples)
    open(os.path.join(root, "lang", "L.fst"), "wb").write(b"")              # (the aligner's decoding side is never used)
    open(os.path.join(root, "lang", "phones", "disambig.int"), "w").write("")
    wavs, alis = {}, {}
    for n, dur in enumerate(durations):
        utt = "spk%d-utt%d" % (n, n)
        w = synth.waveform(rng, dur)
        wavs[utt] = w
        alis[utt] = synth.phone_tid_alignment(rng, synth.num_fbank_frames(w.shape[0]), tm)[0]
    write_wav_zip(os.path.join(root, "data", "train.zip"), wavs)
    write_labels(os.path.join(root, "data", "tid.txt"), alis)
    with open(os.path.join(root, "data", "data.yaml"), "w") as f:
        yaml.safe_dump(dict(clean_source=dict(train=dict(type="Librispeech", wav=os.path.join(root, "data", "train.zip"),
                                                         label=os.path.join(root, "data", "tid.txt")))), f)
    return dict(den=g, tree=tree, trans_model=tm, wavs=wavs, alis=alis, P=P)


BAKIS3 = [(0, 0, [0, 1]), (1, 1, [1, 2]), (2, 2, [2, 3]), (-1, -1, [])]


def lattice_recipe(root, P=90, words=40, durations=(1.2, 1.7), seed=0):
    """<root>/graph: HCLG.fst, words.txt, phones/silence.csl; <root>/final.mdl (text), <root>/final.occs;
    <root>/data: train.zip, pdf.txt, tid.txt, data.yaml -- the synthetic word-loop graph and 3-state model of
    pykaldi2_amd.synth (what `-synthetic` feeds train_se.py / latgen.py) as files."""
    rng = np.random.default_rng(seed)
    for d in ("graph/phones", "data"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    tm = synth.transition_model_arrays(P)
    hclg = synth.decoding_graph_arcs(words, P, seed=seed)
    write_fst_vector(os.path.join(root, "graph", "HCLG.fst"), hclg["num_states"], hclg["start"], hclg["src"], hclg["dst"],
                     hclg["ilabel"], hclg["olabel"], hclg["weight"], hclg["final"])
    with open(os.path.join(root, "graph", "words.txt"), "w") as f:
        f.write("<eps> 0\n" + "".join("w%d %d\n" % (i, i) for i in range(1, words + 1)))
    open(os.path.join(root, "graph", "phones", "silence.csl"), "w").write(":".join(str(p) for p in tm["silence_phones"]) + "\n")
    nph = tm["num_phones"]
    write_transition_model_text(os.path.join(root, "final.mdl"), {ph: 0 for ph in range(1, nph + 1)}, [BAKIS3],
                                [(ph, hs, 3 * (ph - 1) + hs, 3 * (ph - 1) + hs) for ph in range(1, nph + 1) for hs in range(3)])
    counts = rng.integers(1, 50, size=P).astype(np.float64)
    open(os.path.join(root, "final.occs"), "w").write(" [ " + " ".join("%g" % c for c in counts) + " ]\n")
    wavs, pdfs, tids = {}, {}, {}
    for n, dur in enumerate(durations):
        utt = "spk%d-utt%d" % (n, n)
        w = synth.waveform(rng, dur)
        T = synth.num_fbank_frames(w.shape[0])
        ali = synth.tid_alignment(rng, T, P)
        wavs[utt], tids[utt], pdfs[utt] = w, ali, tm["tid2pdf"][ali]
    write_wav_zip(os.path.join(root, "data", "train.zip"), wavs)
    write_labels(os.path.join(root, "data", "pdf.txt"), pdfs)
    write_labels(os.path.join(root, "data", "tid.txt"), tids)
    with open(os.path.join(root, "data", "data.yaml"), "w") as f:
        yaml.safe_dump(dict(clean_source=dict(train=dict(type="Librispeech", wav=os.path.join(root, "data", "train.zip"),
                                                         label=os.path.join(root, "data", "pdf.txt"),
                                                         aux_label=os.path.join(root, "data", "tid.txt")))), f)
    return dict(tm=tm, hclg=hclg, counts=counts, wavs=wavs, pdfs=pdfs, tids=tids, P=P, words=words)
