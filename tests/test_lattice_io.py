"""Host-side readers of the lattice path (no GPU): HCLG.fst in OpenFst binary form through the C ABI, Kaldi's
text transition model, the final.occs priors.  The byte fixtures are assembled here from the published formats
(SURVEY.md Appendix C); Kaldi / OpenFst are not available to write them."""
import ctypes as C
import struct

import numpy as np
import pytest

from pykaldi2_amd import _lib, lattice, se, synth


def _write_fst(path, g, kind):
    S = g["num_states"]
    order = np.argsort(g["src"], kind="stable")
    src, dst, il, w = g["src"][order], g["dst"][order], g["ilabel"][order], g["weight"][order]
    counts = np.bincount(src, minlength=S)
    s = lambda x: struct.pack("<i", len(x)) + x
    flags = 4 if kind == "const_aligned" else 0
    head = struct.pack("<i", 2125659606) + s(b"const" if kind.startswith("const") else b"vector") + s(b"standard")
    head += struct.pack("<iiQqqq", 2, flags, 0, g["start"], S, len(src))
    if kind == "vector":
        body, a = b"", 0
        for st in range(S):
            body += struct.pack("<fq", float(g["final"][st]), int(counts[st]))
            for _ in range(counts[st]):
                body += struct.pack("<iifi", int(il[a]), 7, float(w[a]), int(dst[a])); a += 1
    else:
        pad = lambda b, pos: b + b"\0" * ((16 - (pos + len(b)) % 16) % 16) if kind == "const_aligned" else b
        head = pad(head, 0)
        pos, states = 0, b""
        for st in range(S):
            states += struct.pack("<fIIII", float(g["final"][st]), pos, int(counts[st]), 0, 0)
            pos += int(counts[st])
        states = pad(states, len(head))
        body = states + b"".join(struct.pack("<iifi", int(il[a]), 7, float(w[a]), int(dst[a])) for a in range(len(src)))
    with open(path, "wb") as f:
        f.write(head + body)


@pytest.mark.parametrize("kind", ["vector", "const", "const_aligned"])
def test_decode_graph_from_openfst(tmp_path, kind):
    g = synth.decoding_graph_arcs(12, 30, seed=3)
    path = str(tmp_path / "HCLG.fst")
    _write_fst(path, g, kind)
    graph = lattice.DecodeGraph(path)
    assert (graph.num_states, graph.num_arcs, graph.max_ilabel) == (g["num_states"], len(g["src"]), int(g["ilabel"].max()))


def test_decode_graph_rejects_bad_files(tmp_path):
    p = tmp_path / "bad.fst"
    p.write_bytes(b"not an fst at all")
    with pytest.raises(_lib.Pk2Error, match="bad magic"):
        lattice.DecodeGraph(str(p))
    g = synth.decoding_graph_arcs(5, 12, seed=0)
    path = str(tmp_path / "trunc.fst")
    _write_fst(path, g, "vector")
    raw = open(path, "rb").read()
    open(path, "wb").write(raw[:len(raw) // 2])
    with pytest.raises(_lib.Pk2Error, match="truncated"):
        lattice.DecodeGraph(path)
    with pytest.raises(_lib.Pk2Error):
        lattice.DecodeGraph(dict(num_states=2, start=0, src=[0], dst=[5], ilabel=[1], weight=[0.0], final=[0.0, 0.0]))


TRANS_MODEL_TXT = """<TransitionModel>
<Topology>
<TopologyEntry>
<ForPhones>
2 3
</ForPhones>
<State> 0 <PdfClass> 0 <Transition> 0 0.75 <Transition> 1 0.25 </State>
<State> 1 <PdfClass> 1 <Transition> 1 0.75 <Transition> 2 0.25 </State>
<State> 2 </State>
</TopologyEntry>
<TopologyEntry>
<ForPhones>
1
</ForPhones>
<State> 0 <PdfClass> 0 <Transition> 0 0.5 <Transition> 1 0.25 <Transition> 2 0.25 </State>
<State> 1 <PdfClass> 1 <Transition> 1 0.5 <Transition> 2 0.5 </State>
<State> 2 </State>
</TopologyEntry>
</Topology>
<Triples> 6
1 0 0
1 1 1
2 0 2
2 1 3
3 0 4
3 1 2
</Triples>
<LogProbs>
 [ 0 -0.69 -1.38 -1.38 -0.69 -0.69 -0.28 -1.38 -0.28 -1.38 -0.28 -1.38 -0.28 -1.38 ]
</LogProbs>
</TransitionModel>
"""


def test_transition_model_text_reader(tmp_path):
    p = tmp_path / "final.mdl.txt"
    p.write_text(TRANS_MODEL_TXT)
    tm = lattice.TransitionModel.read(str(p))
    # phone 1: state 0 has 3 transitions, state 1 has 2; phones 2 and 3: 2 transitions per state
    assert tm.num_transition_ids() == 3 + 2 + 2 + 2 + 2 + 2
    assert tm.tid2pdf.tolist() == [-1, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 2, 2]
    assert tm.tid2phone.tolist() == [0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3]
    assert tm.num_pdfs() == 5 and tm.transition_id_to_pdf(12) == 2 and tm.transition_id_to_phone(12) == 3
    chain_like = TRANS_MODEL_TXT.replace("<PdfClass> 0", "<ForwardPdfClass> 0 <SelfLoopPdfClass> 1").replace(
        "<PdfClass> 1", "<ForwardPdfClass> 1 <SelfLoopPdfClass> 1")
    chain_like = chain_like.replace("<Triples> 6\n1 0 0\n1 1 1\n2 0 2\n2 1 3\n3 0 4\n3 1 2", "<Tuples> 6\n1 0 0 5\n1 1 1 6\n2 0 2 7\n2 1 3 8\n3 0 4 9\n3 1 2 10")
    chain_like = chain_like.replace("</Triples>", "</Tuples>")
    p.write_text(chain_like)
    tm2 = lattice.TransitionModel.read(str(p))
    # the self-loop (transition back to the same HMM state) carries the self-loop pdf
    assert tm2.tid2pdf.tolist() == [-1, 5, 0, 0, 6, 1, 7, 2, 8, 3, 9, 4, 10, 2]
    # the same model in Kaldi's binary form (assembled from the published encoding), GMM payload after it ignored
    i32 = lambda v: b"\x04" + struct.pack("<i", v)
    f32 = lambda v: b"\x04" + struct.pack("<f", v)
    ivec = lambda v: b"\x04" + struct.pack("<i", len(v)) + struct.pack("<%di" % len(v), *v)

    def entry(states):
        out = i32(len(states))
        for pdf_class, trans in states:
            out += i32(pdf_class) + i32(len(trans)) + b"".join(i32(d) + f32(p) for d, p in trans)
        return out
    e23 = [(0, [(0, 0.75), (1, 0.25)]), (1, [(1, 0.75), (2, 0.25)]), (-1, [])]
    e1 = [(0, [(0, 0.5), (1, 0.25), (2, 0.25)]), (1, [(1, 0.5), (2, 0.5)]), (-1, [])]
    raw = (b"\0B<TransitionModel> <Topology> " + ivec([1, 2, 3]) + ivec([-1, 1, 0, 0]) + i32(2) + entry(e23) + entry(e1) +
           b"</Topology> <Triples> " + i32(6) +
           b"".join(i32(a) + i32(b) + i32(c) for a, b, c in [(1, 0, 0), (1, 1, 1), (2, 0, 2), (2, 1, 3), (3, 0, 4), (3, 1, 2)]) +
           b"</Triples> <LogProbs> FV " + i32(2) + struct.pack("<2f", 0.0, -0.69) + b"</LogProbs> </TransitionModel> <DIMENSION> junk")
    (tmp_path / "final.mdl").write_bytes(raw)
    tmb = lattice.TransitionModel.read(str(tmp_path / "final.mdl"))
    assert tmb.tid2pdf.tolist() == tm.tid2pdf.tolist() and tmb.tid2phone.tolist() == tm.tid2phone.tolist()


def test_prior_readers(tmp_path):
    v = np.array([3.0, 1.0, 4.0, 2.0])
    (tmp_path / "t.occs").write_text(" [ 3 1 4 2 ]\n")
    assert np.allclose(se.read_kaldi_vector(str(tmp_path / "t.occs")), v)
    (tmp_path / "fv.occs").write_bytes(b"\0BFV " + b"\x04" + struct.pack("<i", 4) + v.astype("<f4").tobytes())
    assert np.allclose(se.read_kaldi_vector(str(tmp_path / "fv.occs")), v)
    (tmp_path / "dm.occs").write_bytes(b"\0BDM " + b"\x04" + struct.pack("<i", 1) + b"\x04" + struct.pack("<i", 4) + v.astype("<f8").tobytes())
    assert np.allclose(se.read_kaldi_vector(str(tmp_path / "dm.occs")), v)
    lp = se.log_prior_from_counts(v)
    assert np.allclose(lp.numpy(), np.log(v / v.sum()), atol=1e-6)


def test_synthetic_tid_alignment_is_consistent():
    rng = np.random.default_rng(0)
    tm = synth.transition_model_arrays(30)
    ali = synth.tid_alignment(rng, 200, 30)
    assert ali.shape == (200,) and ali.min() >= 1 and ali.max() <= 60
    pdf = tm["tid2pdf"][ali]
    assert np.array_equal(pdf, (ali - 1) // 2)
    # forward transitions (even ids) end an HMM state: the next frame starts the next state of the phone or a new phone
    for t in range(199):
        if ali[t] % 2 == 1:
            assert ali[t + 1] in (ali[t], ali[t] + 1)


def test_kaldi_matrix_ark_roundtrip(tmp_path):
    from pykaldi2_amd import kaldi_io
    rng = np.random.default_rng(0)
    mats = {"utt-a": rng.standard_normal((5, 7)).astype(np.float32), "utt-b": rng.standard_normal((1, 3)).astype(np.float32)}
    path = str(tmp_path / "ll.ark")
    with kaldi_io.MatrixWriter("ark:" + path) as w:
        for k, m in mats.items():
            w[k] = m
    raw = open(path, "rb").read()
    assert raw.startswith(b"utt-a \0BFM \x04\x05\x00\x00\x00\x04\x07\x00\x00\x00")      # Kaldi's binary matrix header
    got = dict(kaldi_io.read_matrix_ark(path))
    assert list(got) == ["utt-a", "utt-b"] and all(np.array_equal(got[k], mats[k]) for k in mats)
    (tmp_path / "t.ark").write_text("utt-c  [\n  1 2 3\n  4 5 6 ]\n")
    (k, m), = kaldi_io.read_matrix_ark(str(tmp_path / "t.ark"))
    assert k == "utt-c" and np.array_equal(m, np.array([[1, 2, 3], [4, 5, 6]], np.float32))


def test_mvn_transform_pickle_of_the_reference_class(tmp_path):
    """-transform files are pickles of reader.preprocess.GlobalMeanVarianceNormalization; they load without that
    module being importable."""
    import pickle
    import sys
    import types
    from pykaldi2_amd import fbank
    mod = types.ModuleType("reader.preprocess")
    pkg = types.ModuleType("reader")

    class GlobalMeanVarianceNormalization:
        pass
    GlobalMeanVarianceNormalization.__module__ = "reader.preprocess"
    GlobalMeanVarianceNormalization.__qualname__ = "GlobalMeanVarianceNormalization"
    mod.GlobalMeanVarianceNormalization = GlobalMeanVarianceNormalization
    sys.modules["reader"], sys.modules["reader.preprocess"] = pkg, mod
    try:
        o = GlobalMeanVarianceNormalization()
        o.mean_vec = np.arange(80, dtype=np.float32).reshape(1, 80)
        o.std_vec = np.full((1, 80), 2.0, np.float32)
        o.mean_norm, o.var_norm = True, True
        with open(tmp_path / "mvn.pkl", "wb") as f:
            pickle.dump(o, f)
    finally:
        del sys.modules["reader"], sys.modules["reader.preprocess"]
    t = fbank.GlobalMeanVarianceNormalization.load(str(tmp_path / "mvn.pkl"))
    assert np.array_equal(t.mean_vec, o.mean_vec) and np.array_equal(t.std_vec, o.std_vec)


def test_mvn_transform_save_load_round_trip(tmp_path):
    from pykaldi2_amd import fbank
    rng = np.random.default_rng(3)
    x = rng.normal(5, 2, size=(1000, 80))
    t = fbank.GlobalMeanVarianceNormalization.from_stats(x.sum(0), (x * x).sum(0), 1000)
    assert np.abs(t.mean_vec - x.mean(0)).max() < 1e-3 and np.abs(t.std_vec - x.std(0)).max() < 1e-2
    t.save(str(tmp_path / "transform.pkl"))
    back = fbank.GlobalMeanVarianceNormalization.load(str(tmp_path / "transform.pkl"))
    assert back.mean_vec.shape == (1, 80) and np.array_equal(back.mean_vec, t.mean_vec) and np.array_equal(back.std_vec, t.std_vec)
    assert np.all(t.std_vec >= 1e-2)


def test_compact_lattice_archives_and_link_words(tmp_path):
    """kaldi_io.CompactLatticeWriter (reference bin/latgen.py:156-181 writes `decoder_out["lattice"]` through PyKaldi's
    CompactLatticeWriter): binary OpenFst "vector" / "compactlattice44" entries and the text form, start state renumbered
    to 0; and the word lookup of lattice links on the host side of the decode graph."""
    import numpy as np
    from pykaldi2_amd import kaldi_io, lattice, synth
    lat = dict(num_states=4, start=2, src=np.array([2, 2, 0, 1, 0]), dst=np.array([0, 1, 3, 3, 1]),
               word=np.array([7, 0, 0, 9, 0], np.int32), graph=np.array([0.5, 1.5, 0.0, 2.25, 0.125], np.float32),
               acoustic=np.array([-3.0, -4.0, -5.5, 0.0, -1.0], np.float32), tid=np.array([11, 12, 13, 0, 14], np.int32),
               final=np.array([np.inf, np.inf, np.inf, 0.75], np.float32))
    with kaldi_io.CompactLatticeWriter("ark:" + str(tmp_path / "l.ark")) as w:
        w["utt-a"] = lat
        w["utt-b"] = lat
    got = list(kaldi_io.read_compact_lattice_ark(str(tmp_path / "l.ark")))
    assert [k for k, _ in got] == ["utt-a", "utt-b"]
    g = got[0][1]
    assert g["start"] == 0 and g["num_states"] == 4
    ren = {2: 0, 0: 2, 1: 1, 3: 3}                                   # start state swapped into position 0
    want = sorted((ren[s], ren[d], int(wd), int(wd), (float(gr), float(ac), [int(t)] if t else []))
                  for s, d, wd, gr, ac, t in zip(lat["src"], lat["dst"], lat["word"], lat["graph"], lat["acoustic"], lat["tid"]))
    assert sorted(g["arcs"]) == want
    assert g["finals"][3] == (0.75, 0.0, []) and all(np.isinf(g["finals"][s][0]) for s in (0, 1, 2))
    raw = (tmp_path / "l.ark").read_bytes()
    assert raw.startswith(b"utt-a \0B" + b"\xd6\xfd\xb2\x7e") and b"compactlattice44" in raw[:64]
    with kaldi_io.CompactLatticeWriter("ark,t:" + str(tmp_path / "l.txt")) as w:
        w["utt-a"] = lat
    lines = (tmp_path / "l.txt").read_text().split("\n")
    assert lines[0] == "utt-a" and "0\t2\t7\t0.5,-3.0,11" in lines and "1\t3\t9\t2.25,0.0," in lines and "3\t0.75,0," in lines
    assert lines[-2:] == ["", ""]
    # words of lattice links from the decode graph (no GPU involved)
    gr = synth.decoding_graph_arcs(30, 60, seed=2)
    dg = lattice.DecodeGraph(gr)
    idx = np.flatnonzero(gr["olabel"] > 0)[:5].tolist() + [3, 4, 5]
    words = dg.link_words(gr["src"][idx], gr["dst"][idx], gr["ilabel"][idx], gr["weight"][idx])
    assert words.tolist() == gr["olabel"][idx].tolist()
    assert dg.link_words([0], [0], [5], [0.0]).tolist() == [-1]      # no such arc
