"""Generates tests/golden/*.npz by IMPORTING the reference (jzlianglu/pykaldi2 at /root/reference).

Run in the build container only (the reference does not exist on the GPU box):
    python tools/gen_golden.py
The fixtures hold data only: seeded inputs and the outputs the reference's own code produced for
them.  Fix-ups applied here, never to the reference:
  * np.int = int            simulation/freq_analysis.py:66 and data/dataloader.py:109 use the removed alias
  * LSTMAM.forward          models/lstm.py:59 is a NameError -> call m.output_layer(m.lstm(x)[0])
  * reader/, data/          loaded by file path, never `import reader` (reader/reader.py:28-40 shells out)
  * _logfbank_extractor     lives in a class whose module imports reader; its 12 lines
                            (data/sr_dataset.py:279-296) are re-stated around the imported stft()
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
np.int = int  # noqa


def load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def synth_wav(rng, n):
    x = 0.05 * rng.standard_normal(n)
    y = np.empty(n)
    acc = 0.0
    for i in range(n):
        acc = x[i] + 0.9 * acc
        y[i] = acc
    y *= 0.5 / np.abs(y).max()
    return y.astype(np.float32)


def gen_fbank():
    sys.path.insert(0, REF)
    from simulation.freq_analysis import stft  # the reference's own STFT
    pre = load_by_path("ref_preprocess", "reader/preprocess.py")
    with open(os.path.join(REF, "data/mel80_window.txt")) as f:
        lines = [line.rstrip("\n") for line in f]
    window = np.vstack([np.asarray([np.float32(j) for j in i.split(",")]) for i in lines])

    def logfbank(wav):  # data/sr_dataset.py:279-296 around the imported stft
        preemphasis = 0.96
        t1 = np.sum(window, 0)
        t1[t1 == 0] = -1
        inv = np.diag(1 / t1)
        mel = window.dot(inv).T
        wav = wav[1:] - preemphasis * wav[:-1]
        S = stft(wav, n_fft=512, hop_length=160, win_length=400, window=np.hamming(400), center=False).T
        spec_power = np.abs(S) ** 2
        return np.log(spec_power.T.dot(mel * 32768 ** 2) + 1)

    rng = np.random.default_rng(7)
    out = dict(mel=window)
    lens = [401, 560, 561, 562, 4000, 16401, 33333]
    out["lens"] = np.asarray(lens)
    for i, n in enumerate(lens):
        wav = synth_wav(rng, n)
        if i == 4:
            wav[:] = 0.0  # all-zero input -> log(0+1) = 0 exactly
        fb = logfbank(wav)
        out["wav%d" % i] = wav
        out["fbank%d" % i] = np.asarray(fb)
        out["cmn%d" % i] = np.asarray(pre.cmn(fb, axis=0)).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "fbank.npz"), **out)
    print("fbank", {k: v.shape for k, v in out.items() if k.startswith("fbank")}, out["fbank0"].dtype)


def gen_lstm():
    lstm_mod = load_by_path("ref_lstm", "models/lstm.py")
    out = {}
    for tag, (D_in, P, H, Lr, bi, B, T) in dict(small=(20, 37, 64, 2, True, 3, 7),
                                                uni=(16, 11, 64, 1, False, 2, 5)).items():
        torch.manual_seed(0)
        m = lstm_mod.LSTMAM(D_in, P, H, Lr, 0.0, bi)
        x = torch.randn(B, T, D_in)
        logits = m.output_layer(m.lstm(x)[0])
        w = torch.randn(B, T, P)
        (logits * w).sum().backward()
        out[tag + "_cfg"] = np.asarray([D_in, P, H, Lr, int(bi), B, T])
        out[tag + "_x"] = x.numpy()
        out[tag + "_w"] = w.numpy()
        out[tag + "_logits"] = logits.detach().numpy()
        for k, v in m.state_dict().items():
            out[tag + "_param_" + k] = v.numpy().copy()
        for k, v in m.named_parameters():
            out[tag + "_grad_" + k] = v.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "lstm.npz"), **out)
    print("lstm keys", len(out))
    # default-initialisation pin at the real size: a few weights of the 3x512 model after manual_seed(0)
    torch.manual_seed(0)
    m = lstm_mod.LSTMAM(80, 5768, 512, 3, 0.2, True)
    sd = m.state_dict()
    pin = {k: v.reshape(-1)[:16].numpy().copy() for k, v in sd.items()}
    pin["__num_params"] = np.asarray(sum(v.numel() for v in sd.values()))
    pin["__keys"] = np.asarray(list(sd.keys()))
    np.savez_compressed(os.path.join(OUT, "lstm_init.npz"), **pin)


def gen_ce_optim_misc():
    utils = load_by_path("ref_utils", "utils/utils.py")
    dl = load_by_path("ref_dataloader", "data/dataloader.py")
    out = {}
    # CrossEntropyLoss(ignore_index=-100), mean and sum (bin/train_ce.py:134; bin/train_se.py:214)
    torch.manual_seed(1)
    logits = (3 * torch.randn(23, 301)).requires_grad_()
    tgt = torch.randint(0, 301, (23,))
    tgt[[2, 9, 22]] = -100
    for red in ("mean", "sum"):
        logits.grad = None
        loss = torch.nn.CrossEntropyLoss(ignore_index=-100, reduction=red)(logits, tgt)
        loss.backward()
        out["ce_loss_" + red] = loss.detach().numpy()
        out["ce_grad_" + red] = logits.grad.numpy().copy()
    out["ce_logits"] = logits.detach().numpy()
    out["ce_targets"] = tgt.numpy()
    # noam_decay (utils/utils.py:49-53)
    steps = np.asarray([1, 10, 3999, 4000, 4001, 100000])
    out["noam_steps"] = steps
    out["noam_lr"] = np.asarray([utils.noam_decay(int(s), 4000, 1e-3) for s in steps])
    # Adam(amsgrad) + clip_grad_norm_ (bin/train_ce.py:123,195) and SGD(momentum) (bin/train_se.py:127)
    torch.manual_seed(2)
    p0 = torch.randn(1000)
    grads = [torch.randn(1000) * s for s in (0.1, 3.0, 0.5)]
    out["opt_p0"] = p0.numpy().copy()
    out["opt_grads"] = np.stack([g.numpy() for g in grads])
    for name, mk in dict(adam=lambda p: torch.optim.Adam([p], lr=1e-2, amsgrad=True),
                         sgd=lambda p: torch.optim.SGD([p], lr=1e-2, momentum=0.9)).items():
        p = torch.nn.Parameter(p0.clone())
        opt = mk(p)
        norms, traj = [], []
        for g in grads:
            opt.zero_grad()
            p.grad = g.clone()
            norms.append(float(torch.nn.utils.clip_grad_norm_([p], 5.0)))
            opt.step()
            traj.append(p.detach().numpy().copy())
        out["opt_%s_norms" % name] = np.asarray(norms)
        out["opt_%s_traj" % name] = np.stack(traj)
    # SeqDataloader.collate_fn / ChunkDataloader.collate_fn (data/dataloader.py:55-63,94-136)
    rng = np.random.default_rng(3)

    class Dummy:
        test_only = False
    lens = [7, 5, 9]
    feats = [rng.standard_normal((n, 80)).astype(np.float32) for n in lens]
    labels = [rng.integers(0, 100, size=(n, 1)) for n in lens]
    batch = [(f, "utt%d" % i, l, np.zeros((1, 3), dtype=np.int64)) for i, (f, l) in enumerate(zip(feats, labels))]
    data = dl.SeqDataloader.collate_fn(Dummy(), batch)
    out["collate_lens"] = np.asarray(lens)
    out["collate_feats"] = np.concatenate(feats)
    out["collate_labels"] = np.concatenate(labels)
    out["collate_x"] = data["x"].numpy()
    out["collate_y"] = data["y"].numpy()
    out["collate_num_frs"] = np.asarray(data["num_frs"])
    # frame subsampling of bin/train_chain.py:251-255 for epochs 0..2
    x = data["x"]
    for epoch in range(3):
        shift = (epoch % 3) * -1
        xs = torch.roll(x.to(torch.float32), shift, 1).unfold(1, 1, 3).squeeze(-1)
        out["subsample_e%d" % epoch] = xs.numpy()
    np.savez_compressed(os.path.join(OUT, "misc.npz"), **out)
    print("misc keys", len(out))


def gen_chunk_mvn():
    """_utt2seg (data/sr_dataset.py:40-52; only that function is compiled: importing the module would import
    `reader`) and GlobalMeanVarianceNormalization (reader/preprocess.py:89-229)."""
    import ast
    src = open(os.path.join(REF, "data/sr_dataset.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "_utt2seg"][0]
    ns = {"np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "ref:_utt2seg", "exec"), ns)
    utt2seg = ns["_utt2seg"]
    pre = load_by_path("ref_preprocess2", "reader/preprocess.py")
    rng = np.random.default_rng(11)
    out = {}
    for i, T in enumerate([79, 80, 81, 255, 401]):
        feats = rng.standard_normal((T, 80)).astype(np.float32)
        labels = rng.integers(0, 5768, size=(T, 1))
        fs = utt2seg(feats.T, 80, 80)
        ls = utt2seg(labels.T, 80, 80)
        out["feats%d" % i] = feats
        out["labels%d" % i] = labels
        out["seg_x%d" % i] = np.stack([f.T for f in fs]) if fs else np.zeros((0, 80, 80), np.float32)
        out["seg_y%d" % i] = np.stack([l.T for l in ls]) if ls else np.zeros((0, 80, 1), np.int64)
    mvn = pre.GlobalMeanVarianceNormalization()
    mvn.initialize_stats(80)
    data = [(3.0 * rng.standard_normal((50 + 7 * k, 80)) + 5.0).astype(np.float32) for k in range(6)]
    data[0][:, 3] = 2.0                       # constant dimension -> std floored at 1e-2
    for d in data:
        mvn.accumulate_stats(d)
    mvn.learn_mean_and_variance_from_stats()
    out["mvn_mean"] = mvn.mean_vec
    out["mvn_std"] = mvn.std_vec
    out["mvn_in"] = data[2]
    out["mvn_out"] = mvn.apply_on_ndarray(data[2])
    np.savez_compressed(os.path.join(OUT, "chunk_mvn.npz"), **out)
    print("chunk_mvn keys", len(out))


def gen_e2e():
    """north_star's "same minibatch" clause end to end: raw wav -> the reference's stft-based log-mel fbank
    (data/sr_dataset.py:279-296) -> cmn (reader/preprocess.py:34-41) -> SeqDataloader.collate_fn zero padding
    (data/dataloader.py:94-136) -> the reference's LSTMAM (models/lstm.py, 3x512 bidirectional, P = 5768, default
    initialisation after torch.manual_seed(0), eval mode) -> frame log-posteriors on the CPU.  The fixture holds the
    waveforms and a slice of the log-posteriors (every 16th pdf, all frames) plus each frame's logsumexp and argmax; the
    21 M weights are not stored: the test re-creates them with the same seed (tests/golden/lstm_init.npz pins that the
    default initialisation is bit-identical)."""
    sys.path.insert(0, REF)
    from simulation.freq_analysis import stft
    pre = load_by_path("ref_preprocess3", "reader/preprocess.py")
    dl = load_by_path("ref_dataloader3", "data/dataloader.py")
    lstm_mod = load_by_path("ref_lstm3", "models/lstm.py")
    with open(os.path.join(REF, "data/mel80_window.txt")) as f:
        window = np.vstack([np.asarray([np.float32(j) for j in line.rstrip("\n").split(",")]) for line in f])

    def logfbank(wav):
        t1 = np.sum(window, 0)
        t1[t1 == 0] = -1
        mel = window.dot(np.diag(1 / t1)).T
        wav = wav[1:] - 0.96 * wav[:-1]
        S = stft(wav, n_fft=512, hop_length=160, win_length=400, window=np.hamming(400), center=False).T
        return np.log((np.abs(S) ** 2).T.dot(mel * 32768 ** 2) + 1)

    rng = np.random.default_rng(21)
    lens = [20801, 14400]                                   # 1.3 s and 0.9 s -> 129 and 89 frames
    wavs = [synth_wav(rng, n) for n in lens]
    feats = [np.asarray(pre.cmn(logfbank(w), axis=0)).astype(np.float32) for w in wavs]

    class Dummy:
        test_only = False
    batch = [(f, "utt%d" % i, np.zeros((f.shape[0], 1), dtype=np.int64), np.zeros((1, 3), dtype=np.int64))
             for i, f in enumerate(feats)]
    data = dl.SeqDataloader.collate_fn(Dummy(), batch)
    torch.manual_seed(0)
    m = lstm_mod.LSTMAM(80, 5768, 512, 3, 0.2, True).eval()
    with torch.no_grad():
        x = data["x"].to(torch.float32)
        logits = m.output_layer(m.lstm(x)[0])               # models/lstm.py:59 with the typo fixed
        logp = torch.log_softmax(logits.double(), dim=-1)
    out = dict(lens=np.asarray(lens), frames=np.asarray([f.shape[0] for f in feats]))
    for i, w in enumerate(wavs):
        out["wav%d" % i] = w
        out["feats%d" % i] = feats[i]
    out["logp_sub"] = logp[:, :, ::16].numpy().astype(np.float32)
    out["lse"] = torch.logsumexp(logits.double(), dim=-1).numpy()
    out["argmax"] = logits.argmax(-1).numpy()
    out["logits_sub"] = logits[:, :, ::16].numpy()
    np.savez_compressed(os.path.join(OUT, "e2e_ce.npz"), **out)
    print("e2e", out["logp_sub"].shape, out["frames"])


def gen_transformer():
    """models/transformer.py:52-94 imported as it is.  Fix-up: nn.TransformerEncoder's
    forward reads attributes of the custom layer that torch >= 2 expects of nn.TransformerEncoderLayer, so the module's
    own layers are called one after the other (what nn.TransformerEncoder.forward does on the pinned torch 1.2), then
    its final norm and the output layer.  The positional encoder is commented out in the reference's forward."""
    tr = load_by_path("ref_transformer", "models/transformer.py")
    out = {}
    for tag, (D, C, H, FF, L, P, T, B, look) in dict(small=(20, 128, 2, 96, 2, 37, 19, 3, 2),
                                                     heads16=(24, 64, 4, 128, 3, 29, 40, 2, -1)).items():
        torch.manual_seed(1)
        m = tr.TransformerAM(D, C, H, FF, L, 0.0, P)
        for lp in m.transformer.layers:      # nn.TransformerEncoder deep-copies one initialised layer: break the symmetry
            for p_ in lp.parameters():
                p_.data.add_(0.02 * torch.randn_like(p_))
        x = torch.randn(T, B, D)
        lens = [T] + [max(1, T - 4 - 3 * i) for i in range(B - 1)]
        kpm = torch.ones(B, T)
        for i, n in enumerate(lens):
            kpm[i, :n] = 0
        kpm = kpm.bool()
        src_mask = None
        if look > -1:
            tri = torch.tril(torch.ones(T, T), diagonal=look)
            src_mask = tri.float().masked_fill(tri == 0, float("-inf")).masked_fill(tri == 1, 0.0)
        h = m.input_layer(x)
        for layer in m.transformer.layers:
            h = layer(h, src_mask, kpm)      # TransformerEncoderLayerWithConv1d.forward, models/transformer.py:62-66
        logits = m.output_layer(m.transformer.norm(h))
        w = torch.randn(T, B, P)
        for i, n in enumerate(lens):
            w[n:, i] = 0                     # padded query rows carry no loss in the reference's training scripts
        (logits * w).sum().backward()
        out[tag + "_cfg"] = np.asarray([D, C, H, FF, L, P, T, B, look])
        out[tag + "_lens"] = np.asarray(lens)
        out[tag + "_x"] = x.numpy()
        out[tag + "_w"] = w.numpy()
        out[tag + "_logits"] = logits.detach().numpy()
        for k, v in m.state_dict().items():
            if k != "pos_encoder.pe":        # a 5000 x C sine table, unused by the reference's forward: first rows only
                out[tag + "_param_" + k] = v.numpy().copy()
        out[tag + "_pe_head"] = m.state_dict()["pos_encoder.pe"][:8].numpy().copy()
        for k, v in m.named_parameters():
            out[tag + "_grad_" + k] = v.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "transformer.npz"), **out)
    print("transformer keys", len(out))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gen_chunk_mvn()
    gen_fbank()
    gen_lstm()
    gen_ce_optim_misc()
    gen_e2e()
    gen_transformer()
