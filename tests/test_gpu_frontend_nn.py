"""HIP front end, LSTM, CE, GEMM and optimiser kernels against reference-generated goldens and
the reference PyTorch CPU path (nn.LSTM + nn.Linear, exactly what models/lstm.py:45-54 builds)."""
import numpy as np
import pytest
import torch

from oracle import frontend_ref as F
from pykaldi2_amd import _lib, fbank, lstm, ops, optim

pytestmark = pytest.mark.gpu


def test_fbank_cmn_match_reference_golden(golden):
    g = golden("fbank")
    fb = fbank.FbankExtractor()
    lens = [int(n) for n in g["lens"]]
    wav = torch.from_numpy(np.concatenate([g["wav%d" % i] for i in range(len(lens))])).cuda()
    feats, frames, row_off = fb(wav, lens, apply_cmn=False)
    feats_c, _, _ = fb(wav, lens, apply_cmn=True)
    feats, feats_c = feats.cpu().numpy(), feats_c.cpu().numpy()
    off = np.concatenate([[0], np.cumsum(frames)])
    worst = 0.0
    for i in range(len(lens)):
        want = g["fbank%d" % i]
        got = feats[off[i]:off[i + 1]]
        assert got.shape == want.shape
        # log-mel values are O(10); f32 FFT vs the reference's f64 FFT rounded to complex64: measured 2e-5 on the
        # GPU box (round 2), so 2e-4 leaves one decade (round 1 allowed 2e-3 without having measured)
        worst = max(worst, np.abs(got - want).max(), np.abs(feats_c[off[i]:off[i + 1]] - g["cmn%d" % i]).max())
        assert np.abs(got - want).max() < 2e-4, (i, np.abs(got - want).max())
        assert np.abs(feats_c[off[i]:off[i + 1]] - g["cmn%d" % i]).max() < 2e-4
    assert not feats[off[4]:off[5]].any()
    print("fbank / CMN max abs error vs the reference: %.3g" % worst)


def test_wav_to_ce_posteriors_end_to_end_matches_reference_cpu_path(golden):
    """north_star: "outputs match the reference PyTorch CPU path on the same minibatch -- CE frame posteriors within
    1e-4 rel".  The whole chain at once, raw waveforms in: on-device fbank + CMN + zero padding + 3x512 BLSTM (P = 5768,
    default initialisation after manual_seed(0), eval) against the reference's own stft / cmn / collate_fn / LSTMAM
    run on the CPU by tools/gen_golden.py (tests/golden/e2e_ce.npz).  A relative error of a posterior is an absolute
    error of its logarithm."""
    g = golden("e2e_ce")
    lens = [int(n) for n in g["lens"]]
    frames = [int(n) for n in g["frames"]]
    fb = fbank.FbankExtractor()
    wav = torch.from_numpy(np.concatenate([g["wav%d" % i] for i in range(len(lens))])).cuda()
    feats, fr, row_off = fb(wav, lens, apply_cmn=True)
    assert list(fr) == frames
    off = np.concatenate([[0], np.cumsum(frames)])
    fbank_err = max(np.abs(feats[off[i]:off[i + 1]].cpu().numpy() - g["feats%d" % i]).max() for i in range(len(lens)))
    x = fb.pad_roll_subsample(feats, row_off, fr, shift=0, subsample=1, time_major=True)
    torch.manual_seed(0)
    m = lstm.LSTMAM(80, 5768, 512, 3, 0.2, True).cuda().eval()
    with torch.no_grad():
        logits = m.forward_time_major(x).transpose(0, 1)            # [B, T, P]
        logp = torch.log_softmax(logits.double(), dim=-1).cpu().numpy()
        lse = torch.logsumexp(logits.double(), dim=-1).cpu().numpy()
        am = logits.argmax(-1).cpu().numpy()
    worst = 0.0
    for i, T in enumerate(frames):
        err = np.abs(logp[i, :T, ::16] - g["logp_sub"][i, :T]).max()
        worst = max(worst, err, np.abs(lse[i, :T] - g["lse"][i, :T]).max())
        assert (am[i, :T] == g["argmax"][i, :T]).mean() > 0.99
    print("end-to-end: fbank+CMN max abs error %.3g, log-posterior max abs error %.3g" % (fbank_err, worst))
    assert fbank_err < 5e-4, fbank_err
    assert worst < 1e-4, worst


def test_pad_roll_subsample_matches_reference_golden(golden):
    g = golden("misc")
    lens = [int(v) for v in g["collate_lens"]]
    feats = torch.from_numpy(g["collate_feats"]).cuda()
    row_off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64).cuda()
    x = fbank.FbankExtractor.pad_roll_subsample(feats, row_off, lens)
    assert np.array_equal(x.cpu().numpy(), g["collate_x"])
    for e in range(3):
        xs = fbank.FbankExtractor.pad_roll_subsample(feats, row_off, lens, shift=-(e % 3), subsample=3)
        assert np.array_equal(xs.cpu().numpy(), g["subsample_e%d" % e])
        xt = fbank.FbankExtractor.pad_roll_subsample(feats, row_off, lens, shift=-(e % 3), subsample=3, time_major=True)
        assert np.array_equal(xt.transpose(0, 1).cpu().numpy(), g["subsample_e%d" % e])


@pytest.mark.parametrize("ta,tb,M,N,K", [(0, 1, 257, 130, 80), (0, 0, 100, 37, 515), (1, 0, 300, 64, 1601),
                                         (0, 1, 1600, 2048, 1024), (1, 1, 33, 129, 18),
                                         # row bands (608 tiles), split-K, exactly one tile per CU, a partial last k-slab
                                         # behind the pipelined loop, a shallow product on 64x64 tiles
                                         (0, 1, 2356, 4096, 520), (0, 0, 300, 1024, 4100), (0, 1, 2048, 2048, 512), (0, 1, 2356, 6048, 600),
                                         (0, 1, 1280, 3200, 48), (1, 0, 700, 260, 2357), (1, 1, 640, 520, 1030),
                                         # few tiles and a short K (the TransformerAM's products), all four operand layouts, ragged
                                         # M / N edges, a K that is no multiple of 16
                                         (0, 1, 2276, 512, 512), (0, 0, 2276, 512, 1536), (1, 0, 500, 130, 772), (1, 1, 70, 190, 260),
                                         (0, 1, 65, 64, 1028), (0, 0, 129, 67, 256),
                                         # tile counts just above a multiple of the CU count, ragged edges in both dimensions (edge
                                         # tiles clamp their rows), row counts that are no multiple of four in the contiguous dimension
                                         # (those edge tiles keep the bounds-tested loader)
                                         (0, 1, 576, 1856, 144), (0, 0, 1024, 1028, 256), (0, 1, 2354, 512, 512), (1, 0, 2500, 4096, 250),
                                         (0, 0, 1100, 642, 384), (1, 1, 901, 700, 200), (1, 0, 2502, 1030, 300),
                                         # large grids (several rounds of tiles per CU), all four layouts, ragged edges
                                         (0, 1, 4096, 4096, 100), (0, 0, 6200, 4220, 72), (1, 0, 6200, 4222, 40), (1, 1, 8190, 4096, 48)])
@pytest.mark.parametrize("arith", ["f32", "bf16x3"])
def test_gemm_f32_matches_float64(ta, tb, M, N, K, arith, gemm_arith):
    # both arithmetic paths of pk2_gemm_f32 (f32 MFMA; three-way bf16 split on the bf16 MFMA, csrc/gemm_bf16x3.h) at ONE bound
    gemm_arith(arith)
    rng = np.random.default_rng(M + N)
    A = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    B = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    want = 0.5 * ((A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)) + bias + 2.0 * C0
    a, b, c, bs = (torch.from_numpy(v).cuda() for v in (A, B, C0.copy(), bias))
    lstm._gemm(ta, tb, M, N, K, lstm._p(a), A.shape[1], lstm._p(b), B.shape[1], lstm._p(c), N, bias=lstm._p(bs),
               alpha=0.5, beta=2.0)
    got = c.cpu().numpy()
    err = np.abs(got - want).max()
    assert err < 2e-4 * np.sqrt(K), err
    # the same call again gives the same bits (partial tiles are added in part order, whoever arrives last)
    c2 = torch.from_numpy(C0.copy()).cuda()
    lstm._gemm(ta, tb, M, N, K, lstm._p(a), A.shape[1], lstm._p(b), B.shape[1], lstm._p(c2), N, bias=lstm._p(bs),
               alpha=0.5, beta=2.0)
    big_tiles = -(-M // 128) * -(-N // 128)
    deep_k = 1536 if big_tiles <= 16 else 2048
    if not (K >= deep_k and (big_tiles <= 256 or (ta == 1 and tb == 0 and big_tiles < 512))):   # (the deep-K split adds its slices with float atomics)
        assert np.array_equal(c2.cpu().numpy(), got)


def _load_golden_model(g, tag):
    D_in, P, H, Lr, bi, B, T = [int(v) for v in g[tag + "_cfg"]]
    m = lstm.LSTMAM(D_in, P, H, Lr, 0.0, bool(bi))
    sd = {k[len(tag + "_param_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "_param_")}
    m.load_state_dict(sd)
    return m.cuda(), (D_in, P, H, Lr, bi, B, T)


@pytest.mark.parametrize("tag", ["small", "uni"])
def test_lstmam_matches_reference_golden(golden, tag):
    g = golden("lstm")
    m, _ = _load_golden_model(g, tag)
    x = torch.from_numpy(g[tag + "_x"]).cuda()
    logits = m(x)
    assert logits.is_contiguous()
    err = (logits.cpu() - torch.from_numpy(g[tag + "_logits"])).abs().max().item()
    assert err < 2e-5, err
    (logits * torch.from_numpy(g[tag + "_w"]).cuda()).sum().backward()
    for name, p in m.named_parameters():
        want = g[tag + "_grad_" + name]
        e = np.abs(p.grad.cpu().numpy() - want).max()
        assert e < 1e-4 * max(1.0, np.abs(want).max()), (name, e)


def test_blstm_3x512_posteriors_match_torch_cpu():
    """CE frame posteriors within 1e-4 rel of the reference PyTorch CPU path (north star)."""
    torch.manual_seed(0)
    B, T, P = 4, 23, 5768
    m = lstm.LSTMAM(80, P, 512, 3, 0.0, True)
    ref_lstm = torch.nn.LSTM(80, 512, 3, batch_first=True, bidirectional=True)
    ref_out = torch.nn.Linear(1024, P)
    ref_lstm.load_state_dict({k[5:]: v for k, v in m.state_dict().items() if k.startswith("lstm.")})
    ref_out.load_state_dict({k[13:]: v for k, v in m.state_dict().items() if k.startswith("output_layer.")})
    x = torch.randn(B, T, 80)
    tgt = torch.randint(0, P, (B, T))
    tgt[1, 20:] = -100
    ref_logits = ref_out(ref_lstm(x)[0])
    ref_loss = torch.nn.CrossEntropyLoss(ignore_index=-100)(ref_logits.view(-1, P), tgt.view(-1))
    ref_loss.backward()
    m = m.cuda()
    logits = m(x.cuda())
    post = torch.softmax(logits.double().cpu(), -1)
    ref_post = torch.softmax(ref_logits.double().detach(), -1)
    rel = ((post - ref_post).abs() / ref_post.clamp_min(1e-30)).max().item()
    assert rel < 1e-4, rel
    loss = ops.CrossEntropyLoss(ignore_index=-100)(logits.view(-1, P), tgt.cuda().view(-1))
    assert abs(loss.item() - ref_loss.item()) < 1e-5 * abs(ref_loss.item())
    loss.backward()
    refg = dict(list(("lstm." + k, v.grad) for k, v in ref_lstm.named_parameters()) +
                list(("output_layer." + k, v.grad) for k, v in ref_out.named_parameters()))
    for name, p in m.named_parameters():
        want = refg[name]
        e = (p.grad.cpu() - want).abs().max().item()
        assert e < 2e-4 * max(1e-3, want.abs().max().item()), (name, e, want.abs().max().item())


@pytest.mark.parametrize("impl", ["pair_per_xcd", "direction_per_xcd"])
@pytest.mark.parametrize("B,T,bi,layers", [(3, 301, True, 1), (4, 150, False, 2), (1, 64, True, 1), (8, 120, True, 1), (6, 77, True, 2)])
def test_persistent_recurrence_long_sequences_match_torch_cpu(B, T, bi, layers, impl, monkeypatch):
    """The persistent small-batch recurrences (H = 512): csrc/lstm_persist_seq.hip, a (sequence, direction) pair per XCD
    (the default; more pairs than XCDs queue up: B = 6 / 8 bidirectional), and csrc/lstm_persist.hip (PK2_LSTM_SEQ=0), a
    direction per XCD with B <= 8 in groups of 4 batch rows; h and the d h partials exchanged through the XCD's L2 with
    double-buffered sentinel mailboxes, over hundreds of steps, partial batches and one direction, against the reference's
    torch CPU nn.LSTM: outputs and every gradient."""
    if impl == "direction_per_xcd":
        monkeypatch.setenv("PK2_LSTM_SEQ", "0")
    torch.manual_seed(B * 1000 + T)
    H, Din, P = 512, 80, 40
    m = lstm.LSTMAM(Din, P, H, layers, 0.0, bi)
    ref_lstm = torch.nn.LSTM(Din, H, layers, batch_first=True, bidirectional=bi)
    ref_out = torch.nn.Linear(H * (2 if bi else 1), P)
    ref_lstm.load_state_dict({k[5:]: v for k, v in m.state_dict().items() if k.startswith("lstm.")})
    ref_out.load_state_dict({k[13:]: v for k, v in m.state_dict().items() if k.startswith("output_layer.")})
    x = torch.randn(B, T, Din)
    wgt = torch.randn(B, T, P)
    ref_logits = ref_out(ref_lstm(x)[0])
    (ref_logits * wgt).sum().backward()
    m = m.cuda()
    logits = m(x.cuda())
    err = (logits.cpu() - ref_logits.detach()).abs().max().item()
    assert err < 2e-4, err
    (logits * wgt.cuda()).sum().backward()
    refg = dict(list(("lstm." + k, v.grad) for k, v in ref_lstm.named_parameters()) +
                list(("output_layer." + k, v.grad) for k, v in ref_out.named_parameters()))
    for name, p in m.named_parameters():
        e = (p.grad.cpu() - refg[name]).abs().max().item()
        assert e < 2e-4 * max(1.0, refg[name].abs().max().item()), (name, e)
    from pykaldi2_amd import _lib
    flag = __import__("ctypes").c_uint32(7)
    assert _lib.lib().pk2_lstm_persist_status(__import__("ctypes").byref(flag)) == 0 and flag.value == 0   # no poll timed out


@pytest.mark.parametrize("B,T,H,bi", [(70, 9, 128, True), (256, 5, 512, True), (33, 6, 64, False),
                                      (7, 9, 128, True), (20, 6, 256, True), (3, 11, 64, False), (31, 4, 512, True),
                                      (70, 7, 512, True), (33, 6, 512, False), (130, 12, 512, True), (600, 3, 512, True)])
def test_large_batch_lstm_matches_torch_cpu(B, T, H, bi):
    """B >= 32 takes the GEMM-tiled recurrence kernels (ragged last 64-row tile at B=70 / 33) -- at H = 512 the persistent
    ones of csrc/lstm_persist_big.hip (a (direction, 64-row batch tile) task per XCD team; B = 600: 20 tasks for the 8
    teams, which take turns from the queue); smaller batches the step kernels that loop over groups of 4 batch rows
    (4x4x1 MFMA)."""
    torch.manual_seed(1)
    P, Din = 50, 40
    m = lstm.LSTMAM(Din, P, H, 2, 0.0, bi)
    ref_lstm = torch.nn.LSTM(Din, H, 2, batch_first=True, bidirectional=bi)
    ref_out = torch.nn.Linear(H * (2 if bi else 1), P)
    ref_lstm.load_state_dict({k[5:]: v for k, v in m.state_dict().items() if k.startswith("lstm.")})
    ref_out.load_state_dict({k[13:]: v for k, v in m.state_dict().items() if k.startswith("output_layer.")})
    x = torch.randn(B, T, Din)
    w = torch.randn(B, T, P)
    ref_logits = ref_out(ref_lstm(x)[0])
    (ref_logits * w).sum().backward()
    m = m.cuda()
    logits = m(x.cuda())
    err = (logits.cpu() - ref_logits.detach()).abs().max().item()
    assert err < 2e-5, err
    (logits * w.cuda()).sum().backward()
    refg = dict(list(("lstm." + k, v.grad) for k, v in ref_lstm.named_parameters()) +
                list(("output_layer." + k, v.grad) for k, v in ref_out.named_parameters()))
    for name, p in m.named_parameters():
        want = refg[name]
        e = (p.grad.cpu() - want).abs().max().item()
        assert e < 2e-4 * max(1e-3, want.abs().max().item()), (name, e, want.abs().max().item())
    from pykaldi2_amd import _lib
    flag = __import__("ctypes").c_uint32(7)
    assert _lib.lib().pk2_lstm_persist_status(__import__("ctypes").byref(flag)) == 0 and flag.value == 0   # no poll timed out


def test_large_batch_backward_all_gather_form():
    """The large-batch backward recurrence has two forms (csrc/lstm_persist_big.hip): the default 4 x 8 decomposition of the
    team with a second exchange of partial sums, and the all-gather form (PK2_LSTM_BIG_BWD=1, read once per process): the
    H = 512 cases of the test above in a child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_frontend_nn.py"), "-q", "-m", "gpu",
                          "-k", "test_large_batch_lstm_matches_torch_cpu and 512"], env=dict(os.environ, PK2_LSTM_BIG_BWD="1"),
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "failed" not in out.stdout


def test_full_size_ce_configuration_matches_torch_cpu():
    """configs[1] at its full size (VERDICT r2 #1d): 256 chunks x 80 frames x 80 fbank bins, 3x512 BLSTM, P = 5768,
    dropout 0 -- the large-batch recurrence kernels at the shape bench.py --ce times them -- against the reference's
    torch CPU path (nn.LSTM + nn.Linear = models/lstm.py:45-54, nn.CrossEntropyLoss): frame posteriors 1e-4 rel
    (north star), the loss, and every parameter gradient."""
    torch.manual_seed(3)
    B, T, P = 256, 80, 5768
    m = lstm.LSTMAM(80, P, 512, 3, 0.0, True)
    ref_lstm = torch.nn.LSTM(80, 512, 3, batch_first=True, bidirectional=True)
    ref_out = torch.nn.Linear(1024, P)
    ref_lstm.load_state_dict({k[5:]: v for k, v in m.state_dict().items() if k.startswith("lstm.")})
    ref_out.load_state_dict({k[13:]: v for k, v in m.state_dict().items() if k.startswith("output_layer.")})
    x = torch.randn(B, T, 80)
    tgt = torch.randint(0, P, (B, T))
    tgt[5, 60:] = -100
    ref_logits = ref_out(ref_lstm(x)[0])
    ref_loss = torch.nn.CrossEntropyLoss(ignore_index=-100)(ref_logits.view(-1, P), tgt.view(-1))
    ref_loss.backward()
    m = m.cuda()
    logits = m.forward_time_major(x.cuda().transpose(0, 1).contiguous()).transpose(0, 1)       # as bench.py --ce
    # posteriors in float64 from both sets of logits, a slab of rows at a time (B*T*P doubles would be 9 GB)
    rel = 0.0
    for b0 in range(0, B, 32):
        post = torch.softmax(logits[b0:b0 + 32].double().cpu(), -1)
        ref_post = torch.softmax(ref_logits[b0:b0 + 32].double().detach(), -1)
        rel = max(rel, ((post - ref_post).abs() / ref_post.clamp_min(1e-30)).max().item())
    assert rel < 1e-4, rel
    loss = ops.CrossEntropyLoss(ignore_index=-100)(logits, tgt.cuda())
    assert abs(loss.item() - ref_loss.item()) < 1e-5 * abs(ref_loss.item())
    loss.backward()
    refg = dict(list(("lstm." + k, v.grad) for k, v in ref_lstm.named_parameters()) +
                list(("output_layer." + k, v.grad) for k, v in ref_out.named_parameters()))
    for name, p in m.named_parameters():
        want = refg[name]
        e = (p.grad.cpu() - want).abs().max().item()
        assert e < 2e-4 * max(1e-3, want.abs().max().item()), (name, e, want.abs().max().item())


def test_cross_entropy_matches_reference_golden(golden):
    g = golden("misc")
    for red in ("mean", "sum"):
        x = torch.from_numpy(g["ce_logits"]).cuda().requires_grad_()
        loss = ops.CrossEntropyLoss(ignore_index=-100, reduction=red)(x, torch.from_numpy(g["ce_targets"]).cuda())
        loss.backward()
        assert abs(loss.item() - float(g["ce_loss_" + red])) < 1e-5 * abs(float(g["ce_loss_" + red]))
        assert np.abs(x.grad.cpu().numpy() - g["ce_grad_" + red]).max() < 1e-6
    # a label outside [0, P) (not ignore_index) never indexes the row: the loss turns NaN, that row's gradient is zero
    tg = torch.from_numpy(g["ce_targets"]).clone()
    tg[0] = g["ce_logits"].shape[1] + 5
    x = torch.from_numpy(g["ce_logits"]).cuda().requires_grad_()
    loss = ops.CrossEntropyLoss(ignore_index=-100, reduction="sum")(x, tg.cuda())
    loss.backward()
    assert torch.isnan(loss).item() and not x.grad[0].any().item() and torch.isfinite(x.grad).all().item()


@pytest.mark.parametrize("red", ["mean", "sum"])
def test_cross_entropy_incoming_gradient_and_repeated_backward(red, monkeypatch):
    """The gradient of the mean loss leaves the kernel final (valid targets counted in front); the loss's incoming gradient is
    applied in place and skipped when it is 1: scaled losses, a second backward over a retained graph with another factor,
    targets with ignored and out-of-count rows, and the old scaling pass (PK2_CE_SCALE_PASS=1) against torch on the host."""
    torch.manual_seed(3)
    rows, P = 300, 517
    x0 = torch.randn(rows, P) * 3
    tg = torch.randint(0, P, (rows,))
    tg[::7] = -100
    for mode in ("0", "1"):
        monkeypatch.setenv("PK2_CE_SCALE_PASS", mode)
        ref = x0.clone().double().requires_grad_()
        lref = torch.nn.CrossEntropyLoss(ignore_index=-100, reduction=red)(ref, tg)
        x = x0.clone().cuda().requires_grad_()
        loss = ops.CrossEntropyLoss(ignore_index=-100, reduction=red)(x, tg.cuda())
        assert abs(loss.item() - lref.item()) < 1e-5 * abs(lref.item())
        scale = max(1e-12, torch.autograd.grad(lref, ref, retain_graph=True)[0].abs().max().item())
        for factor in (1.0, 0.37, 1.0, -2.5):           # repeated backward passes over the retained graph
            want = torch.autograd.grad(lref * factor, ref, retain_graph=True)[0]
            got = torch.autograd.grad(loss * factor, x, retain_graph=True)[0]
            assert (got.cpu().double() - want).abs().max().item() < 2e-6 * scale * max(1.0, abs(factor)), (mode, factor)
    # every target ignored: count 0 -> the division is by max(1, 0)
    x = x0.clone().cuda().requires_grad_()
    loss = ops.CrossEntropyLoss(ignore_index=-100, reduction=red)(x, torch.full((rows,), -100).cuda())
    loss.backward()
    assert loss.item() == 0.0 and not x.grad.any().item()


class _Flat:
    def __init__(self, p):
        self.p, self.g = p, torch.zeros_like(p)

    def flat_parameters(self):
        return self.p, self.g

    def parameters(self):
        return [self.p]


@pytest.mark.parametrize("name", ["adam", "sgd"])
def test_fused_optimisers_match_reference_golden(golden, name):
    g = golden("misc")
    model = _Flat(torch.from_numpy(g["opt_p0"].copy()).cuda())
    opt = optim.Adam(model, lr=1e-2, amsgrad=True) if name == "adam" else optim.SGD(model, lr=1e-2, momentum=0.9)
    for i, grad in enumerate(g["opt_grads"]):
        opt.zero_grad()
        model.g.copy_(torch.from_numpy(grad))
        norm = optim.clip_grad_norm_(opt, 5.0)
        opt.step()
        assert abs(norm.item() - g["opt_%s_norms" % name][i]) < 1e-4 * g["opt_%s_norms" % name][i]
        assert np.abs(model.p.cpu().numpy() - g["opt_%s_traj" % name][i]).max() < 2e-6


@pytest.mark.parametrize("name", ["adam", "sgd"])
def test_raised_persist_guard_keeps_the_optimisers_off_the_weights(name):
    """A persistent kernel that timed out poisons its output with NaN and raises the per-device guard (include/pk2hip.h,
    csrc/persist_guard.h): from then on the fused optimiser kernels must leave weights and moments alone -- the launch is
    already queued when the host finds out -- and the next step() must raise.  The guard is raised here the way a check
    kernel does it (pk2_persist_guard_raise: a kernel on the stream), behind a first, healthy step."""
    from pykaldi2_amd import _lib
    L = _lib.lib()
    model = _Flat(torch.linspace(-1.0, 1.0, 4096).cuda())
    opt = optim.Adam(model, lr=1e-2, amsgrad=True) if name == "adam" else optim.SGD(model, lr=1e-2, momentum=0.9)
    try:
        model.g.fill_(0.5)
        opt.step()                                    # healthy
        before = model.p.clone()
        _lib.check(L.pk2_persist_guard_raise(_lib.stream_ptr()))
        model.g.fill_(float("nan"))                   # what a poisoned backward pass would leave
        try:                                          # the host may or may not have seen the word yet: either the call
            opt.step()                                # raises, or its kernel finds the device word set
        except _lib.Pk2Error:
            pass
        torch.cuda.synchronize()
        assert torch.equal(model.p, before) and bool(torch.isfinite(model.p).all())
        with pytest.raises(_lib.Pk2Error, match="persistent kernel"):
            opt.step()
        assert torch.equal(model.p, before)
    finally:
        _lib.check(L.pk2_persist_guard_clear())
    model.g.fill_(0.5)
    opt.step()                                        # lowered: updates again
    torch.cuda.synchronize()
    assert not torch.equal(model.p, before) and not _lib.persist_guard_raised()


def test_chunking_and_mvn_match_reference_golden(golden):
    g = golden("chunk_mvn")
    for i in range(5):
        x = fbank.utt2seg(torch.from_numpy(g["feats%d" % i]).cuda(), 80, 80)
        y = fbank.utt2seg(torch.from_numpy(g["labels%d" % i]).cuda(), 80, 80)
        assert np.array_equal(x.cpu().numpy(), g["seg_x%d" % i]) and np.array_equal(y.cpu().numpy(), g["seg_y%d" % i])
    mvn = fbank.GlobalMeanVarianceNormalization(g["mvn_mean"], g["mvn_std"])
    out = mvn.apply_on_tensor(torch.from_numpy(g["mvn_in"]).cuda()).cpu().numpy()
    assert np.abs(out - g["mvn_out"]).max() <= 1e-6 * np.abs(g["mvn_out"]).max()


def test_dropout_is_an_unbiased_mask_and_backward_reuses_it():
    torch.manual_seed(3)
    m = lstm.LSTMAM(20, 11, 64, 2, 0.5, True).cuda().train()
    x = torch.randn(3, 9, 20).cuda()
    torch.manual_seed(5)
    a = m(x)
    torch.manual_seed(5)
    b = m(x)
    assert torch.equal(a, b)                      # same seed -> same mask
    torch.manual_seed(6)
    assert not torch.equal(a, m(x))
    m.eval()
    e = m(x)
    # raw kernel: keep fraction and scaling
    v = torch.ones(1 << 20, device="cuda")
    out = torch.empty_like(v)
    _lib.check(_lib.lib().pk2_dropout_f32(_lib.ptr(v), _lib.ptr(out), v.numel(), 0.2, 1234, _lib.stream_ptr()))
    keep = (out != 0).float().mean().item()
    assert abs(keep - 0.8) < 2e-3 and torch.allclose(out[out != 0], torch.tensor(1.25, device="cuda"))
    # gradient check by finite differences through the dropped network (mask fixed by the seed)
    m.train()
    xg = x.clone().requires_grad_()
    torch.manual_seed(9)
    w = torch.randn_like(e)
    torch.manual_seed(7)
    (m(xg) * w).sum().backward()
    gx = xg.grad.clone()
    d = torch.zeros_like(x); d[1, 4, 3] = 1e-2
    torch.manual_seed(7); f1 = (m(x + d) * w).sum().item()
    torch.manual_seed(7); f0 = (m(x - d) * w).sum().item()
    assert abs((f1 - f0) / 2e-2 - gx[1, 4, 3].item()) < 2e-2 * max(1.0, abs(gx[1, 4, 3].item()))


@pytest.mark.parametrize("arith", ["f32", "bf16x3"])
@pytest.mark.parametrize("M,N,K", [(6048, 1024, 2276), (512, 512, 2300), (2048, 512, 96), (130, 70, 1601), (4096, 80, 2276),
                                   (1021, 260, 777)])
def test_weight_and_bias_gradient_in_one_launch(M, N, K, arith, gemm_arith):
    """pk2_gemm_f32_tn_colsum: C += A^T B and colsum += column sums of A (a Linear layer's weight and bias gradient; reference
    models/lstm.py:59 / nn.Linear backward) against float64 -- plain tiles, K slices (deep K, few tiles), 64x64 tiles, ragged
    edges (rows clamped in the loader must not be counted), an M that is no multiple of four (bounds-tested loader), both
    arithmetic paths (f32: the colsum kernel behind the product)."""
    gemm_arith(arith)
    rng = np.random.default_rng(M + K)
    A = rng.standard_normal((K, M)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    b0 = rng.standard_normal(M).astype(np.float32)
    a, b, c, bs = (torch.from_numpy(v).cuda() for v in (A, B, C0.copy(), b0.copy()))
    _lib.check(_lib.lib().pk2_gemm_f32_tn_colsum(M, N, K, 1.0, lstm._p(a), M, lstm._p(b), N, 1.0, lstm._p(c), N, lstm._p(bs),
                                                 _lib.stream_ptr()))
    want_c = A.T.astype(np.float64) @ B.astype(np.float64) + C0
    want_b = A.astype(np.float64).sum(0) + b0
    assert np.abs(c.cpu().numpy() - want_c).max() < 2e-4 * np.sqrt(K)
    assert np.abs(bs.cpu().numpy() - want_b).max() < 2e-5 * np.sqrt(K), np.abs(bs.cpu().numpy() - want_b).max()


def test_check_inside_the_recurrence_launch_poisons_and_raises_the_guard(monkeypatch):
    """Round 6: the verdict of a one-launch recurrence is taken by its LAST workgroup to leave (no check kernel behind it).
    PK2_LSTM_SEQ_TEST_FAIL=1 makes that verdict "a poll timed out": the layer's output must come back as NaN, the per-device
    guard must be up, and -- once it is lowered -- the next launch must run clean from the control block the failed one
    left zeroed."""
    L = _lib.lib()
    T, B, H, D = 37, 4, 512, 2
    torch.manual_seed(0)
    gx = (torch.randn(T, B, D * 4 * H, device="cuda") * 0.1)
    whh = torch.randn(D, 4 * H, H, device="cuda") * 0.04

    def run():
        y = torch.empty(T, B, D * H, device="cuda"); gates = torch.empty(D, T, B, 4 * H, device="cuda"); cells = torch.empty(D, T, B, H, device="cuda")
        _lib.check(L.pk2_lstm_layer_fwd(_lib.ptr(gx), _lib.ptr(whh), None, B, T, H, D, _lib.ptr(y), _lib.ptr(gates), _lib.ptr(cells), None,
                                        _lib.stream_ptr()))
        torch.cuda.synchronize()
        return y
    good = run()                     # (the first launch of a process is verified on the host; the second one folds its check)
    good = run()
    assert bool(torch.isfinite(good).all()) and not _lib.persist_guard_raised()
    try:
        monkeypatch.setenv("PK2_LSTM_SEQ_TEST_FAIL", "1")
        bad = run()
        monkeypatch.delenv("PK2_LSTM_SEQ_TEST_FAIL")
        assert bool(torch.isnan(bad).all()) and _lib.persist_guard_raised()
    finally:
        _lib.check(L.pk2_persist_guard_clear())
    again = run()
    assert torch.equal(again, good) and not _lib.persist_guard_raised()
