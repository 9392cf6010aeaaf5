"""GPU parity tests of the lattice path (row a10: MMIFunction / sMBRFunction) through the C ABI:
device decoder + lattice pruning vs oracle/lattice_ref.py::decode (token costs and link sets bit-exact),
lattice forward-backward (MMI, sMBR, MPFE) vs the oracle's float64 restatement of Kaldi."""
import struct

import numpy as np
import pytest
import torch

from oracle import lattice_ref as lr
from pykaldi2_amd import lattice, ops, synth

pytestmark = pytest.mark.gpu


def _setup(num_words, num_pdfs, T, seed, max_phones=3, scale=2.0):
    rng = np.random.default_rng(seed)
    g = synth.decoding_graph_arcs(num_words, num_pdfs, seed=seed, max_phones=max_phones)
    tm = synth.transition_model_arrays(num_pdfs)
    ll = (scale * rng.standard_normal((T, num_pdfs))).astype(np.float32)
    return g, tm, ll, rng


def _ref_graph(g):
    return lr.DecodeGraphRef(g["num_states"], g["start"], g["src"], g["dst"], g["ilabel"], g["weight"], g["final"])


def _recognizer(g, tm, beam, lattice_beam, ac, max_active=2 ** 31 - 1, min_active=200):
    o = lattice.LatticeFasterDecoderOptions(beam=beam, lattice_beam=lattice_beam, max_active=max_active, min_active=min_active)
    return lattice.MappedLatticeFasterRecognizer(lattice.TransitionModel.from_arrays(tm), g, ac, o)


def _canon(a):
    toks = sorted(zip(a["tok_frame"].tolist(), a["tok_state"].tolist(), a["tok_cost"].tolist(), a["tok_final"].tolist()))
    fr, st = a["tok_frame"], a["tok_state"]
    links = sorted((int(fr[s]), int(st[s]), int(st[d]), int(t), float(g), float(c))
                   for s, d, t, g, c in zip(a["link_src"], a["link_dst"], a["link_tid"], a["link_graph"], a["link_ac"]))
    return toks, links


CASES = [
    # words pdfs T seed beam lat_beam ac max_active min_active
    (6, 12, 8, 0, 30.0, 3.0, 1.0, 2 ** 31 - 1, 200),
    (40, 60, 40, 1, 8.0, 4.0, 0.5, 2 ** 31 - 1, 0),
    (200, 150, 60, 2, 13.0, 7.0, 0.1, 300, 200),      # max_active binds (flat scores at acoustic scale 0.1)
    (60, 90, 30, 3, 4.0, 2.0, 1.0, 10000, 40),        # narrow beams
    (300, 300, 120, 4, 10.0, 5.0, 0.3, 700, 200),
]


@pytest.mark.parametrize("pruning_pass", ["lds", "mixed", "global_memory"])
@pytest.mark.parametrize("case", CASES)
def test_decoder_matches_oracle(case, pruning_pass, monkeypatch):
    if pruning_pass != "lds":      # frames with more tokens than the LDS arrays hold are pruned in global memory
        monkeypatch.setenv("PK2_LAT_FIN_CAP", "0" if pruning_pass == "global_memory" else "60")
    nw, P, T, seed, beam, lb, ac, maxa, mina = case
    g, tm, ll, _ = _setup(nw, P, T, seed)
    want = lr.decode(_ref_graph(g), ll, tm["tid2pdf"], lr.DecoderOptionsRef(beam, lb, maxa, mina, 0.5, ac))
    rec = _recognizer(g, tm, beam, lb, ac, maxa, mina)
    lat = rec.decode(torch.from_numpy(ll).cuda())
    assert lat.status[0] == 0
    assert lat.best_cost[0] == np.float32(want.best_cost)
    got_t, got_l = _canon(lat.export(0))
    want_t, want_l = _canon(want.arrays())
    assert len(got_t) == len(want_t) and len(got_l) == len(want_l), (len(got_t), len(want_t), len(got_l), len(want_l))
    assert got_t == want_t          # bit-exact token costs
    assert got_l == want_l          # identical link set, bit-exact graph / acoustic costs


def _ali_near_lattice(rng, lat_ref, P):
    ali = synth.tid_alignment(rng, lat_ref.T, P)
    A = lat_ref.arrays()
    for l in range(A["link_src"].shape[0]):
        if A["link_tid"][l] != 0 and rng.random() < 0.05:
            ali[A["tok_frame"][A["link_src"][l]]] = A["link_tid"][l]
    return ali


@pytest.mark.parametrize("frame_values", ["lds", "mixed", "global_memory"])
@pytest.mark.parametrize("case", CASES[:3])
def test_mmi_matches_oracle(case, frame_values, monkeypatch):
    if frame_values != "lds":      # frames with more tokens than the LDS array holds are accumulated in global memory
        monkeypatch.setenv("PK2_LAT_FIN_CAP", "0" if frame_values == "global_memory" else "60")
    nw, P, T, seed, beam, lb, ac, maxa, mina = case
    g, tm, ll, rng = _setup(nw, P, T, seed)
    want = lr.decode(_ref_graph(g), ll, tm["tid2pdf"], lr.DecoderOptionsRef(beam, lb, maxa, mina, 0.5, ac))
    ali = _ali_near_lattice(rng, want, P)
    rec = _recognizer(g, tm, beam, lb, ac, maxa, mina)
    lat = rec.decode(torch.from_numpy(ll).cuda())
    for drop in (True, False):
        like, post = lat.mmi([ali], 1.0, 0.2, drop)
        wl, wp = lr.lattice_mmi(want, ali, tm["tid2pdf"], P, 1.0, 0.2, drop)
        assert abs(like.item() - wl) < 1e-9 * max(1.0, abs(wl))
        assert np.abs(post[0].cpu().numpy() - wp).max() < 2e-6


def _link_keys(a):
    fr, st = a["tok_frame"], a["tok_state"]
    return set((int(fr[s]), int(st[s]), int(st[d]), int(t)) for s, d, t in zip(a["link_src"], a["link_dst"], a["link_tid"]))


@pytest.mark.parametrize("order", ["ascending", "descending", "random"])
def test_device_lattice_against_kaldis_serial_cutoff_rule(order):
    """The decoder keeps an arc iff its cost is below the frame's FINAL cutoff; Kaldi's ProcessEmitting tightens
    next_cutoff while it walks its hash list (oracle/lattice_ref.py header).  The oracle-only test
    (test_oracle_lattice.py::test_final_cutoff_rule_...) measures the difference between the two rules; here the DEVICE
    lattice itself is held against the emulation of the serial rule at the configuration's beams (13 / 7, max_active
    binding on every frame): best-first order reproduces the device lattice exactly, any other order leaves the best path
    alone, adds at most 0.2 % links (all between the two cutoffs), and moves the device's MMI posteriors by <= 2e-4."""
    nw, P, T, seed, beam, lb, ac, maxa, mina = CASES[2]
    g, tm, ll, rng = _setup(nw, P, T, seed)
    graph = _ref_graph(g)
    opts = lr.DecoderOptionsRef(beam, lb, maxa, mina, 0.5, ac)
    rec = _recognizer(g, tm, beam, lb, ac, maxa, mina)
    lat = rec.decode(torch.from_numpy(ll).cuda())
    assert lat.status[0] == 0
    dev = lat.export(0)
    dev_links = _link_keys(dev)
    serial = lr.decode(graph, ll, tm["tid2pdf"], opts, serial_order=np.random.default_rng(5) if order == "random" else order)
    ser_links = _link_keys(serial.arrays())
    assert lat.best_cost[0] == np.float32(serial.best_cost)
    ali = _ali_near_lattice(rng, lr.decode(graph, ll, tm["tid2pdf"], opts), P)
    like, post = lat.mmi([ali], 1.0, 0.2, False)
    wl, wp = lr.lattice_mmi(serial, ali, tm["tid2pdf"], P, 1.0, 0.2, False)
    if order == "ascending":
        assert dev_links == ser_links
        assert np.abs(post[0].cpu().numpy() - wp).max() < 2e-6
    else:
        assert len(ser_links - dev_links) <= 0.002 * len(dev_links) and len(dev_links - ser_links) <= 0.002 * len(dev_links)
        assert np.abs(post[0].cpu().numpy() - wp).max() <= 2e-4
        assert abs(like.item() - wl) <= 1e-3 * max(1.0, abs(wl))


@pytest.mark.parametrize("criterion", ["smbr", "mpfe"])
def test_mpe_matches_oracle(criterion):
    nw, P, T, seed, beam, lb, ac, maxa, mina = CASES[1]
    g, tm, ll, rng = _setup(nw, P, T, seed)
    want = lr.decode(_ref_graph(g), ll, tm["tid2pdf"], lr.DecoderOptionsRef(beam, lb, maxa, mina, 0.5, ac))
    ali = _ali_near_lattice(rng, want, P)
    rec = _recognizer(g, tm, beam, lb, ac, maxa, mina)
    lat = rec.decode(torch.from_numpy(ll).cuda())
    for osc in (True, False):
        score, post = lat.mpe([ali], criterion, tm["silence_phones"], osc)
        ws, wp = lr.lattice_mpe(want, ali, tm["tid2pdf"], tm["tid2phone"], tm["silence_phones"], P, criterion, osc)
        assert abs(score.item() - ws) < 1e-9 * max(1.0, abs(ws))
        assert np.abs(post[0].cpu().numpy() - wp).max() < 2e-6 * max(1.0, np.abs(wp).max())


def test_ragged_batch_and_ops():
    P, beam, lb, ac = 60, 9.0, 4.0, 0.5
    g, tm, _, rng = _setup(50, P, 1, 7)
    lens = [23, 40, 11]
    lls = [(2.0 * rng.standard_normal((T, P))).astype(np.float32) for T in lens]
    x = torch.zeros(3, max(lens), P)
    for n, a in enumerate(lls):
        x[n, :lens[n]] = torch.from_numpy(a)
    x = x.cuda().requires_grad_()
    rec = _recognizer(g, tm, beam, lb, ac, min_active=0)
    tmod = rec.trans_model
    refs, alis = [], []
    for n in range(3):
        want = lr.decode(_ref_graph(g), lls[n], tm["tid2pdf"], lr.DecoderOptionsRef(beam, lb, 2 ** 31 - 1, 0, 0.5, ac))
        refs.append(want)
        alis.append(_ali_near_lattice(rng, want, P))
    # whole minibatch, MMI
    val = ops.LatticeBatchFunction.apply(x, lens, rec, tmod, alis, "mmi", None)
    val.backward()
    tot = 0.0
    for n in range(3):
        wl, wp = lr.lattice_mmi(refs[n], alis[n], tm["tid2pdf"], P, 1.0, 0.2, True)
        tot += wl
        assert np.abs(-x.grad[n, :lens[n]].cpu().numpy() - wp).max() < 2e-6
        assert x.grad[n, lens[n]:].abs().sum().item() == 0.0
    assert abs(val.item() - tot) < 1e-4 * abs(tot)
    # per-utterance operators with the reference's signatures
    for n in (0, 2):
        xn = torch.from_numpy(lls[n]).cuda().requires_grad_()
        v = ops.MMIFunction.apply(xn, rec, tmod, alis[n].tolist())
        v.backward()
        wl, wp = lr.lattice_mmi(refs[n], alis[n], tm["tid2pdf"], P, 1.0, 0.2, True)
        assert abs(v.item() - wl) < 1e-4 * abs(wl) and np.abs(-xn.grad.cpu().numpy() - wp).max() < 2e-6
        xn = torch.from_numpy(lls[n]).cuda().requires_grad_()
        v = ops.sMBRFunction.apply(xn, rec, tmod, alis[n].tolist(), "smbr", tm["silence_phones"])
        v.backward()
        ws, wp = lr.lattice_mpe(refs[n], alis[n], tm["tid2pdf"], tm["tid2phone"], tm["silence_phones"], P, "smbr", True)
        assert abs(v.item() - ws) < 1e-5 * max(1.0, abs(ws)) and np.abs(-xn.grad.cpu().numpy() - wp).max() < 2e-6


def test_pool_overflow_is_retried_with_larger_pools():
    nw, P, T, seed, beam, lb, ac, maxa, mina = CASES[1]
    g, tm, ll, _ = _setup(nw, P, T, seed)
    rec = _recognizer(g, tm, beam, lb, ac, maxa, mina)
    full = _canon(rec.decode(torch.from_numpy(ll).cuda()).export(0))
    rec.decoder_opts.tokens_per_frame, rec.decoder_opts.links_per_frame = 4, 4
    lat = rec.decode(torch.from_numpy(ll).cuda())
    assert lat.status[0] == 0 and _canon(lat.export(0)) == full


def _write_fst(path, g, kind):
    """OpenFst binary writer for the test (vector / const containers, tropical 'standard' arcs)."""
    S = g["num_states"]
    order = np.argsort(g["src"], kind="stable")
    src, dst, il, w = g["src"][order], g["dst"][order], g["ilabel"][order], g["weight"][order]
    counts = np.bincount(src, minlength=S)

    def s(x):
        return struct.pack("<i", len(x)) + x

    flags = 4 if kind == "const_aligned" else 0
    head = struct.pack("<i", 2125659606) + s(b"const" if kind.startswith("const") else b"vector") + s(b"standard")
    head += struct.pack("<iiQqqq", 2, flags, 0, g["start"], S, len(src))
    body = b""
    if kind == "vector":
        a = 0
        for st in range(S):
            body += struct.pack("<fq", float(g["final"][st]), int(counts[st]))
            for _ in range(counts[st]):
                body += struct.pack("<iifi", int(il[a]), 0, float(w[a]), int(dst[a])); a += 1
    else:
        def pad(b, pos):
            return b + b"\0" * ((16 - (pos + len(b)) % 16) % 16) if kind == "const_aligned" else b
        head = pad(head, 0)
        pos, states = 0, b""
        for st in range(S):
            nie = int((il[pos:pos + counts[st]] == 0).sum())
            states += struct.pack("<fIIII", float(g["final"][st]), pos, int(counts[st]), nie, int(counts[st]))
            pos += int(counts[st])
        states = pad(states, len(head))
        arcs = b"".join(struct.pack("<iifi", int(il[a]), 0, float(w[a]), int(dst[a])) for a in range(len(src)))
        body = states + arcs
    with open(path, "wb") as f:
        f.write(head + body)


@pytest.mark.parametrize("kind", ["vector", "const", "const_aligned"])
def test_hclg_from_openfst_file(tmp_path, kind):
    nw, P, T, seed, beam, lb, ac, maxa, mina = CASES[1]
    g, tm, ll, _ = _setup(nw, P, T, seed)
    path = str(tmp_path / "HCLG.fst")
    _write_fst(path, g, kind)
    graph = lattice.DecodeGraph(path)
    assert (graph.num_states, graph.num_arcs) == (g["num_states"], len(g["src"]))
    o = lattice.LatticeFasterDecoderOptions(beam=beam, lattice_beam=lb, max_active=maxa, min_active=mina)
    rec = lattice.MappedLatticeFasterRecognizer(lattice.TransitionModel.from_arrays(tm), graph, ac, o)
    a = _canon(rec.decode(torch.from_numpy(ll).cuda()).export(0))
    b = _canon(_recognizer(g, tm, beam, lb, ac, maxa, mina).decode(torch.from_numpy(ll).cuda()).export(0))
    assert a == b


def _custom_graph(seed, n_states=40, fan=200, num_pdfs=30, with_final=True):
    """A graph with a hub state that has `fan` emitting arcs and `fan` epsilon arcs (both beyond the decoder's
    per-thread degree limit) plus random sparse structure."""
    rng = np.random.default_rng(seed)
    ntid = 6 * (num_pdfs // 3)
    src, dst, il, w = [], [], [], []
    for k in range(fan):          # hub 0: heavy emitting and epsilon fans
        src.append(0); dst.append(1 + k % (n_states - 1)); il.append(1 + int(rng.integers(ntid))); w.append(float(rng.uniform(0.1, 3.0)))
        src.append(0); dst.append(1 + int(rng.integers(n_states - 1))); il.append(0); w.append(float(rng.uniform(0.5, 4.0)))
    for s in range(1, n_states):
        for _ in range(3):
            src.append(s); dst.append(int(rng.integers(n_states))); il.append(1 + int(rng.integers(ntid))); w.append(float(rng.uniform(0.1, 2.0)))
        src.append(s); dst.append(s); il.append(1 + int(rng.integers(ntid))); w.append(0.3)
        if s % 7 == 0:
            src.append(s); dst.append(0); il.append(0); w.append(0.2)      # back to the hub, no epsilon cycle (hub -> s is emitting or one-way)
    # remove epsilon arcs hub -> s for the states that have an epsilon arc back (keeps the epsilon graph acyclic)
    back = {s for s in range(1, n_states) if s % 7 == 0}
    keep = [i for i in range(len(src)) if not (src[i] == 0 and il[i] == 0 and dst[i] in back)]
    src, dst, il, w = ([x[i] for i in keep] for x in (src, dst, il, w))
    final = np.full(n_states, np.inf, np.float32)
    if with_final:
        final[0] = 0.5
        final[3] = 0.0
    return dict(num_states=n_states, start=0, src=np.asarray(src, np.int32), dst=np.asarray(dst, np.int32),
                ilabel=np.asarray(il, np.int32), weight=np.asarray(w, np.float32), final=final)


@pytest.mark.parametrize("with_final", [True, False])
@pytest.mark.parametrize("T", [1, 17])
def test_heavy_states_short_utterances_and_missing_finals(with_final, T):
    P = 30
    g = _custom_graph(5, with_final=with_final)
    tm = synth.transition_model_arrays(P)
    rng = np.random.default_rng(T)
    ll = (1.5 * rng.standard_normal((T, P))).astype(np.float32)
    want = lr.decode(_ref_graph(g), ll, tm["tid2pdf"], lr.DecoderOptionsRef(7.0, 3.0, 2 ** 31 - 1, 0, 0.5, 0.7))
    rec = _recognizer(g, tm, 7.0, 3.0, 0.7, min_active=0)
    lat = rec.decode(torch.from_numpy(ll).cuda())
    assert lat.best_cost[0] == np.float32(want.best_cost)
    assert _canon(lat.export(0)) == _canon(want.arrays())
    ali = _ali_near_lattice(rng, want, P)
    like, post = lat.mmi([ali], 1.0, 0.2, True)
    wl, wp = lr.lattice_mmi(want, ali, tm["tid2pdf"], P, 1.0, 0.2, True)
    assert abs(like.item() - wl) < 1e-9 * max(1.0, abs(wl)) and np.abs(post[0].cpu().numpy() - wp).max() < 2e-6


def test_decoder_failures_are_reported():
    from pykaldi2_amd import _lib
    P = 30
    g, tm, ll, _ = _setup(20, P, 12, 3)
    # a beam below the cheapest word entry: the start token has nowhere to go
    rec = _recognizer(g, tm, 0.5, 0.25, 1.0, min_active=0)
    with pytest.raises(_lib.Pk2Error, match="no surviving token"):
        rec.decode(torch.from_numpy(ll).cuda())
    # HCLG with transition-ids the model does not have
    g2 = dict(g); g2["ilabel"] = g["ilabel"].copy(); g2["ilabel"][g2["ilabel"] > 0] += 1000
    with pytest.raises(AssertionError):
        _recognizer(g2, tm, 8.0, 4.0, 1.0)
    # more pdfs than the LDS row holds
    big = torch.zeros(4, 9000, device="cuda")
    rec = _recognizer(g, tm, 8.0, 4.0, 1.0)
    with pytest.raises(_lib.Pk2Error, match="LDS row limit"):
        rec.decode(big)


def test_one_workgroup_decoder_variant():
    """The decoder variant selected by PK2_LAT_DECODER=wg (one workgroup per utterance, one launch; also the fallback
    of the default launch-per-frame decoder) passes the same oracle comparisons.  The choice is read once per
    process, hence the child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_lattice.py"), "-q", "-m", "gpu",
                          "-k", "matches_oracle or ragged or heavy or overflow"], env=dict(os.environ, PK2_LAT_DECODER="wg"),
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "failed" not in out.stdout


@pytest.mark.parametrize("env", [dict(PK2_LAT_DECODER="frames"), dict(PK2_LAT_TEAM="8"), dict(PK2_LAT_TEAM="32"),
                                 dict(PK2_LAT_FIELD_BITS="3")],
                         ids=["launch_per_frame", "persistent_team8", "persistent_team32", "persistent_spilled_counts"])
def test_team_decoder_variants(env):
    """The default decoder runs all frames of an utterance inside one persistent launch (a team of 16 workgroups of one
    XCD); the launch-per-frame decoder (its fallback) and the other team sizes pass the same oracle comparisons in their own
    processes (the choice is read once per process).  PK2_LAT_FIELD_BITS=3: the counts that travel inside the team barriers'
    release words (links, tokens, epsilon entries of a frame) get 3-bit fields, so nearly every one of them takes the spill
    slot of the frame record instead -- the path a frame with more than 2^23 links would take."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_lattice.py"), "-q", "-m", "gpu",
                          "-k", "matches_oracle or ragged or heavy or overflow or persistent_decoder_is_in_use or take_turns"],
                         env=dict(os.environ, **env), capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "failed" not in out.stdout


def test_more_utterances_than_teams_take_turns():
    """21 ragged utterances against the oracle, one by one: with a team of 16 workgroups per utterance the persistent decoder
    has 2 teams per XCD = 16 teams, so five teams decode a second utterance from the queue (PK2_LAT_TEAM=32: 8 teams, up to
    three each); a word-loop graph, so every frame has a heavy token dealt to the whole team."""
    P, beam, lb, ac = 80, 9.0, 4.0, 0.5
    g, tm, _, rng = _setup(70, P, 1, 21)
    lens = [int(v) for v in rng.integers(5, 60, size=21)]
    lls = [(2.0 * rng.standard_normal((T, P))).astype(np.float32) for T in lens]
    x = torch.zeros(len(lens), max(lens), P)
    for n, a in enumerate(lls):
        x[n, :lens[n]] = torch.from_numpy(a)
    rec = _recognizer(g, tm, beam, lb, ac)
    lat = rec.decode_batch(x.cuda(), lens)
    assert (lat.status == 0).all()
    for n, a in enumerate(lls):
        want = lr.decode(_ref_graph(g), a, tm["tid2pdf"], lr.DecoderOptionsRef(beam, lb, 2 ** 31 - 1, 200, 0.5, ac))
        assert lat.best_cost[n] == np.float32(want.best_cost), n
        got, ref = _canon(lat.export(n)), _canon(want.arrays())
        assert got[0] == ref[0] and got[1] == ref[1], n


def test_persistent_decoder_is_in_use():
    """After a decode on an MI355X the library reports which decoder ran: the persistent one unless PK2_LAT_DECODER says
    otherwise, and never a time-out."""
    import os
    g = synth.decoding_graph_arcs(30, 40, seed=3)
    tm = synth.transition_model_arrays(40)
    rec = _recognizer(g, tm, 10.0, 5.0, 1.0)
    rng = np.random.default_rng(5)
    lat = rec.decode_batch(torch.from_numpy(rng.standard_normal((3, 25, 40)).astype(np.float32)).cuda(), [25, 11, 18])
    assert (lat.status == 0).all()
    state, abort = lattice.persistent_decoder_status()
    assert abort == 0
    if os.environ.get("PK2_LAT_DECODER") in ("frames", "wg"):
        assert state in (-1, 0)
    else:
        assert state == 1


STALL_WORKER = r'''
import os, sys, time, numpy as np, torch
sys.path.insert(0, %(root)r)
from pykaldi2_amd import _lib, lattice, synth
g = synth.decoding_graph_arcs(30, 40, seed=3)
tm = synth.transition_model_arrays(40)
o = lattice.LatticeFasterDecoderOptions(beam=10.0, lattice_beam=5.0, max_active=2 ** 31 - 1, min_active=200)
rec = lattice.MappedLatticeFasterRecognizer(lattice.TransitionModel.from_arrays(tm), g, 1.0, o)
x = torch.from_numpy(np.random.default_rng(5).standard_normal((3, 25, 40)).astype(np.float32)).cuda()
lens = [25, 11, 18]
def canon(lat, n):
    a = lat.export(n)
    return sorted(zip(a["tok_frame"].tolist(), a["tok_state"].tolist(), a["tok_cost"].tolist())), len(a["link_src"])
lat = rec.decode_batch(x, lens)            # (the first launch of a process is verified on the host)
lat = rec.decode_batch(x, lens)
assert (lat.status == 0).all()
want = [canon(lat, n) for n in range(3)]
state, abort = lattice.persistent_decoder_status()
assert state == 1 and abort == 0 and not _lib.persist_guard_raised()
os.environ["PK2_LAT_TEST_STALL"] = "1"
t0 = time.time()
try:
    rec.decode_batch(x, lens)
    raised = False
except RuntimeError as e:
    raised = True
    msg = str(e)
dt = time.time() - t0
del os.environ["PK2_LAT_TEST_STALL"]
state, abort = lattice.persistent_decoder_status()
assert raised, "a decode whose team lost a workgroup must not return a lattice"
assert abort != 0 and _lib.persist_guard_raised(), (state, abort)
assert 0.9 < dt < 20.0, dt                 # the polls give up after 1 s; nobody hangs
_lib.check(_lib.lib().pk2_persist_guard_clear())
lat = rec.decode_batch(x, lens)
assert (lat.status == 0).all() and [canon(lat, n) for n in range(3)] == want
print("STALL_OK %%.2f s: %%s" %% (dt, msg[:120]))
'''


def test_persistent_decoder_gives_up_when_a_team_loses_a_workgroup(tmp_path):
    """The persistent decoder's failure path, in a process of its own (the time-out flag is sticky): PK2_LAT_TEST_STALL=1 makes
    one workgroup of every team leave in frame 2.  Its team mates must stop polling after the 1 s time-out (barrier release
    words, the cutoff word), the check kernel behind the launch must mark every utterance "not decoded" and raise the
    per-device guard, the call must raise instead of returning a lattice -- and the next call, guard cleared, must decode
    the same lattices as before."""
    import os
    import subprocess
    import sys
    if os.environ.get("PK2_LAT_DECODER") in ("frames", "wg"):
        pytest.skip("the persistent decoder is switched off")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "stall.py"
    script.write_text(STALL_WORKER % dict(root=root))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "STALL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


FULL_SIZE_WORKER = r'''
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from pykaldi2_amd import lattice, synth
P = 5768
g = synth.decoding_graph_arcs(20000, P, seed=0)
tm = synth.transition_model_arrays(P)
o = lattice.LatticeFasterDecoderOptions(beam=13.0, lattice_beam=7.0, max_active=7000, min_active=200)
rec = lattice.MappedLatticeFasterRecognizer(lattice.TransitionModel.from_arrays(tm), g, 0.1, o)
ll = torch.from_numpy(np.load(%(ll)r)).cuda()
lat = rec.decode_batch(ll, [150, 97])
out = {}
for n in range(2):
    for k, v in lat.export(n).items():
        out["%%s%%d" %% (k, n)] = v
np.savez(%(out)r, status=lat.status, best=lat.best_cost, **out)
'''


def test_full_size_lattice_properties(tmp_path):
    """The configuration bench.py --se times (BASELINE configs[3]; reference bin/train_se.py:173-181 decoder options):
    P = 5768, 230k-state / 460k-arc word-loop HCLG, beam 13, lattice-beam 7, max_active 7000 (binding: random-init
    scores are flat at acoustic scale 0.1), two utterances.  No oracle at this size -- properties instead:
    (1) the launch-per-frame decoder (default) and the one-workgroup decoder (PK2_LAT_DECODER=wg, own process) keep
        bit-identical token costs and link sets;
    (2) token costs are the best forward costs of the kept links, and EVERY kept link lies on a complete path within
        lattice_beam of the best one (Kaldi's PruneForwardLinks guarantee);
    (3) MMI posteriors: numerator - denominator sums to 0 on every frame (drop_frames off), the denominator mass of a
        frame is <= 1, padding frames stay 0."""
    import os
    import subprocess
    import sys
    P, lens, ac_scale, lat_beam = 5768, [150, 97], 0.1, 7.0
    rng = np.random.default_rng(11)
    ll = (2.0 * rng.standard_normal((2, 150, P))).astype(np.float32)
    g = synth.decoding_graph_arcs(20000, P, seed=0)
    assert g["num_states"] >= 200000
    tm = synth.transition_model_arrays(P)
    rec = _recognizer(g, tm, 13.0, lat_beam, ac_scale, 7000, 200)
    lat = rec.decode_batch(torch.from_numpy(ll).cuda(), lens)
    assert (lat.status == 0).all()
    exp = [lat.export(n) for n in range(2)]
    # (1) the other decoder, in its own process
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    np.save(tmp_path / "ll.npy", ll)
    script = tmp_path / "w.py"
    script.write_text(FULL_SIZE_WORKER % dict(root=root, ll=str(tmp_path / "ll.npy"), out=str(tmp_path / "wg.npz")))
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, PK2_LAT_DECODER="wg"), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    wg = np.load(tmp_path / "wg.npz")
    assert (wg["status"] == 0).all() and np.array_equal(wg["best"], lat.best_cost)
    for n in range(2):
        other = {k: wg["%s%d" % (k, n)] for k in exp[n]}
        a, b = _canon(exp[n]), _canon(other)
        assert a[0] == b[0] and a[1] == b[1], n
    # (2) forward costs and the lattice-beam property, on the exported arrays
    for n, T in enumerate(lens):
        A = exp[n]
        fr, src, dst = A["tok_frame"], A["link_src"], A["link_dst"]
        cost = A["link_graph"].astype(np.float64) + ac_scale * A["link_ac"].astype(np.float64)
        eps = A["link_tid"] == 0
        assert fr.max() == T and (fr[dst] - fr[src] == np.where(eps, 0, 1)).all()
        ntok = fr.shape[0]
        alpha = np.full(ntok, np.inf); alpha[(fr == 0) & (A["tok_cost"] == 0)] = 0.0
        beta = np.full(ntok, np.inf)
        last = fr == T
        fin = A["tok_final"].astype(np.float64)
        beta[last] = fin[last] if np.isfinite(fin[last]).any() else 0.0
        by_src = [np.flatnonzero(fr[src] == t) for t in range(T + 1)]
        for t in range(T + 1):                   # forward: epsilon closure of frame t, then the emitting links t -> t+1
            e, m = by_src[t][eps[by_src[t]]], by_src[t][~eps[by_src[t]]]
            for _ in range(64):
                new = alpha.copy(); np.minimum.at(new, dst[e], alpha[src[e]] + cost[e])
                if np.array_equal(new, alpha):
                    break
                alpha = new
            np.minimum.at(alpha, dst[m], alpha[src[m]] + cost[m])
        for t in range(T, -1, -1):               # backward
            e, m = by_src[t][eps[by_src[t]]], by_src[t][~eps[by_src[t]]]
            np.minimum.at(beta, src[m], beta[dst[m]] + cost[m])
            for _ in range(64):
                new = beta.copy(); np.minimum.at(new, src[e], beta[dst[e]] + cost[e])
                if np.array_equal(new, beta):
                    break
                beta = new
        assert np.abs(alpha - A["tok_cost"]).max() < 2e-2 * (1 + T / 100), np.abs(alpha - A["tok_cost"]).max()
        best = (alpha[last] + beta[last]).min()
        assert abs(best - float(lat.best_cost[n])) < 2e-2
        through = alpha[src] + cost + beta[dst]
        assert through.min() <= best + 1e-6 and through.max() <= best + lat_beam + 5e-2, (through.max() - best)
        assert np.isfinite(alpha).all() and np.isfinite(beta).all()      # every kept token lies on a complete path
        assert len(src) / T > 1000                                        # a real lattice, not a single path
    # (3) MMI posteriors
    ref = [synth.tid_alignment(rng, T, P) for T in lens]
    like, post = lat.mmi(ref, 1.0, 0.2, drop_frames=False)
    post = post.cpu().numpy()
    assert np.isfinite(like.cpu().numpy()).all()
    for n, T in enumerate(lens):
        assert np.abs(post[n, :T].sum(1)).max() < 1e-4
        den = -np.minimum(post[n, :T], 0).sum(1)
        assert den.max() <= 1 + 1e-4 and den.min() > 0.5
        assert not post[n, T:].any()
