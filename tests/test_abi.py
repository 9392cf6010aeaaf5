"""The C-ABI library loads and exports exactly what include/pk2hip.h declares; host-side graph
preprocessing (no GPU needed) is checked against the oracle."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

from oracle import chain_ref as R
from pykaldi2_amd import _lib, chain, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "pk2hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pk2_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), name
    # and the ctypes signature table covers the header (plus the debug hook)
    assert set(declared) <= set(_lib.SIGNATURES), set(declared) - set(_lib.SIGNATURES)
    assert _lib.lib().pk2_version() == 1


def test_bad_arguments_fail_loudly():
    h = C.c_void_p()
    rc = _lib.lib().pk2_den_graph_create(0, 5, 0, None, None, None, None, 0, C.byref(h))
    assert rc < 0 and b"null" in _lib.lib().pk2_last_error().lower()
    with pytest.raises(_lib.Pk2Error):
        _lib.check(rc)


def _emulate(o, nrows_total, val):
    """numpy model of the kernels' lane-run segmented reduction over one ordering."""
    out = np.zeros(nrows_total)
    arcs, meta, K = o["arcs"], o["meta"], o["arcs_per_lane"]
    f32 = lambda col: arcs[:, col].copy().view(np.float32).astype(np.float64)
    v = val(arcs[:, 0], arcs[:, 1], f32(2), f32(3))
    for ch in range(len(o["row0"])):
        acc = np.zeros(o["nrows"][ch])
        for wb in range(o["wb_off"][ch], o["wb_off"][ch + 1]):
            for lane in range(64):
                c, mask = int(meta[wb * 64 + lane, 0]), int(meta[wb * 64 + lane, 1])
                s = 0.0
                for j in range(K):
                    s += v[(wb * K + j) * 64 + lane]
                    if (mask >> j) & 1:
                        acc[c] += s; s = 0.0; c += 1
                if s != 0.0:      # state-x orderings: the open tail of a lane belongs to the row that ends in a later lane
                    acc[c] += s
        out[o["row0"][ch]:o["row0"][ch] + len(acc)] += acc
    return out


@pytest.mark.parametrize("case", ["small", "bigrow", "sparse"])
def test_den_graph_orderings_reproduce_segment_sums(case, monkeypatch):
    monkeypatch.setenv("PK2_DEN_ORDER", "none")     # the layouts below are compared in the caller's state numbering
    S, A, P, seed = dict(small=(40, 300, 7, 1), bigrow=(200, 20000, 11, 2), sparse=(3000, 9000, 50, 3))[case]
    g = synth.den_graph_arcs(S, A, P, seed)
    if case == "bigrow":  # one state with > 4096 incoming arcs (split chunk) and a pdf without arcs
        g["dst"][:6000] = 5
        g["pdf"][g["pdf"] == 3] = 4
    G = chain.DenominatorGraph(g, P)
    ref = R.DenGraphRef(g["num_states"], g["src"], g["dst"], g["pdf"], g["prob"], 0, P)
    pi = G.initial_probs()
    assert np.abs(pi - ref.initial_probs).max() < 1e-7 and abs(pi.sum() - 1) < 1e-5
    rng = np.random.default_rng(seed)
    S_ = g["num_states"]
    al, be, x = rng.random(S_), rng.random(S_), rng.random(P)
    src, dst, pdf, prob = g["src"], g["dst"], g["pdf"], g["prob"].astype(np.float64)
    o = G.debug_ordering(0)
    if case == "bigrow":
        assert o["atomic"].sum() >= 2
    got = _emulate(o, S_, lambda a, b, p, pp: al[a] * p * x[b])
    assert np.abs(got - np.bincount(dst, weights=al[src] * prob * x[pdf], minlength=S_)).max() < 1e-9
    got = _emulate(o, S_, lambda a, b, p, pp: pp)
    assert np.abs(got - np.bincount(dst, weights=pi[src].astype(np.float64) * prob, minlength=S_)).max() < 1e-7
    got = _emulate(G.debug_ordering(1), S_, lambda a, b, p, pp: be[a] * p * x[b])
    assert np.abs(got - np.bincount(src, weights=be[dst] * prob * x[pdf], minlength=S_)).max() < 1e-9
    got = _emulate(G.debug_ordering(2), P, lambda a, b, p, pp: al[a] * p * be[b])
    assert np.abs(got - np.bincount(pdf, weights=al[src] * prob * be[dst], minlength=P)).max() < 1e-9


@pytest.mark.parametrize("case", ["unique", "chain_topology", "multi_entry", "bigstate", "arc_pdf", "no_peel"])
def test_den_graph_virtual_state_orderings(case, monkeypatch):
    """The state-x kernels' decomposition (chain_internal.h): one self-loop per state is peeled into a per-state term;
    virtual states = distinct (dst, pdf) pairs of the remaining arcs; the forward ordering's rows are virtual states
    with the rows of one state inside one chunk, the backward ordering gathers by virtual destination.  A numpy model
    of the kernels' arithmetic over these tables reproduces alpha[d] = sum_arcs alpha[src] prob x[pdf] and
    beta[s] = sum_arcs prob x[pdf] beta[dst] for ANY graph."""
    S, A, P, seed = 300, 6000, 23, 7
    kw = dict(unique={}, chain_topology=dict(loop_pdf_differs=True), multi_entry=dict(loop_pdf_differs=True, multi_entry_frac=0.3),
              bigstate=dict(loop_pdf_differs=True), arc_pdf={}, no_peel=dict(loop_pdf_differs=True))[case]
    monkeypatch.setenv("PK2_DEN_ORDER", "none")     # the layouts below are compared in the caller's state numbering
    if case == "no_peel":
        monkeypatch.setenv("PK2_DEN_PEEL", "0")
    if case == "bigstate":
        S, A = 200, 20000
    g = synth.den_graph_arcs(S, A, P, seed, **kw)
    rng = np.random.default_rng(seed)
    if case == "bigstate":   # state 5: > 4096 entering arcs over three pdfs, one of the rows alone > 4096
        g["dst"][:9000] = 5
        g["pdf"][:5000] = 1; g["pdf"][5000:8000] = 2; g["pdf"][8000:9000] = 3
    if case == "arc_pdf":
        g["pdf"] = rng.integers(0, P, size=g["pdf"].shape[0]).astype(np.int32)
    G = chain.DenominatorGraph(g, P)
    S_ = g["num_states"]
    src, dst, pdf, prob = g["src"], g["dst"], g["pdf"], g["prob"].astype(np.float64)
    of, ob = G.debug_ordering(3), G.debug_ordering(4)
    voff, vpdf, lpdf, lprob = of["voff"], of["vpdf"], of["loop_pdf"], of["loop_prob"].astype(np.float64)
    V = vpdf.shape[0]
    # the peeled arcs: the first self-loop of each state
    peeled = np.zeros(src.shape[0], bool)
    seen = set()
    if case != "no_peel":
        for i in np.flatnonzero(src == dst):
            if int(src[i]) not in seen:
                seen.add(int(src[i])); peeled[i] = True
                assert lpdf[src[i]] == pdf[i] and abs(lprob[src[i]] - prob[i]) < 1e-7
    assert (lpdf >= 0).sum() == len(seen) and not lprob[lpdf < 0].any()
    if case in ("chain_topology", "unique"):
        assert len(seen) == S_          # every state of the synthetic graph loops
    keep = ~peeled
    # virtual states are the distinct (dst, pdf) pairs of the kept arcs in order, one pdf -1 entry for states nobody enters
    ks, kd, kp, kprob = src[keep], dst[keep], pdf[keep], prob[keep]
    pairs = sorted(set(zip(kd.tolist(), kp.tolist())) | {(d, -1) for d in set(range(S_)) - set(kd.tolist())})
    assert V == len(pairs) and voff[0] == 0 and voff[-1] == V
    vstate = np.repeat(np.arange(S_), np.diff(voff))
    assert [(int(a), int(b)) for a, b in zip(vstate, vpdf)] == pairs
    if case in ("unique", "chain_topology"):
        assert V == S_                  # the arcs entering a state from elsewhere carry one (forward) pdf
    if case == "no_peel":
        assert V > 1.9 * S_
    # occupancy states: the virtual states of a state, then its peeled loop
    want_opdf = []
    for d in range(S_):
        want_opdf += vpdf[voff[d]:voff[d + 1]].tolist() + ([int(lpdf[d])] if lpdf[d] >= 0 else [])
        assert of["ooff"][d + 1] == len(want_opdf)
    assert of["opdf"].tolist() == want_opdf
    al, be, x = rng.random(S_), rng.random(S_), rng.random(P)
    pi = G.initial_probs().astype(np.float64)
    xv = np.where(vpdf >= 0, x[np.maximum(vpdf, 0)], 1.0)
    xl = np.where(lpdf >= 0, x[np.maximum(lpdf, 0)], 1.0)
    # forward: row sums over virtual rows, then x per virtual row, the sum over the rows of a state, the loop term
    rows = _emulate(of, V, lambda a, b, p, pp: al[a] * p)
    want_rows = np.zeros(V)
    key = {pr: i for i, pr in enumerate(pairs)}
    arc_v = np.array([key[(int(d), int(q))] for d, q in zip(kd, kp)])
    np.add.at(want_rows, arc_v, al[ks] * kprob)
    assert np.abs(rows - want_rows).max() < 1e-9
    got = np.bincount(vstate, weights=rows * xv, minlength=S_) + al * lprob * xl
    assert np.abs(got - np.bincount(dst, weights=al[src] * prob * x[pdf], minlength=S_)).max() < 1e-9
    # the per-(chunk, row) leaky term
    leak = np.zeros(V)
    for ch in range(len(of["row0"])):
        n = of["nrows"][ch]
        leak[of["row0"][ch]:of["row0"][ch] + n] += of["row_leak"][of["slot0"][ch]:of["slot0"][ch] + n]
    want_leak = np.zeros(V)
    np.add.at(want_leak, arc_v, pi[ks] * kprob)
    assert np.abs(leak - want_leak).max() < 1e-6
    # chunk invariants: the rows of a state share a chunk unless the chunk is a single-row atomic one; exactly one
    # piece of a state / row that is split carries the flag 2 (it adds the per-state loop term)
    first = {}
    for ch in range(len(of["row0"])):
        r0, n, at = of["row0"][ch], of["nrows"][ch], of["atomic"][ch]
        assert of["real0"][ch] == vstate[r0] and of["nreal"][ch] == vstate[r0 + n - 1] - vstate[r0] + 1
        if at:
            assert n == 1
            first[int(vstate[r0])] = first.get(int(vstate[r0]), 0) + (at == 2)
        else:
            assert r0 == voff[vstate[r0]] and r0 + n == voff[vstate[r0 + n - 1] + 1]
    assert all(c == 1 for c in first.values())
    if case == "bigstate":
        assert of["atomic"].astype(bool).sum() >= 4 and 5 in first
    # backward: gather index = virtual destination
    bev = be[vstate] * xv
    got = _emulate(ob, S_, lambda a, b, p, pp: bev[a] * p) + lprob * xl * be
    assert np.abs(got - np.bincount(src, weights=be[dst] * prob * x[pdf], minlength=S_)).max() < 1e-9


def _emulate_persist(lay, table):
    """numpy model of arc_rows + row_sum of chain_den_persist.hip: the value of every row of an ordering."""
    R_, K, T, _ = lay["arcs"].shape
    W = T // 64
    out = np.zeros(lay["row_begin"][-1])
    for r in range(R_):
        row0, row1 = lay["row_begin"][r], lay["row_begin"][r + 1]
        acc = np.full(max(row1 - row0, 1), np.nan)
        idx = lay["arcs"][r, :, :, 0]
        prob = lay["arcs"][r, :, :, 1].copy().view(np.float32).astype(np.float64)
        tails, has_end = np.zeros(T), np.zeros(T, bool)
        for tid in range(T):
            e, c, s = int(lay["ends"][r, tid]), int(lay["first_row"][r, tid]), 0.0
            assert all((j + 1) % lay["estep"] == 0 for j in range(K) if (e >> j) & 1)     # rows end only at the aligned slots
            for j in range(K):
                s += table[idx[j, tid]] * prob[j, tid]
                if (e >> j) & 1:
                    acc[c] = s; c += 1; s = 0.0
            tails[tid], has_end[tid] = s, e != 0
        wcarry = np.zeros(W)
        for w in range(W):          # segmented scan over the lanes of a wave: the open tails reach the next row end
            x = 0.0
            for lane in range(64):
                tid = w * 64 + lane
                if has_end[tid]:
                    acc[lay["first_row"][r, tid]] += x
                    x = tails[tid]
                else:
                    x += tails[tid]
            wcarry[w] = x
        for k in range(W):          # what is still open at the end of a wave belongs to row wcrow
            if lay["wcrow"][r, k] >= 0:
                acc[lay["wcrow"][r, k]] += wcarry[k]
        out[row0:row1] = acc[:row1 - row0]
    return out


@pytest.mark.parametrize("case", ["chain_topology", "multi_entry", "bigstate", "no_peel"])
@pytest.mark.parametrize("estep", [0, 1, 4])
def test_den_graph_persistent_layouts(case, estep, monkeypatch):
    """The persistent kernel's layouts (chain_internal.h: HostPersist): 32 workgroups x 512 threads x 64 register-resident
    arc slots, rows ending only at aligned slots, wave carries.  A numpy model of the kernel's row sums over these tables
    reproduces alpha[d] = sum_arcs alpha[src] prob x[pdf] and beta[s] = sum_arcs prob x[pdf] beta[dst] -- including a state
    whose 30000 entering arcs span several waves."""
    S, A, P, seed = 300, 6000, 23, 7
    kw = dict(chain_topology=dict(loop_pdf_differs=True), multi_entry=dict(loop_pdf_differs=True, multi_entry_frac=0.3),
              bigstate=dict(loop_pdf_differs=True), no_peel=dict(loop_pdf_differs=True))[case]
    monkeypatch.setenv("PK2_DEN_ORDER", "none")
    if estep:
        monkeypatch.setenv("PK2_DEN_ESTEP", str(estep))
    if case == "no_peel":
        monkeypatch.setenv("PK2_DEN_PEEL", "0")
    if case == "bigstate":
        S, A = 200, 40000
    g = synth.den_graph_arcs(S, A, P, seed, **kw)
    if case == "bigstate":
        g["dst"][:30000] = 5
        g["pdf"][:20000] = 1; g["pdf"][20000:28000] = 2; g["pdf"][28000:30000] = 3
    G = chain.DenominatorGraph(g, P)
    of = G.debug_ordering(3)
    voff, vpdf, lpdf, lprob = of["voff"], of["vpdf"], of["loop_pdf"], of["loop_prob"].astype(np.float64)
    S_, V = g["num_states"], len(of["vpdf"])
    vstate = np.repeat(np.arange(S_), np.diff(voff))
    rng = np.random.default_rng(1)
    al, be, x = rng.random(S_), rng.random(S_), rng.random(P)
    xv = np.where(vpdf >= 0, x[np.maximum(vpdf, 0)], 1.0)
    xl = np.where(lpdf >= 0, x[np.maximum(lpdf, 0)], 1.0)
    src, dst, pdf, prob = g["src"], g["dst"], g["pdf"], g["prob"].astype(np.float64)
    pf, pb = G.debug_persist(0), G.debug_persist(1)
    assert pf is not None and pb is not None
    if estep:
        assert pf["estep"] == estep and pb["estep"] == estep
    else:
        assert pf["estep"] == 8       # small graphs leave room for the widest alignment
    assert pf["grp_begin"][-1] == S_ and pf["row_begin"][-1] == V and pb["row_begin"][-1] == S_
    assert (np.diff(pf["grp_begin"]) >= 0).all() and (voff[pf["grp_begin"]] == pf["row_begin"]).all()   # whole states per workgroup
    rows = _emulate_persist(pf, al)
    got = np.bincount(vstate, weights=rows * xv, minlength=S_) + al * lprob * xl
    assert np.abs(got - np.bincount(dst, weights=al[src] * prob * x[pdf], minlength=S_)).max() < 1e-9
    pi = G.initial_probs().astype(np.float64)
    keep = np.ones(len(src), bool)
    if case != "no_peel":
        seen = set()
        for i in np.flatnonzero(src == dst):
            if int(src[i]) not in seen:
                seen.add(int(src[i])); keep[i] = False
    want_leak = np.zeros(S_)
    np.add.at(want_leak, dst[keep], pi[src[keep]] * prob[keep])
    assert np.abs(np.bincount(vstate, weights=pf["row_leak"], minlength=S_) - want_leak).max() < 1e-6
    want_psum = np.zeros(S_)
    np.add.at(want_psum, dst[keep], prob[keep])
    assert np.abs(np.bincount(vstate, weights=pf["row_psum"], minlength=S_) - want_psum).max() < 1e-5
    # the backward kernel predicts a frame's sums over the states from the vector it gathers (one exchange per frame):
    # sum_s pi[s] b'[s] = sum_v w[v] row_leak[v] + the peeled loops, sum_s b'[s] = sum_v w[v] row_psum[v] + loops
    w = be[vstate] * xv
    b_new = _emulate_persist(pb, w) + lprob * xl * be
    assert abs((pi * b_new).sum() - ((w * pf["row_leak"]).sum() + (pi * lprob * xl * be).sum())) < 1e-6 * max(1.0, (pi * b_new).sum())
    assert abs(b_new.sum() - ((w * pf["row_psum"]).sum() + (lprob * xl * be).sum())) < 1e-5 * max(1.0, b_new.sum())
    got = _emulate_persist(pb, be[vstate] * xv) + lprob * xl * be
    assert np.abs(got - np.bincount(src, weights=be[dst] * prob * x[pdf], minlength=S_)).max() < 1e-9


def _emulate_persist2(lay, table):
    """numpy model of frame_rows + row_val of chain_den_persist2.hip over the host layout (chain_internal.h: HostPersist2):
    chunked LDS table (entries never copied are NaN: a null slot must not touch them), pass A / pass B with plain stores,
    streamed segments adding into pass A's array, segmented wave scans, wave carries of every segment."""
    R_, K, T = lay["prob"].shape
    Q, SP, W, NC = K // 2, lay["SP"], T // 64, lay["K"]
    cbeg, off = lay["cbeg"], lay["lds_off"]
    out = np.zeros(lay["row_begin"][-1])
    table = np.asarray(table, np.float64)
    assert len(table) == lay["R"] and cbeg[NC] == lay["R"] and cbeg[0] == 0
    idx_of = lambda words, j, tid: (int(words[j // 2, tid]) >> (16 * (j & 1))) & 0xffff

    def scan(acc, tails, has_end, first_row, wcrow_seg, add):
        """open tails of a segment -> the next row end of the wave, or the wave's carry"""
        for w in range(W):
            x = 0.0
            for lane in range(64):
                tid = w * 64 + lane
                if has_end[tid]:
                    acc[first_row[tid]] += x
                    x = tails[tid]
                else:
                    x += tails[tid]
            add.append((int(wcrow_seg[w]), x))

    for r in range(R_):
        row0, row1 = int(lay["row_begin"][r]), int(lay["row_begin"][r + 1])
        n = row1 - row0
        lds = np.full(lay["tfloats"], np.nan)

        def load(c):
            lds[off[c]:off[c] + cbeg[c + 1] - cbeg[c]] = table[cbeg[c]:cbeg[c + 1]]
        nc = lay["ncomp"][r]
        acc = [np.full(max(int(nc[0]), 1), np.nan), np.full(max(int(nc[1]), 1), np.nan)]     # compact rows of list 0 / 1
        for ps in (0, 1):
            acc[ps][lay["uncovered"][r, ps]:] = 0.0
            m = lay["rmap"][ps, row0:row1]
            assert sorted(m[m >= 0]) == list(range(int(nc[ps])))        # the present rows, numbered in order
        accS = np.zeros(max(n, 1))                                       # rank-local rows
        carries = []
        pb = lay["pbeg"][r]

        def resident(ps):
            tails, has_end = np.zeros(T), np.zeros(T, bool)
            for tid in range(T):
                e, c, sm = int(lay["ends"][r, ps, tid]), int(lay["first_row"][r, ps, tid]), 0.0
                assert all((j + 1) % lay["estep"] == 0 for j in range(Q) if (e >> j) & 1)
                for j in range(Q):
                    sm += lds[idx_of(lay["idx2"][r], ps * Q + j, tid)] * float(lay["prob"][r, ps * Q + j, tid])
                    if (e >> j) & 1:
                        acc[ps][c] = sm; c += 1; sm = 0.0
                tails[tid], has_end[tid] = sm, e != 0
            scan(acc[ps], tails, has_end, lay["first_row"][r, ps], lay["wcrow"][r, ps], carries)

        def streamed(c):
            if pb[c + 1] == pb[c]:
                assert (lay["wcrow"][r, 2 + c] == -1).all()
                return
            tails, has_end = np.zeros(T), np.zeros(T, bool)
            for tid in range(T):
                cc, sm, had = int(lay["sfirst_row"][r, c, tid]), 0.0, False
                for p in range(pb[c], pb[c + 1]):
                    e = int(lay["sends"][p, tid])
                    assert all((j + 1) % lay["estep"] == 0 for j in range(SP) if (e >> j) & 1)
                    for j in range(SP):
                        sm += lds[(int(lay["sidx2"][p, tid, j // 2]) >> (16 * (j & 1))) & 0xffff] * float(lay["sprob"][p, tid, j])
                        if (e >> j) & 1:
                            accS[cc] += sm; cc += 1; sm = 0.0
                    had = had or e != 0
                tails[tid], has_end[tid] = sm, had
            scan(accS, tails, has_end, lay["sfirst_row"][r, c], lay["wcrow"][r, 2 + c], carries)

        shared = NC == 2 and off[1] == off[0] and cbeg[2] > cbeg[1]     # two chunks taking turns in one buffer (round 4)
        load(0)
        if not shared:
            load(1)
        resident(0)
        streamed(0)
        if shared:
            lds[:] = np.nan                                              # pass B must not depend on what chunk 0 left behind
            load(1)
        if NC > 2:
            load(2)
        resident(1)
        streamed(1)
        for c in range(2, NC):
            if c + 1 < NC:
                load(c + 1)
            streamed(c)
        rows = accS.copy()
        for ps in (0, 1):
            m = lay["rmap"][ps, row0:row1]
            rows[:n] += np.where(m >= 0, acc[ps][np.maximum(m, 0)], 0.0)
        for q, x in carries:
            if q >= 0:
                rows[q] += x
        out[row0:row1] = rows[:n]
    return out


@pytest.mark.parametrize("case", ["default", "bigstate", "overflow", "overflow_estep1", "chunks", "chunks_overflow", "res1", "res2_chunks",
                                  "res2_shared", "shared", "rows7"])
def test_den_graph_persistent2_layouts(case, monkeypatch):
    """Second persistent layout (chain_den_persist2.hip): table chunks, two resident passes, streamed overflow.  The numpy
    model of the kernel's frame reproduces both recursions' row sums: back-to-back chunks (default), a row spanning several
    waves (bigstate), lists longer than the resident slots (overflow: PK2_DP2_RES=8 leaves 8 register slots per pass), a
    vector longer than the LDS table (chunks: PK2_DP2_TCAP=512 -> two 256-entry buffers, up to 6 chunks), and both."""
    S, A, P, seed = 300, 6000, 23, 7
    kw = dict(loop_pdf_differs=True)
    monkeypatch.setenv("PK2_DEN_ORDER", "none")
    if case == "bigstate":
        S, A = 200, 40000
    if case.startswith("overflow"):
        S, A = 600, 300000
        kw["multi_entry_frac"] = 0.2
        monkeypatch.setenv("PK2_DP2_RES", "8")
        if case == "overflow_estep1":
            monkeypatch.setenv("PK2_DEN_ESTEP", "1")
    if case == "res1":              # what the GPU parity tests use to reach the streamed path on small graphs
        monkeypatch.setenv("PK2_DP2_RES", "1")
        S, A = 400, 30000
    if case in ("res2_chunks", "res2_shared", "shared"):
        if case != "shared":
            monkeypatch.setenv("PK2_DP2_RES", "2")
        monkeypatch.setenv("PK2_DP2_TCAP", "1024")
        if case == "res2_chunks":          # (a vector of up to twice the table would take the shared-buffer form)
            monkeypatch.setenv("PK2_DP2_SHARED", "0")
        S, A = 1400, (150000 if case == "res2_shared" else 50000)      # (res2_shared: lists beyond 2 slots x 512 threads -> streamed pieces)
        kw["multi_entry_frac"] = 0.3
    if case == "rows7":             # the LDS layout without the pdf arrays of the x gather (a graph they would cost a table chunk)
        monkeypatch.setenv("PK2_DP2_ROWARRAYS", "7")
    if case.startswith("chunks"):
        S, A = 1100, 20000
        monkeypatch.setenv("PK2_DP2_TCAP", "512")
        if case == "chunks_overflow":
            A = 200000
            monkeypatch.setenv("PK2_DP2_RES", "8")
    g = synth.den_graph_arcs(S, A, P, seed, **kw)
    if case == "bigstate":
        g["dst"][:30000] = 5
        g["pdf"][:20000] = 1; g["pdf"][20000:28000] = 2; g["pdf"][28000:30000] = 3
    G = chain.DenominatorGraph(g, P)
    of = G.debug_ordering(3)
    voff, vpdf, lpdf, lprob = of["voff"], of["vpdf"], of["loop_pdf"], of["loop_prob"].astype(np.float64)
    S_, V = g["num_states"], len(of["vpdf"])
    vstate = np.repeat(np.arange(S_), np.diff(voff))
    rng = np.random.default_rng(1)
    al, be, x = rng.random(S_), rng.random(S_), rng.random(P)
    xv = np.where(vpdf >= 0, x[np.maximum(vpdf, 0)], 1.0)
    xl = np.where(lpdf >= 0, x[np.maximum(lpdf, 0)], 1.0)
    src, dst, pdf, prob = g["src"], g["dst"], g["pdf"], g["prob"].astype(np.float64)
    pf, pb = G.debug_persist2(0), G.debug_persist2(1)
    assert pf is not None and pb is not None
    assert pf["R"] == S_ and pb["R"] == V and pf["row_begin"][-1] == V and pb["row_begin"][-1] == S_
    assert (voff[pf["grp_begin"]] == pf["row_begin"]).all()                     # whole states per workgroup
    assert pf["row_arrays"] == (7 if case == "rows7" else 8)      # (small graphs: the eighth array never costs a chunk)
    if case in ("default", "bigstate", "rows7"):
        assert pf["K"] == 2 and pb["K"] == 2 and pf["pieces"] == 0 and pb["pieces"] == 0
        assert pf["lds_off"][1] == pf["cbeg"][1] or pf["cbeg"][1] == pf["cbeg"][2]      # back to back
    if case.startswith("overflow"):
        assert pf["pieces"] > 0 and pb["pieces"] > 0 and pf["K"] == 2
        assert (pf["uncovered"] < pf["ncomp"]).any()                              # some list really is truncated
    if case == "res1":
        assert pf["estep"] == 1 and pf["pieces"] > 0 and pb["pieces"] > 0
    if case == "res2_chunks":
        assert pf["estep"] <= 2 and pf["K"] == 3 and pb["K"] >= 3 and pf["pieces"] > 0
    if case in ("res2_shared", "shared"):
        # S = 1400 -> 1536 table entries > the 1024 of the table, <= twice that: two chunks in ONE buffer, both lists resident
        assert pf["K"] == 2 and pf["lds_off"][:2] == [0, 0] and 0 < pf["cbeg"][1] < pf["cbeg"][2] == S_ and pf["tfloats"] <= 1024
        assert (pf["pieces"] > 0) == (case == "res2_shared")
        arcs_below = int((src[src != dst] < pf["cbeg"][1]).sum())
        assert 0.3 < arcs_below / float((src != dst).sum()) < 0.7       # the split halves the arcs, not the states
    if case.startswith("chunks"):
        assert pf["K"] == 5 and pb["K"] >= 5 and pf["tfloats"] == 512 and pf["lds_off"][:5] == [0, 256, 0, 256, 0]
        assert pf["pieces"] > 0            # chunks >= 2 have no resident pass
    rows = _emulate_persist2(pf, al)
    got = np.bincount(vstate, weights=rows * xv, minlength=S_) + al * lprob * xl
    want = np.bincount(dst, weights=al[src] * prob * x[pdf], minlength=S_)
    assert np.abs(got - want).max() < 1e-9 * max(1.0, np.abs(want).max())
    got = _emulate_persist2(pb, be[vstate] * xv) + lprob * xl * be
    want = np.bincount(src, weights=be[dst] * prob * x[pdf], minlength=S_)
    assert np.abs(got - want).max() < 1e-9 * max(1.0, np.abs(want).max())
    # the per-row constants the backward recursion predicts its sums with are those of the first layout
    p1 = G.debug_persist(0)
    if p1 is not None:
        assert np.array_equal(p1["row_leak"], pf["row_leak"]) and np.array_equal(p1["row_psum"], pf["row_psum"])


def test_den_graph_internal_state_order_is_invisible():
    """By default the library renumbers the states by in-degree (gather locality); initial_probs() still answers in the
    caller's numbering."""
    g = synth.den_graph_arcs(300, 6000, 23, 7, loop_pdf_differs=True)
    G = chain.DenominatorGraph(g, 23)
    ref = R.DenGraphRef(g["num_states"], g["src"], g["dst"], g["pdf"], g["prob"], 0, 23)
    assert np.abs(G.initial_probs() - ref.initial_probs).max() < 1e-7
    o = G.debug_ordering(3)
    indeg_sorted = np.sort(np.bincount(g["dst"][g["src"] != g["dst"]], minlength=g["num_states"]))[::-1]
    # rows of the forward ordering are (virtual) destination states in internal order: most-entered first
    lens = [int((o["arcs"][:, 2] != 0).sum())]       # total arcs kept
    assert lens[0] == int((g["src"] != g["dst"]).sum()) and indeg_sorted[0] >= indeg_sorted[-1]


def test_openfst_den_fst_reader(tmp_path):
    """Hand-assembled OpenFst binary: vector / standard, version 2."""
    arcs = {0: [(1, 1, 0.5, 1), (2, 2, 1.5, 0)], 1: [(3, 3, 0.25, 0), (1, 1, 0.0, 1)]}
    def s(b):
        return struct.pack("<i", len(b)) + b
    body = b""
    for st in range(2):
        body += struct.pack("<fq", 0.0, len(arcs[st]))
        for il, ol, w, ns in arcs[st]:
            body += struct.pack("<iifi", il, ol, w, ns)
    hdr = struct.pack("<i", 2125659606) + s(b"vector") + s(b"standard") + struct.pack("<iiQqqq", 2, 0, 0, 0, 2, 4)
    path = tmp_path / "den.fst"
    path.write_bytes(hdr + body)
    G = chain.DenominatorGraph(str(path), 3)
    assert (G.num_states(), G.num_pdfs(), G.num_arcs()) == (2, 3, 4)
    src = np.array([0, 0, 1, 1]); dst = np.array([1, 0, 0, 1]); pdf = np.array([0, 1, 2, 0])
    prob = np.exp(-np.array([0.5, 1.5, 0.25, 0.0]))
    want = R.initial_probs_ref(2, src, dst, prob, 0)
    assert np.abs(G.initial_probs() - want).max() < 1e-7
    bad = tmp_path / "bad.fst"
    bad.write_bytes(b"\x00" * 64)
    with pytest.raises(_lib.Pk2Error):
        chain.DenominatorGraph(str(bad), 3)


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.Pk2Error):
        _lib.require_gpu()
