"""Lattice determinisation (csrc/lattice_det.hip, pykaldi2_amd.lattice.determinize_lattice; reference bin/latgen.py:149
`determinize_lattice = True`) against brute-force path enumeration on small random lattices: host code only, no GPU."""
import itertools
import os

import numpy as np
import pytest

from pykaldi2_amd import kaldi_io, lattice


def _random_lattice(rng, frames, width, words, eps_frac=0.5, tid_frac=0.8, extra_eps=True):
    """A frame-layered acyclic lattice like the decoder's: `width` states per layer, arcs layer t -> t + 1 carry a
    transition-id and mostly no word, some epsilon arcs (no tid) inside a layer, finals in the last layer."""
    n = (frames + 1) * width
    src, dst, word, tid, g, a = [], [], [], [], [], []
    st = lambda t, i: t * width + i
    for t in range(frames):
        for i in range(width):
            for j in rng.choice(width, size=rng.integers(1, min(3, width) + 1), replace=False):
                src.append(st(t, i)); dst.append(st(t + 1, int(j)))
                word.append(0 if rng.random() < eps_frac else int(rng.integers(1, words + 1)))
                tid.append(int(rng.integers(1, 50)) if rng.random() < tid_frac else 0)
                g.append(float(rng.uniform(0, 2))); a.append(float(rng.normal(0, 2)))
        if extra_eps:                      # epsilon arcs inside the layer (i -> i + 1: acyclic)
            for i in range(width - 1):
                if rng.random() < 0.4:
                    src.append(st(t + 1, i)); dst.append(st(t + 1, i + 1)); word.append(0); tid.append(0)
                    g.append(float(rng.uniform(0, 1))); a.append(0.0)
    fin = np.full(n, np.inf, np.float32)
    for i in range(width):
        if rng.random() < 0.7 or i == 0:
            fin[st(frames, i)] = rng.uniform(0, 1)
    return dict(num_states=n, start=0, src=np.array(src, np.int32), dst=np.array(dst, np.int32), word=np.array(word, np.int32),
                tid=np.array(tid, np.int32), graph=np.array(g, np.float32), acoustic=np.array(a, np.float32), final=fin)


def _paths_raw(lat):
    """{word sequence: (best total, graph, acoustic, tids)} by exhaustive enumeration (float64 sums of the float32 costs)."""
    out = {}
    arcs = {}
    for i in range(lat["src"].shape[0]):
        arcs.setdefault(int(lat["src"][i]), []).append(i)
    def walk(s, ws, g, a, ts):
        if np.isfinite(lat["final"][s]):
            key, cand = tuple(ws), (g + float(lat["final"][s]) + a, g + float(lat["final"][s]), a, tuple(ts))
            if key not in out or cand[:2] < out[key][:2]:
                out[key] = cand
        for i in arcs.get(s, []):
            w, t = int(lat["word"][i]), int(lat["tid"][i])
            walk(int(lat["dst"][i]), ws + [w] if w else ws, g + float(lat["graph"][i]), a + float(lat["acoustic"][i]),
                 ts + [t] if t else ts)
    walk(int(lat["start"]), [], 0.0, 0.0, [])
    return out


def _paths_det(det):
    out = {}
    arcs = {}
    for i in range(det["src"].shape[0]):
        arcs.setdefault(int(det["src"][i]), []).append(i)
    def walk(s, ws, g, a, ts):
        if np.isfinite(det["final"][s]):
            fids = [int(t) for t in det["final_tids"][det["final_tid_off"][s]:det["final_tid_off"][s + 1]]]
            key = tuple(ws)
            assert key not in out, "a word sequence is accepted along two paths: not deterministic"
            out[key] = (g + float(det["final"][s]) + a + float(det["final_acoustic"][s]), g + float(det["final"][s]),
                        a + float(det["final_acoustic"][s]), tuple(ts + fids))
        for i in arcs.get(s, []):
            ids = [int(t) for t in det["tids"][det["tid_off"][i]:det["tid_off"][i + 1]]]
            walk(int(det["dst"][i]), ws + [int(det["word"][i])], g + float(det["graph"][i]), a + float(det["acoustic"][i]), ts + ids)
    walk(int(det["start"]), [], 0.0, 0.0, [])
    return out


@pytest.mark.parametrize("seed,frames,width,words,beam", [(0, 5, 3, 2, 1e9), (1, 6, 3, 3, 1e9), (2, 7, 2, 2, 1e9), (3, 6, 4, 2, 1e9),
                                                           (4, 6, 3, 2, 3.0), (5, 7, 3, 3, 2.0), (6, 5, 4, 4, 4.0)])
def test_determinised_lattice_holds_the_best_path_of_every_word_sequence(seed, frames, width, words, beam):
    rng = np.random.default_rng(seed)
    lat = _random_lattice(rng, frames, width, words)
    want = _paths_raw(lat)
    det = lattice.determinize_lattice(lat, beam)
    # deterministic: one arc per (state, word), no epsilon words
    assert (det["word"] > 0).all()
    assert len(set(zip(det["src"].tolist(), det["word"].tolist()))) == det["src"].shape[0]
    got = _paths_det(det)
    best = min(v[0] for v in want.values())
    # pruning is by ARC, as in Kaldi: everything within the beam survives, and every surviving arc lies on some complete
    # path within the beam (a patchwork of such arcs may still spell a word sequence that costs more)
    keep = {k: v for k, v in want.items() if v[0] <= best + beam + 1e-4}
    assert set(keep) <= set(got), (len(keep), len(got), len(want))
    if beam < 1e8:
        assert len(got) < len(want)
        on_good_path = np.zeros(det["src"].shape[0], bool)
        outa = {}
        for i in range(det["src"].shape[0]):
            outa.setdefault(int(det["src"][i]), []).append(i)
        def mark(s_, tot, used):
            if np.isfinite(det["final"][s_]) and tot + float(det["final"][s_]) + float(det["final_acoustic"][s_]) <= best + beam + 1e-4:
                on_good_path[used] = True
            for i in outa.get(s_, []):
                mark(int(det["dst"][i]), tot + float(det["graph"][i]) + float(det["acoustic"][i]), used + [i])
        mark(int(det["start"]), 0.0, [])
        assert on_good_path.all()
    for k, v in got.items():
        w = want[k]
        assert abs(v[0] - w[0]) < 1e-4 and abs(v[1] - w[1]) < 1e-4 and abs(v[2] - w[2]) < 1e-4, (k, v, w)
        assert v[3] == w[3], (k, v[3], w[3])             # the alignment of the best path
    # the best path overall is untouched
    kb = min(want, key=lambda k: want[k][0])
    assert abs(got[kb][0] - best) < 1e-4


def test_acoustic_scale_is_applied_for_pruning_and_removed_in_the_output():
    rng = np.random.default_rng(11)
    lat = _random_lattice(rng, 6, 3, 2)
    sc = 0.1
    scaled = dict(lat, acoustic=(lat["acoustic"].astype(np.float64) * sc).astype(np.float32))
    want = _paths_raw(scaled)
    det = lattice.determinize_lattice(lat, 1e9, acoustic_scale=sc)
    got = _paths_det(dict(det, acoustic=(det["acoustic"].astype(np.float64) * sc).astype(np.float32),
                          final_acoustic=np.where(np.isfinite(det["final_acoustic"]), det["final_acoustic"] * sc, np.inf).astype(np.float32)))
    assert set(got) == set(want)
    for k in want:
        assert abs(got[k][0] - want[k][0]) < 1e-4 and got[k][3] == want[k][3]


def test_state_limit_reduces_the_beam_and_errors_are_loud():
    rng = np.random.default_rng(3)
    lat = _random_lattice(rng, 8, 4, 3)
    full = lattice.determinize_lattice(lat, 50.0)
    tight = lattice.determinize_lattice(lat, 3.0)
    assert tight["num_states"] < full["num_states"]
    small = lattice.determinize_lattice(lat, 50.0, max_states=tight["num_states"])      # 50 -> 25 -> ... -> 3.125 or less
    assert small["beam_used"] <= 3.125 and small["num_states"] <= tight["num_states"]
    got, want = _paths_det(small), _paths_raw(lat)
    best = min(v[0] for v in want.values())
    kb = min(want, key=lambda k: want[k][0])
    assert abs(got[kb][0] - best) < 1e-4                 # the best path survives any beam
    from pykaldi2_amd import _lib
    with pytest.raises(_lib.Pk2Error):                   # a cycle is not a lattice
        bad = dict(lat, src=np.append(lat["src"], 5).astype(np.int32), dst=np.append(lat["dst"], 0).astype(np.int32),
                   word=np.append(lat["word"], 0).astype(np.int32), tid=np.append(lat["tid"], 0).astype(np.int32),
                   graph=np.append(lat["graph"], 0).astype(np.float32), acoustic=np.append(lat["acoustic"], 0).astype(np.float32))
        lattice.determinize_lattice(bad, 10.0)
    with pytest.raises(_lib.Pk2Error):                   # no final state reachable
        lattice.determinize_lattice(dict(lat, final=np.full_like(lat["final"], np.inf)), 10.0)


def test_determinised_lattice_round_trips_through_the_compact_lattice_archive(tmp_path):
    rng = np.random.default_rng(7)
    lat = _random_lattice(rng, 6, 3, 3)
    det = lattice.determinize_lattice(lat, 1e9)
    for ext in ("ark", "txt"):
        path = str(tmp_path / ("lat." + ext))
        with kaldi_io.CompactLatticeWriter(("ark,t:" if ext == "txt" else "ark:") + path) as w:
            w["utt1"] = det
        assert os.path.getsize(path) > 0
    (key, back), = list(kaldi_io.read_compact_lattice_ark(str(tmp_path / "lat.ark")))
    assert key == "utt1" and back["num_states"] == det["num_states"] and len(back["arcs"]) == det["src"].shape[0]
    # (the writer renumbers the start state to 0)
    seqs = {}
    arcs = {}
    for s, d, il, ol, (g, a, ids) in back["arcs"]:
        assert il == ol
        arcs.setdefault(s, []).append((d, il, g, a, ids))
    def walk(s, ws, tot, ts):
        g, a, ids = back["finals"][s]
        if np.isfinite(g):
            seqs[tuple(ws)] = (tot + g + a, tuple(ts + ids))
        for d, w, g2, a2, ids2 in arcs.get(s, []):
            walk(d, ws + [w], tot + g2 + a2, ts + ids2)
    walk(0, [], 0.0, [])
    want = _paths_det(det)
    assert set(seqs) == set(want)
    for k in want:
        assert abs(seqs[k][0] - want[k][0]) < 1e-4 and seqs[k][1] == want[k][3]
    text = open(str(tmp_path / "lat.txt")).read().splitlines()
    assert text[0] == "utt1" and any("_" in l.split("\t")[-1] for l in text[1:] if l)     # multi-id strings are written joined by '_'


def _decoder_lattice(nw, P, T, seed, beam, lb, ac, maxa, mina):
    """A raw state-level lattice from the CPU oracle decoder (oracle/lattice_ref.py), with the HCLG's word labels."""
    from oracle import lattice_ref as lr
    from pykaldi2_amd import synth
    rng = np.random.default_rng(seed)
    g = synth.decoding_graph_arcs(nw, P, seed=seed, max_phones=3)
    tm = synth.transition_model_arrays(P)
    ll = (2.0 * rng.standard_normal((T, P))).astype(np.float32)
    graph = lr.DecodeGraphRef(g["num_states"], g["start"], g["src"], g["dst"], g["ilabel"], g["weight"], g["final"])
    base = lr.decode(graph, ll, tm["tid2pdf"], lr.DecoderOptionsRef(beam, lb, maxa, mina, 0.5, ac))
    A = base.arrays()
    olabel = {(int(s), int(d), int(i), float(np.float32(w))): int(o)
              for s, d, i, o, w in zip(g["src"], g["dst"], g["ilabel"], g["olabel"], g["weight"])}
    st, fr = A["tok_state"], A["tok_frame"]
    words = np.array([olabel[(int(st[s]), int(st[d]), int(t), float(np.float32(gc)))]
                      for s, d, t, gc in zip(A["link_src"], A["link_dst"], A["link_tid"], A["link_graph"])], np.int32)
    last = fr == fr.max()
    fin = A["tok_final"].astype(np.float32).copy()
    if not np.isfinite(fin[last]).any():
        fin[last] = 0.0
    fin[~last] = np.inf
    indeg = np.bincount(A["link_dst"], minlength=fr.shape[0])
    start = int(np.flatnonzero((fr == 0) & (indeg == 0))[0])
    return dict(num_states=int(fr.shape[0]), start=start, src=A["link_src"], dst=A["link_dst"], word=words, tid=A["link_tid"],
                graph=A["link_graph"], acoustic=A["link_ac"], final=fin), base


def _best_path_with_words(lat, ac, seq):
    """Cost and transition-ids of the best raw path that spells `seq`: Viterbi over (lattice state, words consumed)."""
    n, K = lat["num_states"], len(seq)
    order = np.argsort(lat["src"], kind="stable")
    indeg = np.bincount(lat["dst"], minlength=n)
    lo = np.searchsorted(lat["src"][order], np.arange(n + 1))
    cost = np.full((n, K + 1), np.inf); back = {}
    cost[lat["start"], 0] = 0.0
    stack = [s for s in range(n) if indeg[s] == 0]
    while stack:                                            # topological order
        s = stack.pop()
        for l in order[lo[s]:lo[s + 1]]:
            d, w = int(lat["dst"][l]), int(lat["word"][l])
            c = float(lat["graph"][l]) + ac * float(lat["acoustic"][l])
            for k in range(K + 1):
                if not np.isfinite(cost[s, k]):
                    continue
                k2 = k if w == 0 else (k + 1 if k < K and seq[k] == w else -1)
                if k2 >= 0 and cost[s, k] + c < cost[d, k2]:
                    cost[d, k2] = cost[s, k] + c; back[(d, k2)] = (s, k, int(l))
            indeg[d] -= 1
            if indeg[d] == 0:
                stack.append(d)
    ends = [(cost[s, K] + float(lat["final"][s]), s) for s in range(n) if np.isfinite(lat["final"][s]) and np.isfinite(cost[s, K])]
    if not ends:
        return np.inf, None
    total, s = min(ends)
    tids, k = [], K
    while (s, k) in back:
        s, k, l = back[(s, k)]
        if lat["tid"][l] > 0:
            tids.append(int(lat["tid"][l]))
    return total, tuple(reversed(tids))


@pytest.mark.parametrize("case", [(40, 60, 40, 1, 8.0, 4.0, 0.5, 2 ** 31 - 1, 0), (200, 150, 60, 2, 13.0, 7.0, 0.1, 300, 200)])
def test_decoder_lattice_is_determinised_exactly(case):
    """On lattices of the decoder's shape (thousands of states, most arcs without a word): the output is deterministic,
    the best path and its cost are the raw lattice's, and the word sequences spelled by random walks through the output
    carry exactly the cost and the alignment of the best raw path with those words (constrained Viterbi on the raw
    lattice)."""
    nw, P, T, seed, beam, lb, ac, maxa, mina = case
    lat, base = _decoder_lattice(*case)
    det = lattice.determinize_lattice(lat, lb, acoustic_scale=ac)
    assert det["beam_used"] == lb and det["num_states"] < lat["num_states"] and (det["word"] > 0).all()
    assert len(set(zip(det["src"].tolist(), det["word"].tolist()))) == det["src"].shape[0]
    outa = {}
    for i in range(det["src"].shape[0]):
        outa.setdefault(int(det["src"][i]), []).append(i)
    def walk(choose):
        s, ws, tot, ts = int(det["start"]), [], 0.0, []
        while True:
            arcs = outa.get(s, [])
            stop = np.isfinite(det["final"][s]) and (not arcs or choose(len(arcs) + 1) == 0)
            if stop:
                ts += [int(t) for t in det["final_tids"][det["final_tid_off"][s]:det["final_tid_off"][s + 1]]]
                return ws, tot + float(det["final"][s]) + ac * float(det["final_acoustic"][s]), tuple(ts)
            i = arcs[choose(len(arcs)) % len(arcs)]
            ws.append(int(det["word"][i])); tot += float(det["graph"][i]) + ac * float(det["acoustic"][i])
            ts += [int(t) for t in det["tids"][det["tid_off"][i]:det["tid_off"][i + 1]]]
            s = int(det["dst"][i])
    rng = np.random.default_rng(0)
    seen = set()
    for _ in range(40):
        ws, tot, ts = walk(lambda k: int(rng.integers(k)))
        if tuple(ws) in seen:
            continue
        seen.add(tuple(ws))
        want, want_ts = _best_path_with_words(lat, ac, ws)
        assert abs(tot - want) < 2e-3, (ws, tot, want)       # (float32 costs summed in different orders)
        assert ts == want_ts, ws
    # the best path overall: same cost in both lattices (shortest path by relaxation in topological order)
    def shortest(n, start, src, dst, cost, fin):
        d = np.full(n, np.inf); d[start] = 0.0
        indeg = np.bincount(dst, minlength=n)
        order = np.argsort(src, kind="stable"); lo = np.searchsorted(src[order], np.arange(n + 1))
        stack = [s_ for s_ in range(n) if indeg[s_] == 0]
        while stack:
            s_ = stack.pop()
            for l in order[lo[s_]:lo[s_ + 1]]:
                d[dst[l]] = min(d[dst[l]], d[s_] + cost[l])
                indeg[dst[l]] -= 1
                if indeg[dst[l]] == 0:
                    stack.append(int(dst[l]))
        return float(np.min(d + fin))
    raw_best = shortest(lat["num_states"], lat["start"], lat["src"], lat["dst"],
                        lat["graph"].astype(np.float64) + ac * lat["acoustic"].astype(np.float64), lat["final"].astype(np.float64))
    det_fin = det["final"].astype(np.float64) + ac * np.where(np.isfinite(det["final_acoustic"]), det["final_acoustic"], 0.0)
    det_best = shortest(det["num_states"], det["start"], det["src"], det["dst"],
                        det["graph"].astype(np.float64) + ac * det["acoustic"].astype(np.float64), det_fin)
    assert abs(raw_best - det_best) < 2e-3 and abs(raw_best - float(base.best_cost)) < 2e-3
