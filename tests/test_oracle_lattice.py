"""Pins for oracle/lattice_ref.py (the CPU restatement of the reference's MMI / sMBR lattice path, row a10):
brute-force path enumeration, exhaustive Viterbi, invariants.  Parity with Kaldi itself is unpinned
(no Kaldi here, no fixtures in the reference) -- see the oracle's header."""
import math

import numpy as np
import pytest

from oracle import lattice_ref as lr
from pykaldi2_amd import synth


def _setup(num_words=5, num_pdfs=12, T=7, seed=0, beam=30.0, lattice_beam=2.5, ac=1.0, max_phones=2, **kw):
    rng = np.random.default_rng(seed)
    g = synth.decoding_graph_arcs(num_words, num_pdfs, seed=seed, max_phones=max_phones)
    graph = lr.DecodeGraphRef(g["num_states"], g["start"], g["src"], g["dst"], g["ilabel"], g["weight"], g["final"])
    tm = synth.transition_model_arrays(num_pdfs)
    loglikes = (2.0 * rng.standard_normal((T, num_pdfs))).astype(np.float32)
    opts = lr.DecoderOptionsRef(beam=beam, lattice_beam=lattice_beam, acoustic_scale=ac, **kw)
    return graph, tm, loglikes, opts, rng


def _all_paths(lat, opts):
    A = lat.arrays()
    outs = {}
    for l in range(A["link_src"].shape[0]):
        outs.setdefault(int(A["link_src"][l]), []).append(l)
    res = []

    def walk(k, c, links):
        if A["tok_final"][k] != lr.INF:
            res.append((c + float(A["tok_final"][k]), tuple(links)))
        for l in outs.get(k, []):
            walk(int(A["link_dst"][l]), c + float(A["link_graph"][l]) + float(opts.acoustic_scale) * float(A["link_ac"][l]),
                 links + [l])

    walk(lat.start_tok, 0.0, [])
    return res


def test_best_path_matches_exhaustive_viterbi():
    for seed in range(4):
        graph, tm, ll, opts, _ = _setup(seed=seed, T=9)
        lat = lr.decode(graph, ll, tm["tid2pdf"], opts)
        want = lr.viterbi_best_cost(graph, ll, tm["tid2pdf"], float(opts.acoustic_scale))
        assert abs(lat.best_cost - want) < 1e-4 * max(1.0, abs(want))
        paths = _all_paths(lat, opts)
        assert abs(min(c for c, _ in paths) - want) < 1e-3


def test_lattice_beam_property():
    """Every kept link lies on a complete path within lattice_beam of the best one, and the lattice holds
    every such path of the (unpruned) search space."""
    graph, tm, ll, opts, _ = _setup(seed=1, T=6, lattice_beam=2.0)
    lat = lr.decode(graph, ll, tm["tid2pdf"], opts)
    paths = _all_paths(lat, opts)
    best = min(c for c, _ in paths)
    on_good_path = set()
    for c, links in paths:
        if c <= best + float(opts.lattice_beam) + 1e-4:
            on_good_path.update(links)
    assert on_good_path == set(range(len(lat.link_src)))
    # a wider lattice beam must contain the narrow lattice, and both agree on the paths within the narrow beam
    opts2 = lr.DecoderOptionsRef(beam=30.0, lattice_beam=6.0, acoustic_scale=1.0)
    lat2 = lr.decode(graph, ll, tm["tid2pdf"], opts2)
    assert lat.link_set() <= lat2.link_set()
    n_narrow = sum(1 for c, _ in paths if c <= best + 2.0 - 1e-4)
    n_wide = sum(1 for c, _ in _all_paths(lat2, opts2) if c <= best + 2.0 - 1e-4)
    assert n_narrow == n_wide


def test_get_cutoff_follows_kaldi():
    costs = np.arange(20, dtype=np.float32)
    o = lr.DecoderOptionsRef(beam=100.0, max_active=5, min_active=0)
    cut, ab = lr.get_cutoff(costs, o)
    assert cut == 5.0 and abs(ab - 5.5) < 1e-6          # cost of the (max_active+1)-th token, + beam_delta
    o = lr.DecoderOptionsRef(beam=3.0, max_active=1000, min_active=8)
    cut, ab = lr.get_cutoff(costs, o)
    assert cut == 8.0 and abs(ab - 8.5) < 1e-6          # fewer than min_active tokens inside the beam: widen
    o = lr.DecoderOptionsRef(beam=3.0, max_active=1000, min_active=2)
    cut, ab = lr.get_cutoff(costs, o)
    assert cut == 3.0 and ab == 3.0


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_mmi_matches_brute_force(seed):
    graph, tm, ll, opts, rng = _setup(seed=seed, T=6, lattice_beam=3.0)
    lat = lr.decode(graph, ll, tm["tid2pdf"], opts)
    P = ll.shape[1]
    ali = synth.tid_alignment(rng, lat.T, P)
    tot, post = lr.lattice_mmi(lat, ali, tm["tid2pdf"], P, 1.0, 0.2, drop_frames=False)
    bt, bden, _, _ = lr.brute_force_lattice(lat, ali, tm["tid2pdf"], tm["tid2phone"], tm["silence_phones"], P, 1.0, 0.2)
    assert abs(tot - bt) < 1e-9
    num = np.zeros_like(post)
    num[np.arange(lat.T), tm["tid2pdf"][ali]] = 1.0
    assert np.abs((num - post) - bden).max() < 1e-9
    assert np.abs(bden.sum(1) - 1.0).max() < 1e-9
    # drop_frames: frames whose reference transition-id is not in the lattice contribute nothing
    _, post_d = lr.lattice_mmi(lat, ali, tm["tid2pdf"], P, 1.0, 0.2, drop_frames=True)
    A = lat.arrays()
    for t in range(lat.T):
        present = any(A["tok_frame"][A["link_src"][l]] == t and A["link_tid"][l] == ali[t] for l in range(len(lat.link_src)))
        if present:
            assert np.abs(post_d[t] - post[t]).max() < 1e-12
        else:
            assert np.abs(post_d[t]).max() == 0.0


@pytest.mark.parametrize("criterion", ["smbr", "mpfe"])
def test_mpe_matches_brute_force(criterion):
    for seed in range(3):
        graph, tm, ll, opts, rng = _setup(seed=seed, T=6, lattice_beam=3.0)
        lat = lr.decode(graph, ll, tm["tid2pdf"], opts)
        P = ll.shape[1]
        # a reference alignment that follows the best path part of the time, so accuracies are not all zero
        ali = synth.tid_alignment(rng, lat.T, P)
        A = lat.arrays()
        for l in range(len(lat.link_src)):
            if A["link_tid"][l] != 0 and rng.random() < 0.3:
                ali[A["tok_frame"][A["link_src"][l]]] = A["link_tid"][l]
        score, post = lr.lattice_mpe(lat, ali, tm["tid2pdf"], tm["tid2phone"], tm["silence_phones"], P, criterion)
        _, _, bacc, bgrad = lr.brute_force_lattice(lat, ali, tm["tid2pdf"], tm["tid2phone"], tm["silence_phones"], P,
                                                   1.0, 1.0, criterion)
        assert abs(score - bacc) < 1e-9
        assert np.abs(post - bgrad).max() < 1e-9
        assert np.abs(post.sum(1)).max() < 1e-9


def test_beam_pruning_limits_tokens():
    graph, tm, ll, opts, _ = _setup(num_words=30, num_pdfs=30, T=12, seed=3, beam=4.0, lattice_beam=2.0, ac=0.5,
                                    max_phones=3, max_active=25, min_active=0)
    lat = lr.decode(graph, ll, tm["tid2pdf"], opts)
    wide = lr.decode(graph, ll, tm["tid2pdf"], lr.DecoderOptionsRef(beam=1e3, lattice_beam=2.0, acoustic_scale=0.5))
    assert lat.best_cost >= wide.best_cost - 1e-4
    if abs(lat.best_cost - wide.best_cost) < 1e-4:   # same best path: pruning can only remove links
        assert lat.link_set() <= wide.link_set()
    # the pruned search expands at most max_active (+ ties) tokens per frame
    assert len(lat.link_src) <= len(wide.link_src) or abs(lat.best_cost - wide.best_cost) >= 1e-4


SERIAL_CASES = [
    # words pdfs T seed beam lattice_beam ac max_active min_active | most extra links (fraction), largest posterior change
    ((40, 60, 40, 1, 8.0, 4.0, 0.5, 2 ** 31 - 1, 0), 0.06, 0.1),          # narrow beams, nothing binds
    ((200, 150, 60, 2, 13.0, 7.0, 0.1, 300, 200), 0.002, 2e-4),            # the configuration's beams, max_active binds
    ((60, 90, 30, 3, 4.0, 2.0, 1.0, 10000, 40), 0.0, 1e-12),
    ((300, 300, 120, 4, 10.0, 5.0, 0.3, 700, 200), 0.007, 7e-3),           # max_active binds: a few links are lost as well
]


@pytest.mark.parametrize("case,max_extra,max_dpost", SERIAL_CASES)
def test_final_cutoff_rule_against_kaldis_serial_process_emitting(case, max_extra, max_dpost):
    """Kaldi's ProcessEmitting tightens next_cutoff while it walks the token list, so the arcs it keeps depend on the
    order of its hash list; this build (oracle and HIP decoder) keeps an arc iff its cost is below the frame's FINAL
    cutoff.  Measured here with an emulation of the serial rule (lattice_ref.decode(serial_order=...)):
      * walking the tokens best-first, the serial rule IS the final-cutoff rule (identical lattices);
      * any other order only ADDS links whose cost lies between the final and the running cutoff (nothing this build
        keeps is lost while max_active does not bind), never changes the best path, and moves the MMI posteriors by
        what the bounds above record: ~1e-4 at the configuration's beams (13 / 7), 6e-3 at 10 / 5 with max_active
        binding (the extra tokens then also shift later frames: 0.1 % of this build's links are missing), up to 8e-2 at
        beam 8 / lattice-beam 4 on a 40-word graph, where a few dozen links are 5 % of the lattice."""
    nw, P, T, seed, beam, lb, ac, maxa, mina = case
    rng = np.random.default_rng(seed)
    g = synth.decoding_graph_arcs(nw, P, seed=seed, max_phones=3)
    graph = lr.DecodeGraphRef(g["num_states"], g["start"], g["src"], g["dst"], g["ilabel"], g["weight"], g["final"])
    tm = synth.transition_model_arrays(P)
    ll = (2.0 * rng.standard_normal((T, P))).astype(np.float32)
    opts = lr.DecoderOptionsRef(beam, lb, maxa, mina, 0.5, ac)
    base = lr.decode(graph, ll, tm["tid2pdf"], opts)
    ali = synth.tid_alignment(rng, base.T, P)
    like0, post0 = lr.lattice_mmi(base, ali, tm["tid2pdf"], P, 1.0, 0.2, False)
    links0 = base.link_set()
    best_first = lr.decode(graph, ll, tm["tid2pdf"], opts, serial_order="ascending")
    assert best_first.link_set() == links0
    assert np.abs(lr.lattice_mmi(best_first, ali, tm["tid2pdf"], P, 1.0, 0.2, False)[1] - post0).max() < 1e-12
    for order in ("descending", np.random.default_rng(5)):
        lat = lr.decode(graph, ll, tm["tid2pdf"], opts, serial_order=order)
        links = lat.link_set()
        assert lat.best_cost == base.best_cost
        if maxa > 10 ** 6:
            assert links0 <= links
        assert len(links - links0) <= max_extra * len(links0)
        like, post = lr.lattice_mmi(lat, ali, tm["tid2pdf"], P, 1.0, 0.2, False)
        assert np.abs(post - post0).max() <= max_dpost


@pytest.mark.parametrize("kw", [dict(seed=0, T=9), dict(seed=1, T=14, beam=8.0, lattice_beam=3.0), dict(seed=2, T=11, ac=0.3),
                                dict(seed=3, T=25, num_words=40, num_pdfs=60, max_phones=3, beam=11.0, lattice_beam=4.0, ac=0.1,
                                     max_active=30, min_active=5),
                                dict(seed=4, T=40, num_words=200, num_pdfs=90, max_phones=3, beam=13.0, lattice_beam=7.0, ac=0.1,
                                     max_active=150, min_active=20)])
def test_c_port_equals_numpy_oracle(kw):
    """oracle/lattice_oracle.c (the cpu_baseline of `bench.py --se`) against oracle/lattice_ref.py on the same input: best
    cost equal as float32, the same numbers of surviving tokens and kept links, MMI log-likelihood 1e-12 rel, posteriors
    1e-12 abs -- with max_active / min_active binding, a small acoustic scale, and frames that drop_frames drops."""
    from oracle import lattice_c
    graph, tm, ll, opts, rng = _setup(**kw)
    T = ll.shape[0]
    P = ll.shape[1]
    lat = lr.decode(graph, ll, tm["tid2pdf"], opts)
    # a reference alignment: the transition-ids along one path of the lattice for some frames, an absent id for others
    A = lat.arrays()
    ref = np.zeros(T, np.int32)
    for t in range(T):
        em = [int(A["link_tid"][l]) for l in range(A["link_tid"].shape[0]) if A["link_tid"][l] != 0 and A["tok_frame"][A["link_src"][l]] == t]
        ref[t] = em[int(rng.integers(len(em)))] if (em and t % 4 != 3) else 1 + int(rng.integers(len(tm["tid2pdf"]) - 1))
    want_like, want_post = lr.lattice_mmi(lat, ref, tm["tid2pdf"], P, 1.0, 0.2, True)
    got = lattice_c.decode_mmi(graph, ll, tm["tid2pdf"], opts, ref, P, 1.0, 0.2, True)
    assert np.float32(got["best_cost"]) == np.float32(lat.best_cost)
    assert got["toks"] == A["tok_state"].shape[0] and got["links"] == A["link_src"].shape[0]
    assert abs(got["like"] - want_like) <= 1e-12 * max(1.0, abs(want_like))
    assert np.abs(got["post"] - want_post).max() < 1e-12
