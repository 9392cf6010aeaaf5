"""oracle/chain_oracle.c (the CPU-baseline port) pinned against the numpy oracle."""
import numpy as np

from oracle import chain_c, chain_ref as R
from pykaldi2_amd import chain, synth


def test_c_port_matches_numpy_oracle():
    P = 60
    g = synth.den_graph_arcs(400, 6000, P, seed=3)
    ref = R.DenGraphRef(g["num_states"], g["src"], g["dst"], g["pdf"], g["prob"], 0, P)
    rng = np.random.default_rng(1)
    sups = [chain.Supervision(synth.numerator_fst_from_alignment(synth.pdf_alignment(rng, T, P))) for T in (151, 88)]
    Tm = max(s.frames_per_sequence for s in sups)
    lg = rng.normal(0, 2, size=(2, Tm, P)).astype(np.float32)
    lp, gam, chk = chain_c.den_fb(g, ref.initial_probs, lg[0], 1e-4)
    want_lp, want_g, _ = R.den_forward_backward(lg[0].astype(np.float64), ref, 1e-4)
    assert abs(lp - want_lp) < 1e-5 * abs(want_lp) and np.abs(gam - want_g).max() < 1e-5 and abs(chk - 1) < 1e-4
    out, grad = chain_c.chain_batch(g, ref.initial_probs, lg, sups, 1e-4, 0.1)
    for n, s in enumerate(sups):
        T = s.frames_per_sequence
        f = R.NumFstRef(s.num_states, s.src, s.dst, s.pdf, s.arc_weight, s.final_states, s.final_weights, s.state_time)
        objf, want, aux = R.chain_objf_and_deriv(lg[n, :T].astype(np.float64), ref, f, leaky=1e-4, xent_regularize=0.1)
        assert abs(out[0, n] - objf) < 1e-5 * abs(objf)
        assert np.abs(grad[n, :T] - want).max() < 1e-5
        assert not grad[n, T:].any()


def test_c_port_float64_build_matches_numpy_oracle_tightly():
    """libchain_oracle_f64.so (-DORC_REAL=double): the parity reference of the bench-size tests and of bench.py's
    `parity` block -- the same recursion as the float build, without its rounding."""
    P = 60
    g = synth.den_graph_arcs(400, 6000, P, seed=3, loop_pdf_differs=True)
    ref = R.DenGraphRef(g["num_states"], g["src"], g["dst"], g["pdf"], g["prob"], 0, P)
    rng = np.random.default_rng(2)
    sups = [chain.Supervision(synth.numerator_fst_from_alignment(synth.pdf_alignment(rng, T, P))) for T in (120, 301)]
    Tm = max(s.frames_per_sequence for s in sups)
    lg = rng.normal(0, 2, size=(2, Tm, P)).astype(np.float32)
    # (initial_probs is handed over as float32, like the float build receives it)
    pi32 = ref.initial_probs.astype(np.float32)
    lp, gam, chk = chain_c.den_fb(g, pi32, lg[1], 1e-4, double=True)
    want_lp, want_g, _ = R.den_forward_backward(lg[1].astype(np.float64), ref, 1e-4)
    assert abs(lp - want_lp) < 1e-6 * abs(want_lp) and np.abs(gam - want_g).max() < 2e-7 and abs(chk - 1) < 1e-6
    out, grad = chain_c.chain_batch(g, pi32, lg, sups, 1e-4, 0.1, double=True)
    out32, grad32 = chain_c.chain_batch(g, pi32, lg, sups, 1e-4, 0.1)
    for n, s in enumerate(sups):
        T = s.frames_per_sequence
        f = R.NumFstRef(s.num_states, s.src, s.dst, s.pdf, s.arc_weight, s.final_states, s.final_weights, s.state_time)
        objf, want, aux = R.chain_objf_and_deriv(lg[n, :T].astype(np.float64), ref, f, leaky=1e-4, xent_regularize=0.1)
        assert abs(out[0, n] - objf) < 1e-6 * abs(objf)
        assert np.abs(grad[n, :T] - want).max() < 5e-7
        assert np.abs(grad32[n, :T] - want).max() < 1e-5
