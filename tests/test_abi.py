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
                assert s == 0.0
        out[o["row0"][ch]:o["row0"][ch] + len(acc)] += acc
    return out


@pytest.mark.parametrize("case", ["small", "bigrow", "sparse"])
def test_den_graph_orderings_reproduce_segment_sums(case):
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
