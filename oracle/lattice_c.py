"""ctypes wrapper of oracle/_build/liblattice_oracle.so -- TEST / BASELINE INFRASTRUCTURE ONLY (see lattice_oracle.c)."""
import ctypes as C

import numpy as np

from . import build_oracle

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle.build_lattice())
        _lib.lat_oracle_mmi.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def decode_mmi(graph, loglikes, tid2pdf, opts, ref_tids, num_pdfs, lm_scale=1.0, ac_scale=0.2, drop_frames=True, want_post=True):
    """graph: lattice_ref.DecodeGraphRef (arcs sorted by source state), opts: lattice_ref.DecoderOptionsRef.
    Returns dict(like, best_cost, links, toks, post[T, P] or None).  Releases the GIL for the whole call."""
    ll = np.ascontiguousarray(loglikes, np.float32)
    T, P = ll.shape
    t2p = np.ascontiguousarray(tid2pdf, np.int32)
    ref = np.ascontiguousarray(ref_tids, np.int32)[:T]
    post = np.zeros((T, num_pdfs), np.float64) if want_post else None
    like, best, links, toks = C.c_double(), C.c_float(), C.c_int64(), C.c_int64()
    off = np.ascontiguousarray(graph.off, np.int64)
    rc = lib().lat_oracle_mmi(C.c_int(graph.S), C.c_int(graph.start), _p(off), _p(graph.dst), _p(graph.ilabel), _p(graph.weight),
                              _p(graph.final), _p(ll), C.c_int(T), C.c_int(P), _p(t2p), C.c_float(float(opts.beam)),
                              C.c_float(float(opts.lattice_beam)), C.c_int(opts.max_active), C.c_int(opts.min_active),
                              C.c_float(float(opts.beam_delta)), C.c_float(float(opts.acoustic_scale)), _p(ref),
                              C.c_double(lm_scale), C.c_double(ac_scale), C.c_int(1 if drop_frames else 0), C.byref(like),
                              C.byref(best), C.byref(links), C.byref(toks), _p(post) if want_post else None)
    if rc:
        raise RuntimeError("lattice_oracle.c: %s" % ("no surviving token" if rc == 1 else "epsilon cycle"))
    return dict(like=like.value, best_cost=best.value, links=links.value, toks=toks.value, post=post)
