"""ctypes wrapper of oracle/_build/libchain_oracle.so -- TEST / BASELINE INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from . import build_oracle

_libs = {}


def lib(double=False):
    """double=False: float32 state like Kaldi's BaseFloat (the timed cpu_baseline); True: the float64 build."""
    if double not in _libs:
        _libs[double] = C.CDLL(build_oracle.build(double))
        _libs[double].orc_den_fb.restype = C.c_double
        _libs[double].orc_set_beta_seed.argtypes = [C.c_double]
    return _libs[double]


def set_beta_seed(v, double=False):
    """Test hook (tests of the abandon rule): beta'[T] = v / tot; 1.0 restores Kaldi's seed."""
    lib(double).orc_set_beta_seed(C.c_double(v))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def den_fb(g, pi, logits, leaky, double=False):
    """g: dict of arc arrays; returns (logprob, gamma[T,P], check)."""
    T, P = logits.shape
    lg = np.ascontiguousarray(logits, dtype=np.float32)
    gamma = np.zeros((T, P), dtype=np.float32)
    chk = C.c_double()
    src, dst, pdf = (np.ascontiguousarray(g[k], dtype=np.int32) for k in ("src", "dst", "pdf"))
    prob = np.ascontiguousarray(g["prob"], dtype=np.float32)
    pi = np.ascontiguousarray(pi, dtype=np.float32)
    lp = lib(double).orc_den_fb(C.c_int(g["num_states"]), C.c_int(P), C.c_int64(src.shape[0]), _p(src), _p(dst), _p(pdf),
                          _p(prob), _p(pi), _p(lg), C.c_int64(P), C.c_int(T), C.c_float(leaky), C.c_float(1.0),
                          _p(gamma), C.c_int64(P), C.byref(chk))
    return lp, gamma, chk.value


def chain_batch(g, pi, logits, sups, leaky, xent, weight=1.0, double=False):
    """logits [N,Tmax,P] float32; sups: list of objects with the chain.Supervision fields.
    Returns (out[3,N], grad[N,Tmax,P])."""
    N, Tmax, P = logits.shape
    lg = np.ascontiguousarray(logits, dtype=np.float32)
    grad = np.zeros_like(lg)
    src, dst, pdf = (np.ascontiguousarray(g[k], dtype=np.int32) for k in ("src", "dst", "pdf"))
    prob = np.ascontiguousarray(g["prob"], dtype=np.float32)
    pi = np.ascontiguousarray(pi, dtype=np.float32)
    lengths = np.asarray([s.frames_per_sequence for s in sups], dtype=np.int32)
    arc_base = np.cumsum([0] + [s.src.shape[0] for s in sups]).astype(np.int32)
    n_src = np.concatenate([s.src for s in sups]).astype(np.int32)
    n_dst = np.concatenate([s.dst for s in sups]).astype(np.int32)
    n_pdf = np.concatenate([s.pdf for s in sups]).astype(np.int32)
    n_w = np.concatenate([s.arc_weight for s in sups]).astype(np.float32)
    n_foff = np.concatenate([s.frame_offsets + arc_base[i] for i, s in enumerate(sups)]).astype(np.int32)
    n_states = np.asarray([s.num_states for s in sups], dtype=np.int32)
    n_fin = np.concatenate([s.final_states for s in sups]).astype(np.int32)
    n_finw = np.concatenate([s.final_weights for s in sups]).astype(np.float32)
    n_finoff = np.cumsum([0] + [s.final_states.shape[0] for s in sups]).astype(np.int32)
    out = np.zeros(3 * N, dtype=np.float64)
    lib(double).orc_chain_batch(C.c_int(g["num_states"]), C.c_int(P), C.c_int64(src.shape[0]), _p(src), _p(dst), _p(pdf),
                          _p(prob), _p(pi), _p(lg), C.c_int64(Tmax * P), C.c_int64(P), _p(lengths), C.c_int(N),
                          _p(n_src), _p(n_dst), _p(n_pdf), _p(n_w), _p(n_foff), _p(arc_base), _p(n_states),
                          _p(n_fin), _p(n_finw), _p(n_finoff), C.c_float(leaky), C.c_float(xent), C.c_float(weight),
                          _p(grad), _p(out))
    return out.reshape(3, N), grad
