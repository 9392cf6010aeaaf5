"""LDS bank conflicts of the persistent denominator layouts on the bench graph (CPU, no GPU needed): cycles a 32-lane half
wave needs for one ds_read_b32 of the arc loop = the largest number of DISTINCT addresses that fall into one of the 32 banks
(lanes reading the same address are served by one broadcast)."""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.argv = ['bench.py']
import bench
from pykaldi2_amd import chain


def cycles(idx):
    """idx [R][K][T]: LDS word offsets; returns mean over (rank, slot, half wave)"""
    R, K, T = idx.shape
    a = idx.reshape(R, K, T // 32, 32)
    a = np.sort(a, axis=-1)
    first = np.ones(a.shape, bool)
    first[..., 1:] = a[..., 1:] != a[..., :-1]          # distinct addresses only
    bank = a % 32
    m = np.zeros(a.shape[:3], np.int64)
    for k in range(32):
        m = np.maximum(m, ((bank == k) & first).sum(-1))
    return m.mean()


g = bench.den_graph_arrays()
G = chain.DenominatorGraph(g, bench.P)
for w in (0, 1):
    p = G.debug_persist2(w)
    idx2 = p["idx2"]
    R, K2, T = idx2.shape
    idx = np.empty((R, K2 * 2, T), np.int64)
    idx[:, 0::2] = idx2 & 0xffff
    idx[:, 1::2] = idx2 >> 16
    print("second form, ordering %d: %.3f cycles per half-wave gather (pass A %.3f, pass B %.3f)" %
          (w, cycles(idx), cycles(idx[:, :K2]), cycles(idx[:, K2:])))
for w in (0, 1):
    p1 = G.debug_persist(w)
    if p1 is not None:
        print("first form, ordering %d: %.3f" % (w, cycles(p1["arcs"][..., 0].astype(np.int64))))
print("random offsets: %.3f" % cycles(np.random.default_rng(0).integers(0, 30000, size=(32, 64, 512))))
