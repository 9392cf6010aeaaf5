"""TFLOP/s of pk2_gemm_f32 (default schedule) on every product of a BLSTM step: rows = frames x batch of the workload."""
import sys

import torch
from pykaldi2_amd.lstm import _gemm, _p
from pykaldi2_amd import _lib

dev = torch.device("cuda")


def run(ta, tb, M, N, K, batch=1):
    A = torch.randn(batch, *((K, M) if ta else (M, K)), device=dev)
    B = torch.randn(batch, *((N, K) if tb else (K, N)), device=dev)
    C = torch.empty(batch, M, N, device=dev)
    L = _lib.lib()

    def go():
        if batch == 1:
            _gemm(ta, tb, M, N, K, _p(A), A.shape[-1], _p(B), B.shape[-1], _p(C), N)
        else:
            _lib.check(L.pk2_gemm_f32_batched(ta, tb, M, N, K, 1.0, _p(A), A.shape[-1], A[0].numel(), 0, _p(B), B.shape[-1], B[0].numel(), 0,
                                              0.0, _p(C), N, M * N, 0, batch, 1, _lib.stream_ptr()))
    best = 1e9
    for _ in range(3):
        go(); go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            go()
        e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / 10)
    return best


for rows, P, name in ((2276, 6048, "LF-MMI (4 x 569 frames)"), (20480, 5768, "CE (256 x 80)"), (12896, 5768, "lattice-MMI (8 x 1612)")):
    shapes = [("gx l0", 0, 1, rows, 4096, 80, 1), ("gx l1/l2", 0, 1, rows, 4096, 1024, 1), ("logits", 0, 1, rows, P, 1024, 1),
              ("dy", 0, 0, rows, 1024, P, 1), ("dW_out", 1, 0, P, 1024, rows, 1), ("dprev", 0, 0, rows, 1024, 4096, 1),
              ("dW_ih", 1, 0, 4096, 1024, rows, 1), ("dW_ih l0", 1, 0, 4096, 80, rows, 1), ("dW_hh x2", 1, 0, 2048, 512, rows, 2)]
    tot_us = tot_f = 0.0
    print("== %s" % name)
    for nm, ta, tb, M, N, K, b in shapes:
        us = run(ta, tb, M, N, K, b)
        fl = 2.0 * M * N * K * b
        mult = {"gx l1/l2": 2, "dprev": 2, "dW_ih": 2, "dW_hh x2": 3}.get(nm, 1)
        tot_us += mult * us; tot_f += mult * fl
        print("%-10s ta=%d tb=%d %6d x %5d x %6d: %8.1f us %6.1f TFLOP/s (x%d per step)" % (nm, ta, tb, M, N, K, us, fl / us * 1e-6, mult), flush=True)
    print("   per step: %.2f ms, %.1f TFLOP/s" % (tot_us * 1e-3, tot_f / tot_us * 1e-6))
