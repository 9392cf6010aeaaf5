"""f32 vs bf16x3 arithmetic of pk2_gemm_f32 on the GPU: error against float64 and time, same process, interleaved.
Shapes / value distributions follow tools/bf16x3_error_study.py (the CPU model of the same question).
  python tools/dbg/gemm_arith.py [err] [speed]"""
import sys

import numpy as np
import torch
from pykaldi2_amd import _lib
from pykaldi2_amd.lstm import _gemm, _p

dev = torch.device("cuda")
L = _lib.lib()


def run(ta, tb, A, B, arith):
    _lib.check(L.pk2_gemm_set_arith(arith))
    M = A.shape[1] if ta else A.shape[0]
    K = A.shape[0] if ta else A.shape[1]
    N = B.shape[0] if tb else B.shape[1]
    C = torch.empty(M, N, device=dev)
    _gemm(ta, tb, M, N, K, _p(A), A.shape[1], _p(B), B.shape[1], _p(C), N)
    return C


def err_study():
    torch.manual_seed(0)
    H = 512
    act = torch.tanh(torch.randn(2276, 1024)) * torch.sigmoid(torch.randn(2276, 1024))
    cases = [("BLSTM input projection x W_ih^T", 0, 1, act, ((torch.rand(4096, 1024) * 2 - 1) / H ** 0.5)),
             ("output layer x W_out^T", 0, 1, act, ((torch.rand(6048, 1024) * 2 - 1) / 1024 ** 0.5)),
             ("dX = dG W_ih (NN)", 0, 0, torch.randn(2276, 4096) * 1e-3, ((torch.rand(4096, 1024) * 2 - 1) / H ** 0.5)),
             ("weight gradient dG^T x act, K = 20480 (split-K)", 1, 0, torch.randn(20480, 1024) * 1e-3 * torch.rand(20480, 1).pow(4),
              torch.tanh(torch.randn(20480, 512)) * torch.sigmoid(torch.randn(20480, 512))),
             ("weight gradient K = 2276", 1, 0, torch.randn(2276, 4096) * 1e-3, act),
             ("N(0,1) x N(0,1), K = 4096", 0, 1, torch.randn(1024, 4096), torch.randn(1024, 4096)),
             ("layer-0 projection fbank x W_ih^T (K = 80)", 0, 1, torch.randn(2276, 80) * 4 + 10, ((torch.rand(4096, 80) * 2 - 1) / H ** 0.5)),
             ("TT", 1, 1, torch.randn(1030, 640), torch.randn(520, 1030))]
    ok = True
    for name, ta, tb, A, B in cases:
        a64 = (A.t() if ta else A).double()
        b64 = (B.t() if tb else B).double()
        ref = a64 @ b64
        scale = a64.abs() @ b64.abs()
        Ad, Bd = A.to(dev), B.to(dev)
        row = []
        for arith in (0, 1):
            got = run(ta, tb, Ad, Bd, arith).cpu().double()
            e = (got - ref).abs() / scale
            row.append((float(e.max()), float(torch.sqrt((e ** 2).mean())), float(((got - ref) / scale).mean())))
        print("%-52s f32: max %.3e rms %.3e bias %+.2e | bf16x3: max %.3e rms %.3e bias %+.2e  (units of sum|ab|)" %
              ((name,) + row[0] + row[1]), flush=True)
        ok = ok and row[1][0] <= 1.05 * row[0][0] and row[1][1] <= 1.05 * row[0][1]
    print("gate (bf16x3 max and rms error <= f32's on every case): %s" % ("PASS" if ok else "FAIL"))


def t_us(ta, tb, M, N, K, arith, reps=20):
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    import os
    if os.environ.get("PK2_ZERO"):      # (DVFS probe: zero operands toggle nothing; the same instruction stream clocks higher)
        A.zero_(); B.zero_()
    C = torch.empty(M, N, device=dev)
    _lib.check(L.pk2_gemm_set_arith(arith))
    best = 1e9
    for _ in range(3):
        for _ in range(2):
            _gemm(ta, tb, M, N, K, _p(A), A.shape[1], _p(B), B.shape[1], _p(C), N)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            _gemm(ta, tb, M, N, K, _p(A), A.shape[1], _p(B), B.shape[1], _p(C), N)
        e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best


def speed():
    SH = [(0, 1, 2276, 4096, 1024), (0, 1, 2276, 6048, 1024), (0, 0, 2276, 1024, 6048), (1, 0, 6048, 1024, 2276),
          (0, 0, 2276, 1024, 4096), (1, 0, 4096, 1024, 2276), (1, 0, 2048, 512, 2276), (0, 1, 2276, 4096, 80), (1, 0, 4096, 80, 2276),
          (0, 1, 20480, 4096, 1024), (0, 1, 20480, 5768, 1024), (0, 0, 20480, 1024, 5768), (1, 0, 5768, 1024, 20480),
          (0, 0, 20480, 1024, 4096), (1, 0, 4096, 1024, 20480), (1, 0, 2048, 512, 20480),
          (0, 1, 2300, 512, 512), (0, 1, 2300, 1536, 512), (0, 1, 2300, 2048, 512), (0, 0, 2300, 512, 2048), (1, 0, 512, 512, 2300),
          (0, 1, 4096, 4096, 4096), (0, 1, 8192, 8192, 1024)]
    for ta, tb, M, N, K in SH:
        f, x = t_us(ta, tb, M, N, K, 0), t_us(ta, tb, M, N, K, 1)
        fl = 2.0 * M * N * K * 1e-6
        print("ta=%d tb=%d %6d x %5d x %6d: f32 %8.1f us %6.1f TF/s | bf16x3 %8.1f us %6.1f TF/s-equiv  (x%.2f)" %
              (ta, tb, M, N, K, f, fl / f, x, fl / x, f / x), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["err", "speed"]
    if "err" in what:
        err_study()
    if "speed" in what:
        speed()
