"""Deep-K products with between one and two 128x128 tiles per CU: 64x64 tiles (PK2_GEMM_SPLIT_MID=0, a child process: the
switch is read once) against 128x128 tiles over K slices (default)."""
import os
import subprocess
import sys

SHAPES = [(1, 0, 5768, 1024, 20480), (1, 0, 6048, 1024, 2276), (1, 0, 4096, 1536, 4096), (1, 0, 5768, 1024, 8000), (1, 0, 6048, 1024, 7000)]
if len(sys.argv) > 1:
    import torch
    from pykaldi2_amd.lstm import _gemm, _p
    dev = torch.device("cuda")
    for ta, tb, M, N, K in SHAPES:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        B = torch.randn((N, K) if tb else (K, N), device=dev)
        C = torch.empty(M, N, device=dev)
        best = 1e9
        for _ in range(4):
            for _ in range(2):
                _gemm(ta, tb, M, N, K, _p(A), A.shape[1], _p(B), B.shape[1], _p(C), N)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                _gemm(ta, tb, M, N, K, _p(A), A.shape[1], _p(B), B.shape[1], _p(C), N)
            e1.record(); torch.cuda.synchronize()
            best = min(best, 1e3 * e0.elapsed_time(e1) / 10)
        ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
        err = float((C.double() - ref).abs().max() / ref.abs().max())
        print("%s ta=%d tb=%d %5d x %5d x %5d: %7.1f us  %6.1f TFLOP/s  rel err %.1e" % (sys.argv[1], ta, tb, M, N, K, best, 2e-6 * M * N * K / best, err), flush=True)
else:
    for mode in ("1", "0"):
        subprocess.run([sys.executable, __file__, "split_mid=" + mode], env=dict(os.environ, PK2_GEMM_SPLIT_MID=mode))

