"""Slice-count sweep (PK2_GEMM_SPLITK, read per call) on the deep-K products of the LF-MMI step."""
import os
import torch
from pykaldi2_amd.lstm import _gemm, _p

dev = torch.device("cuda")
SHAPES = [(0, 0, 2276, 1024, 4096), (0, 0, 2276, 1024, 6048), (1, 0, 4096, 1024, 2276), (1, 0, 6048, 1024, 2276)]
for ta, tb, M, N, K in SHAPES:
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    out = []
    for ks in ("auto", "2", "3", "4", "5", "6", "7", "8", "10", "12", "14", "16"):
        if ks == "auto":
            os.environ.pop("PK2_GEMM_SPLITK", None)
        else:
            os.environ["PK2_GEMM_SPLITK"] = ks
        best = 1e9
        for _ in range(3):
            for _ in range(2):
                _gemm(ta, tb, M, N, K, _p(A), A.shape[1], _p(B), B.shape[1], _p(C), N)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                _gemm(ta, tb, M, N, K, _p(A), A.shape[1], _p(B), B.shape[1], _p(C), N)
            e1.record(); torch.cuda.synchronize()
            best = min(best, 1e3 * e0.elapsed_time(e1) / 10)
        out.append("%s %.0f" % (ks, best))
    print("ta=%d tb=%d %5d x %5d x %5d us: " % (ta, tb, M, N, K) + " | ".join(out), flush=True)
