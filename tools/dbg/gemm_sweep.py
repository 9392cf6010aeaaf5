"""GEMM timing sweeps (debug): how the time of the BLSTM input projection shape scales with K and M."""
import sys
import torch
from pykaldi2_amd.lstm import _gemm, _p

dev = torch.device("cuda")


def t(ta, tb, M, N, K, reps=20):
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    for _ in range(3):
        _gemm(ta, tb, M, N, K, _p(A), A.shape[1], _p(B), B.shape[1], _p(C), N)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _gemm(ta, tb, M, N, K, _p(A), A.shape[1], _p(B), B.shape[1], _p(C), N)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    print("ta=%d tb=%d M=%5d N=%5d K=%5d  %7.1f us  %6.1f TF/s" % (ta, tb, M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "small":       # the TransformerAM's products (d_model 512, FFN 2048)
        for sh in [(0, 1, 2300, 512, 512), (0, 1, 2300, 1536, 512), (0, 1, 2300, 2048, 512), (0, 1, 2300, 512, 2048),
                   (0, 0, 2300, 512, 512), (0, 0, 2300, 512, 2048), (0, 0, 2300, 2048, 512), (1, 0, 512, 512, 2300),
                   (1, 0, 2048, 512, 2300), (1, 0, 512, 2048, 2300), (0, 1, 2300, 512, 512)]:
            t(*sh, reps=50)
    else:
        for K in (512, 1024, 2048, 4096):
            t(0, 1, 2356, 4096, K)
        for M in (2048, 2304, 2356, 3072, 4096, 6144):
            t(0, 1, M, 4096, 1024)
