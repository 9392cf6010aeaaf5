import sys, os
sys.path.insert(0, os.getcwd())
import torch
from pykaldi2_amd.lstm import _gemm, _p
dev = torch.device("cuda")
shapes = [(0, 1, 2276, 4096, 1024), (0, 1, 2276, 6048, 1024), (1, 0, 6048, 1024, 2276), (0, 0, 2276, 1024, 6048),
          (1, 0, 4096, 1024, 2276), (1, 0, 2048, 512, 2272), (0, 0, 2276, 1024, 4096), (0, 1, 2276, 4096, 80), (1, 0, 4096, 80, 2276)]
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tot_a = tot_b = 0
for ta, tb, M, N, K in shapes:
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    ours = t(lambda: _gemm(ta, tb, M, N, K, _p(A), A.shape[1], _p(B), B.shape[1], _p(C), N))
    Aop = A.t() if ta else A
    Bop = B.t() if tb else B
    lib = t(lambda: torch.mm(Aop, Bop, out=C))
    tot_a += ours; tot_b += lib
    fl = 2.0 * M * N * K / 1e9
    print("ta=%d tb=%d M=%5d N=%5d K=%5d  ours %7.1f us %6.1f TF/s | torch.mm (hipBLASLt/rocBLAS) %7.1f us %6.1f TF/s" % (ta, tb, M, N, K, 1e3 * ours, fl / ours, 1e3 * lib, fl / lib), flush=True)
print("sum ours %.3f ms, library %.3f ms" % (tot_a, tot_b))
