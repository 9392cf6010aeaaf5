import torch
from pykaldi2_amd.lstm import _gemm, _p
dev = torch.device("cuda")
M, N, K = 2356, 4096, 1024
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
for _ in range(3):
    _gemm(0, 1, M, N, K, _p(A), K, _p(B), K, _p(C), N)
torch.cuda.synchronize()
print("---- measured launch")
_gemm(0, 1, M, N, K, _p(A), K, _p(B), K, _p(C), N)
torch.cuda.synchronize()
