"""TFLOP/s of pk2_gemm_f32 on the TransformerAM's products (rows = frames x batch of the LF-MMI minibatch)."""
import torch
from pykaldi2_amd.lstm import _gemm, _p

dev = torch.device("cuda")
rows = 2276
SHAPES = [("in", 0, 1, rows, 512, 80), ("qkv", 0, 1, rows, 1536, 512), ("out / conv tap", 0, 1, rows, 512, 512), ("ffn up", 0, 1, rows, 2048, 512),
          ("ffn down", 0, 1, rows, 512, 2048), ("logits", 0, 1, rows, 6048, 512), ("d qkv", 0, 0, rows, 512, 1536), ("d out", 0, 0, rows, 512, 512),
          ("d ffn up", 0, 0, rows, 512, 2048), ("d ffn down", 0, 0, rows, 2048, 512), ("dW 512x512", 1, 0, 512, 512, rows),
          ("dW 1536x512", 1, 0, 1536, 512, rows), ("dW 2048x512", 1, 0, 2048, 512, rows), ("dW 512x2048", 1, 0, 512, 2048, rows)]
for nm, ta, tb, M, N, K in SHAPES:
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    best = 1e9
    for _ in range(3):
        for _ in range(2):
            _gemm(ta, tb, M, N, K, _p(A), A.shape[1], _p(B), B.shape[1], _p(C), N)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            _gemm(ta, tb, M, N, K, _p(A), A.shape[1], _p(B), B.shape[1], _p(C), N)
        e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / 20)
    print("%-14s ta=%d tb=%d %5d x %5d x %5d: %7.1f us %6.1f TFLOP/s" % (nm, ta, tb, M, N, K, best, 2e-6 * M * N * K / best), flush=True)
