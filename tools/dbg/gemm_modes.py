"""Which schedule of pk2_gemm_f32 is fastest on a shape (debug): every mode measured several times, interleaved; the
best time of each mode is reported."""
import os
import sys

import torch
from pykaldi2_amd.lstm import _gemm, _p

dev = torch.device("cuda")
MODES = {
    "default": {},
    "no_bands": {"PK2_GEMM_BANDS": "0"},
    "plain64": {"PK2_GEMM_SPLITK": "1", "PK2_GEMM_TILES": "1"},
    "plain128": {"PK2_GEMM_SPLITK": "1", "PK2_GEMM_TILES": "2"},
}
KEYS = ("PK2_GEMM_BANDS", "PK2_GEMM_SPLITK", "PK2_GEMM_TILES")


def measure(ta, tb, M, N, K, reps=20):
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    best = {m: 1e9 for m in MODES}
    for _ in range(4):
        for name, env in MODES.items():
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            for _ in range(2):
                _gemm(ta, tb, M, N, K, _p(A), A.shape[1], _p(B), B.shape[1], _p(C), N)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                _gemm(ta, tb, M, N, K, _p(A), A.shape[1], _p(B), B.shape[1], _p(C), N)
            e1.record(); torch.cuda.synchronize()
            best[name] = min(best[name], 1e3 * e0.elapsed_time(e1) / reps)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print("ta=%d tb=%d M=%5d N=%5d K=%5d tiles=%4d | " % (ta, tb, M, N, K, tiles) +
          "  ".join("%s %6.1f" % (m, best[m]) for m in MODES), flush=True)


SHAPES = [(0, 1, 2356, 4096, 1024), (0, 1, 2356, 6048, 1024), (1, 0, 6048, 1024, 2356), (0, 0, 2356, 1024, 6048),
          (1, 0, 4096, 1024, 2356), (1, 0, 2048, 512, 2352), (0, 0, 2356, 1024, 4096), (0, 1, 2356, 4096, 80),
          (1, 0, 4096, 80, 2356), (0, 1, 2300, 512, 512), (0, 0, 2300, 512, 2048), (0, 1, 2300, 2048, 512), (1, 0, 512, 512, 2300),
          (1, 0, 2048, 512, 20480), (1, 0, 4096, 1024, 20480), (0, 1, 20480, 4096, 1024), (1, 0, 5768, 1024, 20480),
          (0, 1, 2300, 1536, 512)]
for sh in SHAPES:
    measure(*sh)
