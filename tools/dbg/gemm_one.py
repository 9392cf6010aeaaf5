"""Time of one product under both arithmetic paths: python tools/dbg/gemm_one.py ta tb M N K"""
import sys
sys.path.insert(0, "tools/dbg")
from gemm_arith import t_us
ta, tb, M, N, K = [int(v) for v in sys.argv[1:6]]
f, x = t_us(ta, tb, M, N, K, 0), t_us(ta, tb, M, N, K, 1)
fl = 2.0 * M * N * K * 1e-6
print("ta=%d tb=%d %6d x %5d x %6d: f32 %8.1f us %6.1f TF/s | bf16x3 %8.1f us %6.1f TF/s-equiv" % (ta, tb, M, N, K, f, fl / f, x, fl / x))
