"""Step 0 of the bf16x3 GEMM: is a 3-way bf16 split with six products and f32 accumulation as
accurate as the f32 product?  CPU only.

x = hi + mid + lo exactly (hi = bf16_rne(x), mid = bf16_rne(x - hi), lo = bf16_rne(x - hi - mid): 3 x 8 significant bits
cover the 24 of an f32; each difference is exact in f32).  Products kept: hh, hm, mh, mm, hl, lh; dropped: ml, lm, ll
(each <= 2^-24 |a b|).  The product of two bf16 numbers is exact in f32, so an f32 matmul of the parts (converted back
to f32) models the MFMA's exact products; the accumulation order of the CPU BLAS stands in for the hardware's (the MFMA
adds 16 products per instruction, rounding behaviour measured on the GPU by test_gemm_f32_matches_float64).

Shapes and value distributions: the model's (BLSTM input projection 2276 x 4096 x 1024 with tanh*sigmoid activations and
U(-1/sqrt(H), 1/sqrt(H)) weights; the CE output layer; a weight gradient over K = 20480 frames) plus N(0,1) x N(0,1).
"""
import sys
import numpy as np
import torch

torch.manual_seed(0)
torch.set_num_threads(8)


def split3(x):
    hi = x.to(torch.bfloat16).to(torch.float32)
    r1 = x - hi
    mid = r1.to(torch.bfloat16).to(torch.float32)
    r2 = r1 - mid
    lo = r2.to(torch.bfloat16).to(torch.float32)
    return hi, mid, lo, (r2 - lo)


def study(name, a, b):
    ref = a.double() @ b.double()
    scale = (a.double().abs() @ b.double().abs())          # sum |a b| per output: the natural error unit
    f32 = (a @ b).double()
    ah, am, al, ar = split3(a)
    bh, bm, bl, br = split3(b)
    assert float(ar.abs().max()) == 0.0 and float(br.abs().max()) == 0.0, "3-way split is not exact"
    # six products, small terms first into one f32 accumulator chain (as the kernel orders them)
    six32 = ((al @ bh + ah @ bl) + (am @ bm)) + (am @ bh + ah @ bm) + ah @ bh
    six64 = (al.double() @ bh.double() + ah.double() @ bl.double() + am.double() @ bm.double()
             + am.double() @ bh.double() + ah.double() @ bm.double() + ah.double() @ bh.double())
    three32 = (am @ bh + ah @ bm) + ah @ bh
    rows = []
    for label, got in (("f32 sgemm", f32), ("bf16x3 six products, f32 acc", six32.double()),
                       ("bf16x3 six products, exact acc (dropped terms only)", six64),
                       ("bf16x2-like three products, f32 acc", three32.double())):
        e = (got - ref).abs()
        rows.append((label, float((e / scale).max()), float(torch.sqrt(((e / scale) ** 2).mean())),
                     float(e.max() / ref.abs().max())))
    print("%s  [%d x %d x %d]" % (name, a.shape[0], b.shape[1], a.shape[1]))
    for r in rows:
        print("   %-58s max err/sum|ab| %.3e   rms %.3e   max err/max|c| %.3e" % r)
    return rows


def main():
    H = 512
    out = []
    act = torch.tanh(torch.randn(2276, 1024)) * torch.sigmoid(torch.randn(2276, 1024))
    w = (torch.rand(1024, 4096) * 2 - 1) / H ** 0.5
    out.append(study("BLSTM input projection (activations x W_ih^T)", act, w))
    wout = (torch.rand(1024, 6048) * 2 - 1) / 1024 ** 0.5
    out.append(study("output layer (activations x W_out^T)", act, wout))
    dg = torch.randn(20480, 1024) * 1e-3 * torch.rand(20480, 1).pow(4)
    a2 = torch.tanh(torch.randn(20480, 512)) * torch.sigmoid(torch.randn(20480, 512))
    out.append(study("weight gradient over K = 20480 frames (dG^T x activations)", dg.t().contiguous(), a2))
    out.append(study("N(0,1) x N(0,1)", torch.randn(1024, 4096), torch.randn(4096, 1024)))
    fb = torch.randn(2276, 80) * 4 + 10
    w0 = (torch.rand(80, 4096) * 2 - 1) / H ** 0.5
    out.append(study("layer-0 projection (fbank x W_ih^T, K = 80)", fb, w0))
    ok = all(r[1][1] <= 1.05 * r[0][1] and r[1][2] <= 1.05 * r[0][2] for r in out)
    print("gate (six products, f32 acc: max and rms error <= the f32 product's): %s" % ("PASS" if ok else "FAIL"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
