"""Which parameter gradients differ between the one-stream and the side-stream backward of TransformerAM? (debug)"""
import copy, os, sys
import torch
sys.path.insert(0, ".")
from pykaldi2_amd import transformer

torch.manual_seed(0)
cfg = dict(D=80, C=512, H=8, FF=2048, L=2, P=301, T=48, B=2)
m = transformer.TransformerAM(cfg["D"], cfg["C"], cfg["H"], cfg["FF"], cfg["L"], 0.0, cfg["P"]).cuda().train()
x = torch.randn(cfg["T"], cfg["B"], cfg["D"]).cuda()
kpm = torch.zeros(cfg["B"], cfg["T"], dtype=torch.bool).cuda()
w = torch.randn(cfg["T"], cfg["B"], cfg["P"]).cuda()

def run(side):
    os.environ["PK2_TR_SIDE_STREAM"] = "1" if side else "0"
    y = m(x, None, kpm)
    (y * w).sum().backward()
    torch.cuda.synchronize()
    out = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    out["__y__"] = y.detach().clone()
    return out

ref = run(False)
bad = {}
for it in range(30):
    got = run(os.environ.get('DBG_SIDE', '1') == '1')
    for n in ref:
        e = (got[n] - ref[n]).abs().max().item() / max(1e-6, ref[n].abs().max().item())
        if e > 1e-4:
            bad.setdefault(n, []).append((it, round(e, 4)))
print("params that differed at least once:", len(bad))
for n, v in bad.items():
    print(n, v[:6])
