import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pykaldi2_amd import chain, synth
S, A, P, lens, leaky = 200, 20000, 11, [40, 25], 1e-3
g = synth.den_graph_arcs(S, A, P, S, loop_pdf_differs=True)
g["dst"][:6000] = 5
g["pdf"][:6000] = g["pdf"][0]
G = chain.DenominatorGraph(g, P)
of, ob = G.debug_ordering(3), G.debug_ordering(4)
V = of["vpdf"].shape[0]; ncf, ncb = len(of["row0"]), len(ob["row0"])
print("V", V, "ncf", ncf, "ncb", ncb, "bwd atomic", ob["atomic"])
rng = np.random.default_rng(1)
lg = rng.normal(0, 3, size=(2, 40, P)).astype(np.float32)
x = torch.from_numpy(lg).cuda()
lp, gamma = chain.den_forward_backward(G, x, lens, leaky)
torch.cuda.synchronize()
ws = chain._workspace_cache[0].cpu().numpy()
Tm, NG = 40, 4
off = 0
def take(count, dt=np.float32):
    global off
    off = (off + 255) // 256 * 256
    a = ws[off:off + count * 4].view(dt)
    off += count * 4
    return a
alpha = take(NG * (Tm + 1) * S).reshape(Tm + 1, S, NG)
alphav = take(NG * (Tm + 1) * V).reshape(Tm + 1, V, NG)
bx = take(NG * (Tm + 1) * V * 2).reshape(Tm + 1, V, 2, NG)
xs = take(NG * Tm * P); gam = take(NG * Tm * P)
apart = take(NG * (Tm + 1) * ncf).reshape(Tm + 1, ncf, NG)
bpart = take(NG * (Tm + 1) * ncb * 2).reshape(Tm + 1, ncb, 2, NG)
asum = take(NG * (Tm + 1)).reshape(Tm + 1, NG)
take(NG); take(NG); take(NG); take(NG, np.int32)
csum = take(NG * (Tm + 1) * 2).reshape(Tm + 1, 2, NG)[:, 0]; ks = take(NG * (Tm + 1)).reshape(Tm + 1, NG)
for name, a in [("alpha", alpha), ("alphav", alphav), ("btilde", bx[:, :, 0]), ("x", bx[:, :, 1]), ("bpart", bpart), ("csum", csum), ("kscale", ks), ("asum", asum)]:
    bad = np.argwhere(~np.isfinite(a[..., 0]))
    print(name, "nonfinite (seq 0):", len(bad), bad[:6].tolist())
print("csum[:6,0]", csum[:6, 0], "kscale[:6,0]", ks[:6, 0])
print("btilde[3] range", bx[3, :, 0, 0].min(), bx[3, :, 0, 0].max(), "btilde[2] first bad", np.argwhere(~np.isfinite(bx[2, :, 0, 0]))[:10].ravel())
vstate = np.repeat(np.arange(S), np.diff(of["voff"]))

for t in range(39, -1, -1):
    b = bx[t, :, 0, 0]
    print(t, "btilde min/max", b.min(), b.max(), "cu", csum[t, 0], "bpart", bpart[t, :, 0, 0].sum(), bpart[t, :, 1, 0].sum(), "asum", asum[t, 0], "x max", bx[t, :, 1, 0].max())
