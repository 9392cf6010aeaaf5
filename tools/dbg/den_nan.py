import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import chain_ref as R
from pykaldi2_amd import chain, synth
S, A, P, lens, leaky = 200, 20000, 11, [40, 25], 1e-3
g = synth.den_graph_arcs(S, A, P, S, loop_pdf_differs=True)
g["dst"][:6000] = 5
g["pdf"][:6000] = g["pdf"][0]
G = chain.DenominatorGraph(g, P)
o = G.debug_ordering(3)
print("V", o["vpdf"].shape, "atomic chunks", np.flatnonzero(o["atomic"]), "row0", o["row0"], "nrows", o["nrows"], "real0", o["real0"], "nreal", o["nreal"])
print("voff[4:8]", o["voff"][4:8])
ref = R.DenGraphRef(g["num_states"], g["src"], g["dst"], g["pdf"], g["prob"], 0, P)
rng = np.random.default_rng(1)
lg = rng.normal(0, 3, size=(2, 40, P)).astype(np.float32)
x = torch.from_numpy(lg).cuda()
lp, gamma = chain.den_forward_backward(G, x, lens, leaky)
gm = gamma.cpu().numpy()
print("lp", lp.cpu().numpy())
for n, Tn in enumerate(lens):
    want_lp, want_g, _ = R.den_forward_backward(lg[n, :Tn].astype(np.float64), ref, leaky)
    print(n, "want lp", want_lp)
    bad = np.argwhere(~np.isfinite(gm[n, :Tn]))
    print("nan count", len(bad), bad[:10])
    d = np.abs(gm[n, :Tn] - want_g)
    print("max err", np.nanmax(d), np.unravel_index(np.nanargmax(d), d.shape))
print("nan frames seq0", sorted(set(np.argwhere(~np.isfinite(gm[0]))[:, 0].tolist())))
os.environ["PK2_DEN_GAMMA_GATHER"] = "1"
lp, gamma = chain.den_forward_backward(G, x, lens, leaky)
gm = gamma.cpu().numpy()
print("gather variant nan frames seq0", sorted(set(np.argwhere(~np.isfinite(gm[0]))[:, 0].tolist())))
del os.environ["PK2_DEN_GAMMA_GATHER"]
for rep in range(3):
    lp, gamma = chain.den_forward_backward(G, x, lens, leaky)
    gm = gamma.cpu().numpy()
    print("again nan frames seq0", sorted(set(np.argwhere(~np.isfinite(gm[0]))[:, 0].tolist())))
