import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import chain_ref as R
from pykaldi2_amd import chain, synth
S, A, P, lens = 200, 20000, 11, [40, 25]
for nred in (6000, 5000, 4200):
  for leaky in (1e-3, 1e-2):
    for kw in (dict(loop_pdf_differs=True), dict(loop_pdf_differs=True, multi_entry_frac=0.3)):
        g = synth.den_graph_arcs(S, A, P, S, **kw)
        g["dst"][:nred] = 5
        g["pdf"][:nred] = g["pdf"][0]
        G = chain.DenominatorGraph(g, P)
        ref = R.DenGraphRef(g["num_states"], g["src"], g["dst"], g["pdf"], g["prob"], 0, P)
        rng = np.random.default_rng(1)
        lg = rng.normal(0, 3, size=(2, 40, P)).astype(np.float32)
        x = torch.from_numpy(lg).cuda()
        res = []
        for mode in ("sx", "general"):
            os.environ["PK2_DEN_MODE"] = mode
            lp, gamma = chain.den_forward_backward(G, x, lens, leaky)
            gm = gamma.cpu().numpy()
            errs = []
            for n, Tn in enumerate(lens):
                want_lp, want_g, _ = R.den_forward_backward(lg[n, :Tn].astype(np.float64), ref, leaky)
                errs.append(float(np.abs(gm[n, :Tn] - want_g).max()))
            res.append((mode, errs))
        print(nred, leaky, sorted(kw), res, flush=True)
