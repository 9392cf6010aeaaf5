"""Persistent denominator kernel against the launch-per-frame kernels and the oracle (GPU debug / timing aid)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pykaldi2_amd import chain, synth

def run(G, x, lens, leaky, persist):
    os.environ["PK2_DEN_PERSIST"] = "1" if persist else "0"
    lp, gamma = chain.den_forward_backward(G, x, lens, leaky)
    torch.cuda.synchronize()
    return lp.cpu().numpy(), gamma.cpu().numpy()

def small():
    from oracle import chain_ref as R
    for (S, A, P, lens, kw) in [(200, 6000, 11, [40, 25], dict(loop_pdf_differs=True)),
                                (300, 20000, 23, [33, 7, 50, 21, 50], dict(loop_pdf_differs=True, multi_entry_frac=0.3)),
                                (1500, 60000, 64, [61], {}),]:
        g = synth.den_graph_arcs(S, A, P, 3, **kw)
        G = chain.DenominatorGraph(g, P)
        ref = R.DenGraphRef(g["num_states"], g["src"], g["dst"], g["pdf"], g["prob"], 0, P)
        rng = np.random.default_rng(1)
        T = max(lens)
        lg = rng.normal(0, 2, size=(len(lens), T, P)).astype(np.float32)
        x = torch.from_numpy(lg).cuda()
        lp1, g1 = run(G, x, lens, 1e-3, True)
        lp0, g0 = run(G, x, lens, 1e-3, False)
        errs = []
        for n, Tn in enumerate(lens):
            want_lp, want_g, _ = R.den_forward_backward(lg[n, :Tn].astype(np.float64), ref, 1e-3)
            errs.append((abs(float(lp1[n]) - want_lp), float(np.abs(g1[n, :Tn] - want_g).max()), float(np.abs(g0[n, :Tn] - want_g).max())))
        print("small", S, A, lens, "lp persist", lp1, "lp frames", lp0, "| (lp err, gamma err persist, gamma err frames):", errs, flush=True)

def big():
    import bench
    g = bench.den_graph_arrays()
    G = chain.DenominatorGraph(g, g["num_pdfs"])
    lens = bench.DEN_ROOF_LENS
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.normal(0, 2, size=(len(lens), max(lens), g["num_pdfs"])).astype(np.float32)).cuda()
    res = {}
    for persist in (True, False, True):
        lp, gm = run(G, x, lens, 1e-4, persist)
        t0 = time.time()
        for _ in range(5): chain.den_forward_backward(G, x, lens, 1e-4)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / 5 * 1e3
        res[persist] = (lp, gm)
        print("big persist" if persist else "big frames ", "lp", lp, "ms/call %.2f" % ms, flush=True)
    print("big: max |gamma diff|", float(np.abs(res[True][1] - res[False][1]).max()), "lp diff", np.abs(res[True][0] - res[False][0]).max())

if __name__ == "__main__":
    which = sys.argv[1:] or ["small", "big"]
    if "small" in which: small()
    if "big" in which: big()
