"""Host enqueue time against GPU time of the TransformerAM LF-MMI step (configs[4], bench.py --transformer's Trainer):
is the step bound by the ~470 launches the host has to issue, or by the kernels?  One GPU, no arguments."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from pykaldi2_amd import chain
dev = torch.device('cuda', 0)
g = bench.den_graph_arrays()
den = chain.DenominatorGraph(g, bench.P)
rng = np.random.default_rng(1234)
batches = bench.make_batches(rng, 8, 4, dev)
tr = bench.Trainer(dev, den, arch="transformer")
for i in range(4): tr.step(batches[i])
torch.cuda.synchronize()
for rep in range(2):
    host = []
    t0 = time.perf_counter()
    for i in range(16):
        a = time.perf_counter(); tr.prefetch(batches[(i + 1) % 8]); tr.step(batches[i % 8]); host.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue per step ms: mean %.2f min %.2f max %.2f | loop %.1f ms, drain %.1f ms, total per step %.2f"
          % (1e3*np.mean(host), 1e3*min(host), 1e3*max(host), 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t2-t0)/16))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(8): tr.step(batches[i % 8])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(30)
