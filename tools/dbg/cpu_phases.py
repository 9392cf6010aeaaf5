"""Host-side time of the phases of one LF-MMI step of bench.py (no synchronisation inside the steps): where does the host
spend the time in which the GPU idles between two steps?  python tools/dbg/cpu_phases.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from pykaldi2_amd import chain, ops, optim, utils  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
den = chain.DenominatorGraph(bench.den_graph_arrays(), bench.P)
rng = np.random.default_rng(1234)
batches = bench.make_batches(rng, 8, 4, dev, None)
tr = bench.Trainer(dev, den)
for i in range(4):
    tr.step(batches[i % 8])
torch.cuda.synchronize()
names = ["prefetch", "fbank", "pad", "forward", "sups", "chain", "zero_grad", "backward", "clip", "opt"]
acc = {k: [] for k in names}
t_all0 = time.perf_counter()
for i in range(steps):
    mb = batches[(4 + i) % 8]
    t = [time.perf_counter()]
    if i + 1 < steps:
        tr.prefetch(batches[(4 + i + 1) % 8])
    t.append(time.perf_counter())
    feats, frames, row_off = tr.fb(mb["wav"], mb["lens"])
    t.append(time.perf_counter())
    x = tr.fb.pad_roll_subsample(feats, row_off, frames, shift=0, subsample=3, time_major=True)
    t.append(time.perf_counter())
    logits = tr.model.forward_time_major(x)
    t.append(time.perf_counter())
    fut = tr._pending.pop(id(mb), None)
    sups = fut.result() if fut is not None else bench.build_supervisions(mb["alis"])
    t.append(time.perf_counter())
    loss = ops.ChainObjtiveBatch.apply(logits.transpose(0, 1), tr.den, sups, tr.opts)
    t.append(time.perf_counter())
    tr.opt.zero_grad()
    t.append(time.perf_counter())
    loss.backward()
    t.append(time.perf_counter())
    tr.step_no += 1
    lr = utils.noam_decay(tr.step_no, 4000, tr.base_lr)
    for grp in tr.opt.param_groups:
        grp["lr"] = lr
    norm = optim.clip_grad_norm_(tr.opt, 5.0)
    t.append(time.perf_counter())
    tr.opt.step()
    t.append(time.perf_counter())
    for k, a, b in zip(names, t[:-1], t[1:]):
        acc[k].append(1e3 * (b - a))
torch.cuda.synchronize()
wall = 1e3 * (time.perf_counter() - t_all0) / steps
print("wall %.3f ms per step over %d steps; host time per phase (ms, median | max):" % (wall, steps))
tot = 0.0
for k in names:
    v = np.array(acc[k]); tot += np.median(v)
    print("  %-10s %7.3f | %7.3f" % (k, np.median(v), v.max()))
print("  host sum of medians %.3f ms" % tot)
