import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.argv = ["bench.py", "--se", "--steps", "3", "--warmup", "3", "--no-cpu-baseline"]
import bench
from pykaldi2_amd import lattice
orig = lattice.MappedLatticeFasterRecognizer._workspace
def ws(self, nbytes, dev):
    a0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    sl = self.__dict__.get("_ws_slots", [None, None])
    sys.stderr.write("[ws] before: " + " | ".join("none" if c is None else "%.2f GB refs %d dev %s" % (c.numel() / 2**30, sys.getrefcount(c) - 1, c.device) for c in sl) + " max %.2f dev %s\n" % (getattr(self, "_ws_max", 0) / 2**30, dev))
    t = orig(self, nbytes, dev)
    a1 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    sys.stderr.write("[ws] need %.2f GB got %.2f GB grow %s allocs %d\n" % (nbytes / 2**30, t.numel() / 2**30, getattr(self, "_grow", None), a1 - a0))
    return t
lattice.MappedLatticeFasterRecognizer._workspace = ws
real_empty = torch.empty
def emp(*a, **k):
    dev = k.get("device")
    if dev is not None and str(dev).startswith("cuda"):
        a0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
        t = real_empty(*a, **k)
        a1 = torch.cuda.memory_stats().get("num_device_alloc", 0)
        if a1 != a0:
            import traceback
            sys.stderr.write("[alloc] %.3f GB at %s\n" % (t.numel() * t.element_size() / 2**30, traceback.extract_stack(limit=4)[0]))
        return t
    return real_empty(*a, **k)
torch.empty = emp
bench.main()
