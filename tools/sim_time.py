import torch, numpy as np, sys
sys.path.insert(0,'.')
from pykaldi2_amd import simulation
for n,k in ((192000, 8000), (192000, 1600), (560000, 8000)):
    x = torch.randn(n, device='cuda'); r = torch.randn(k, device='cuda')
    for _ in range(3): simulation.Distorter.apply_rir(x, r, 40)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): simulation.Distorter.apply_rir(x, r, 40)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/10
    print("apply_rir n=%d k=%d: %.3f ms  %.1f TFLOP/s" % (n, k, ms, 2.0*n*k/ms/1e9))
