import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pykaldi2_amd import _lib
dev = torch.device("cuda", 0)
T, B, H, D = 589, 4, 512, 2
gx = torch.randn(T, B, D * 4 * H, device=dev) * 0.1
whh = torch.randn(D, 4 * H, H, device=dev) * 0.04
y = torch.empty(T, B, D * H, device=dev); gates = torch.empty(D, T, B, 4 * H, device=dev); cells = torch.empty(D, T, B, H, device=dev)
L = _lib.lib()
def run():
    _lib.check(L.pk2_lstm_layer_fwd(_lib.ptr(gx), _lib.ptr(whh), None, B, T, H, D, _lib.ptr(y), _lib.ptr(gates), _lib.ptr(cells), None, _lib.stream_ptr()))
def timeit(tag):
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    print("%-50s lstm_fwd us/step %.2f" % (tag, 1e3 * e0.elapsed_time(e1) / 5 / T), flush=True)
timeit("baseline (one stream)")
s = torch.cuda.Stream(device=dev)
timeit("after creating a second stream")
with torch.cuda.stream(s):
    z = torch.zeros(1024, device=dev) + 1
torch.cuda.synchronize()
timeit("after one kernel on the second stream")
ev = torch.cuda.Event(); ev.record(s); torch.cuda.current_stream().wait_event(ev)
timeit("after a cross-stream event wait")
del s, z
torch.cuda.synchronize()
timeit("after deleting the second stream")
