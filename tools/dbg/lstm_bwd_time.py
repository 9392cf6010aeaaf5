import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from pykaldi2_amd import _lib
dev = torch.device("cuda")
T, B, H, D = 589, 4, 512, 2
L = _lib.lib()
gx = torch.randn(T, B, D * 4 * H, device=dev) * 0.1
whh = torch.randn(D, 4 * H, H, device=dev) * 0.04
y = torch.empty(T, B, D * H, device=dev); gates = torch.empty(D, T, B, 4 * H, device=dev); cells = torch.empty(D, T, B, H, device=dev)
_lib.check(L.pk2_lstm_layer_fwd(_lib.ptr(gx), _lib.ptr(whh), None, B, T, H, D, _lib.ptr(y), _lib.ptr(gates), _lib.ptr(cells), None, _lib.stream_ptr()))
dy = torch.randn(T, B, D * H, device=dev) * 0.1
dgx = torch.empty(T, B, D * 4 * H, device=dev)
scratch = torch.empty(L.pk2_lstm_bwd_scratch_floats(B, H, D), device=dev)
def run():
    _lib.check(L.pk2_lstm_layer_bwd(_lib.ptr(dy), _lib.ptr(whh), _lib.ptr(gates), _lib.ptr(cells), B, T, H, D, _lib.ptr(dgx), _lib.ptr(scratch), _lib.stream_ptr()))
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
print("lstm_bwd us/step %.2f" % (1e3 * e0.elapsed_time(e1) / 5 / T), flush=True)
