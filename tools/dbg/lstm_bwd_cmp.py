import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from pykaldi2_amd import _lib
dev = torch.device("cuda")
L = _lib.lib()
for B, T in ((4, 23), (8, 17), (3, 9)):
    H, D = 512, 2
    torch.manual_seed(B)
    gx = torch.randn(T, B, D * 4 * H, device=dev) * 0.5
    whh = torch.randn(D, 4 * H, H, device=dev) * 0.04
    y = torch.empty(T, B, D * H, device=dev); gates = torch.empty(D, T, B, 4 * H, device=dev); cells = torch.empty(D, T, B, H, device=dev)
    _lib.check(L.pk2_lstm_layer_fwd(_lib.ptr(gx), _lib.ptr(whh), None, B, T, H, D, _lib.ptr(y), _lib.ptr(gates), _lib.ptr(cells), None, _lib.stream_ptr()))
    dy = torch.randn(T, B, D * H, device=dev) * 0.1
    outs = []
    for mode in ("persist", "step"):
        if mode == "step": os.environ["PK2_LSTM_PERSIST_FWD_ONLY"] = "1"
        else: os.environ.pop("PK2_LSTM_PERSIST_FWD_ONLY", None)
        dgx = torch.zeros(T, B, D * 4 * H, device=dev)
        scratch = torch.empty(L.pk2_lstm_bwd_scratch_floats(B, H, D), device=dev)
        _lib.check(L.pk2_lstm_layer_bwd(_lib.ptr(dy), _lib.ptr(whh), _lib.ptr(gates), _lib.ptr(cells), B, T, H, D, _lib.ptr(dgx), _lib.ptr(scratch), _lib.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(dgx.cpu().numpy().reshape(T, B, D, 4, H))
    diff = np.abs(outs[0] - outs[1])
    print("B", B, "T", T, "max diff", diff.max(), "max |step|", np.abs(outs[1]).max())
    if diff.max() > 1e-4:
        bad = np.argwhere(diff > 1e-4)
        print(" bad count", len(bad), "first", bad[:5].tolist(), "frames", sorted(set(bad[:, 0].tolist()))[:10], "rows", sorted(set(bad[:, 1].tolist())), "dirs", sorted(set(bad[:, 2].tolist())), "units%16", sorted(set((bad[:, 4] % 16).tolist()))[:16])
