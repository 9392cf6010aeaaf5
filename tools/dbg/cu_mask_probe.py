"""Where do the workgroups of a launch on a CU-masked stream run?  (pk2_stream_create_cu_mask / pk2_debug_where)"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pykaldi2_amd import _lib

L = _lib.lib()
for k in (0, 4, 8, 16):
    if k:
        h = C.c_void_p()
        _lib.check(L.pk2_stream_create_cu_mask(k, C.byref(h)))
        st = h
    else:
        st = _lib.stream_ptr()
    out = torch.full((2048 * 3,), -1, dtype=torch.int32, device="cuda")
    _lib.check(L.pk2_debug_where(_lib.ptr(out), 2048, st))
    torch.cuda.synchronize()
    a = out.cpu().numpy().reshape(-1, 3)
    cus = {}
    for x, se, cu in a:
        cus.setdefault(int(x), set()).add((int(se), int(cu)))
    print("cus_per_xcd=%d: XCCs used %s, distinct (SE, CU) per XCC %s" % (k, sorted(cus), [len(cus[x]) for x in sorted(cus)]))
    if k:
        _lib.check(L.pk2_stream_destroy(h))
