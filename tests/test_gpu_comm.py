"""The gradient exchange behind the C ABI (include/pk2hip.h: pk2_comm_*, pk2_allreduce_bucket) on the GPU box: a
one-rank RCCL communicator (the box has one GPU; RCCL refuses two ranks on a device), driven directly and through hvd."""
import ctypes as C

import pytest
import torch

from pykaldi2_amd import _lib

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_allreduce_through_the_c_abi():
    L = _lib.lib()
    n = L.pk2_comm_unique_id_bytes()
    assert n == 128
    uid = C.create_string_buffer(n)
    _lib.check(L.pk2_comm_unique_id(uid))
    assert any(uid.raw)
    torch.cuda.set_device(0)
    h = C.c_void_p()
    _lib.check(L.pk2_comm_init(0, 1, uid, C.byref(h)))
    r, w, path = C.c_int32(-1), C.c_int32(-1), C.create_string_buffer(256)
    _lib.check(L.pk2_comm_info(h, C.byref(r), C.byref(w), path, 256))
    assert (r.value, w.value) == (0, 1) and b"rccl" in path.value
    g = torch.randn(21_000_000, device="cuda")           # the size of the BLSTM's flat gradient buffer
    want = g.clone()
    side = torch.cuda.Stream()
    # a bucket on a side stream fenced by events, the rest on the compute stream -- the two schedules of hvd.py
    ev = torch.cuda.Event(); ev.record()
    side.wait_event(ev)
    _lib.check(L.pk2_allreduce_bucket(h, C.c_void_p(g[:6_000_000].data_ptr()), 6_000_000, C.c_void_p(side.cuda_stream)))
    _lib.check(L.pk2_allreduce_bucket(h, C.c_void_p(g[6_000_000:].data_ptr()), g.numel() - 6_000_000, _lib.stream_ptr()))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(g, want)                           # the sum over one rank
    assert L.pk2_allreduce_bucket(h, None, 4, None) < 0   # bad arguments fail loudly
    _lib.check(L.pk2_comm_destroy(h))
