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


def test_collective_guard_entries_through_the_c_abi():
    """Round 5: the guard word of every rank travels with the gradients.  export -> (all-reduce, here
    pk2_allreduce_guarded on a one-rank communicator) -> import raises the local guard from the combined word and
    publishes the verdict of the step for the host."""
    L = _lib.lib()
    torch.cuda.set_device(0)
    uid = C.create_string_buffer(L.pk2_comm_unique_id_bytes())
    _lib.check(L.pk2_comm_unique_id(uid))
    h = C.c_void_p()
    _lib.check(L.pk2_comm_init(0, 1, uid, C.byref(h)))
    slot = torch.full((1,), 7.0, device="cuda")
    g = torch.arange(1000, dtype=torch.float32, device="cuda")
    ready, raised = C.c_uint32(9), C.c_uint32(9)
    try:
        _lib.check(L.pk2_persist_guard_clear())
        _lib.check(L.pk2_persist_guard_export(_lib.ptr(slot), _lib.stream_ptr()))
        _lib.check(L.pk2_allreduce_guarded(h, _lib.ptr(g), g.numel(), _lib.ptr(slot), _lib.stream_ptr()))
        _lib.check(L.pk2_persist_guard_import(_lib.ptr(slot), 1, _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert slot.item() == 0.0 and torch.equal(g, torch.arange(1000, dtype=torch.float32, device="cuda"))
        _lib.check(L.pk2_persist_guard_verdict(1, C.byref(ready), C.byref(raised)))
        assert (ready.value, raised.value) == (1, 0) and not _lib.persist_guard_raised()
        _lib.check(L.pk2_persist_guard_verdict(2, C.byref(ready), C.byref(raised)))
        assert ready.value == 0                                  # step 2 has not been imported yet
        # a peer's raised guard arrives as a non-zero combined word: the local guard goes up, the verdict says so
        slot.fill_(1.0)
        _lib.check(L.pk2_persist_guard_import(_lib.ptr(slot), 2, _lib.stream_ptr()))
        torch.cuda.synchronize()
        _lib.check(L.pk2_persist_guard_verdict(2, C.byref(ready), C.byref(raised)))
        assert (ready.value, raised.value) == (1, 1) and _lib.persist_guard_raised()
        _lib.check(L.pk2_persist_guard_export(_lib.ptr(slot), _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert slot.item() == 1.0                                # and this rank now reports it to the others
        _lib.check(L.pk2_persist_guard_verdict(1, C.byref(ready), C.byref(raised)))
        assert (ready.value, raised.value) == (1, 0)             # the ring keeps the earlier step's verdict
        assert L.pk2_persist_guard_import(_lib.ptr(slot), 0, _lib.stream_ptr()) < 0
    finally:
        _lib.check(L.pk2_persist_guard_clear())
        _lib.check(L.pk2_comm_destroy(h))
    assert not _lib.persist_guard_raised()


_GUARD_SCRIPT = r'''
import os, sys, torch
sys.path.insert(0, %(root)r)
from pykaldi2_amd import _lib, hvd, optim
hvd.init()
rank = hvd.rank()
torch.cuda.set_device(0)
class Flat:
    def __init__(self, p): self.p, self.g = p, torch.zeros_like(p)
    def flat_parameters(self): return self.p, self.g
    def parameters(self): return [self.p]
model = Flat(torch.linspace(-1.0, 1.0, 4096).cuda())
opt = hvd.DistributedOptimizer(optim.Adam(model, lr=1e-2, amsgrad=True))
snap, step = None, 0
try:
    for step in range(1, 10):
        opt.zero_grad()
        model.g.fill_(0.5 + rank)
        if step == 3:
            torch.cuda.synchronize()
            snap = model.p.clone()               # the weights before the failing step
            if rank == 1:                        # what a timed-out persistent kernel leaves: guard up, NaN gradient
                _lib.check(_lib.lib().pk2_persist_guard_raise(_lib.stream_ptr()))
                model.g.fill_(float("nan"))
        opt.step()
    print("RANK %%d NEVER RAISED" %% rank, flush=True)
except _lib.Pk2Error as e:
    torch.cuda.synchronize()
    print("RANK %%d RAISED step=%%d same=%%d finite=%%d local=%%d msg=%%s" %% (
        rank, step, int(torch.equal(model.p, snap)), int(bool(torch.isfinite(model.p).all())),
        int(_lib.persist_guard_raised()), str(e)[:60]), flush=True)
'''


def test_guard_of_one_rank_stops_every_rank_at_the_same_step(tmp_path):
    """rank 1's persistent kernel 'times out' in step 3 (guard raised on the device, NaN gradient).
    The all-reduce hands the NaN to rank 0, whose own guard is down -- the combined guard word, exchanged with the
    gradients, must raise rank 0's guard before its optimiser kernel runs (weights of BOTH ranks stay what they were before
    step 3), and both ranks must raise Pk2Error in the same step (5: the verdict of step 3), not rank 1 alone.  Two ranks
    on the one GPU of the box through the gloo backend (RCCL refuses two ranks on a device)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "guard2.py"
    script.write_text(_GUARD_SCRIPT % dict(root=root))
    env = dict(os.environ, PK2_HVD_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29547", str(script)],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    import re
    # (the two ranks share the pipe: their lines may run into each other)
    got = sorted(re.findall(r"RANK (\d) RAISED step=(\d+) same=(\d) finite=(\d) local=(\d)", r.stdout))
    assert got == [("0", "5", "1", "1", "1"), ("1", "5", "1", "1", "1")], r.stdout[-2000:] + r.stderr[-2000:]
    assert "NEVER RAISED" not in r.stdout
