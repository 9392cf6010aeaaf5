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
# an earlier optimiser of the same process: the next one continues the device's stamp sequence (ADVICE r5: a ring word of the
# verdict ring is never reused, so a stale verdict cannot be taken for a fresh one)
first = hvd.DistributedOptimizer(optim.Adam(Flat(torch.zeros(64).cuda()), lr=1e-2, amsgrad=True))
for _ in range(3):
    first.zero_grad(); first.step()
first.finish()
model = Flat(torch.linspace(-1.0, 1.0, 4096).cuda())
opt = hvd.DistributedOptimizer(optim.Adam(model, lr=1e-2, amsgrad=True))
assert first._guard_first == 1 and opt._guard_first == 4 and opt._guard_stamp == 3, (opt._guard_first, opt._guard_stamp)
snap, step = None, 0
try:
    for step in range(1, %(nsteps)d):
        opt.zero_grad()
        model.g.fill_(0.5 + rank)
        if step == %(fail)d:
            torch.cuda.synchronize()
            snap = model.p.clone()               # the weights before the failing step
            if rank == 1:                        # what a timed-out persistent kernel leaves: guard up, NaN gradient
                _lib.check(_lib.lib().pk2_persist_guard_raise(_lib.stream_ptr()))
                model.g.fill_(float("nan"))
        opt.step()
    step = 99
    hvd.finish()          # the verdicts of the last two steps: raised here, on both ranks, when the loop ended first
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
    script.write_text(_GUARD_SCRIPT % dict(root=root, nsteps=10, fail=3))
    got, r = _run_guard_script(script, root, 29547)
    assert got == [("0", "5", "1", "1", "1"), ("1", "5", "1", "1", "1")], r.stdout[-2000:] + r.stderr[-2000:]
    assert "NEVER RAISED" not in r.stdout


def _run_guard_script(script, root, port):
    import os
    import re
    import subprocess
    import sys
    env = dict(os.environ, PK2_HVD_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    # (the two ranks share the pipe: their lines may run into each other)
    return sorted(re.findall(r"RANK (\d) RAISED step=(\d+) same=(\d) finite=(\d) local=(\d)", r.stdout)), r


def test_guard_raised_in_the_last_step_is_caught_by_the_collective_flush(tmp_path):
    """the per-step check looks two steps back, so a time-out in the LAST step of a run was never raised (exit 0,
    checkpoint written).  Rank 1's guard goes up in the last step of a 6-step loop; hvd.finish() (epoch end / shutdown)
    must raise on BOTH ranks, with the weights of both ranks untouched by the failing step."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "guard_last.py"
    script.write_text(_GUARD_SCRIPT % dict(root=root, nsteps=7, fail=6))
    got, r = _run_guard_script(script, root, 29549)
    assert got == [("0", "99", "1", "1", "1"), ("1", "99", "1", "1", "1")], r.stdout[-2000:] + r.stderr[-2000:]
    assert "NEVER RAISED" not in r.stdout
