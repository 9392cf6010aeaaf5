"""Horovod-shaped collective shim over torch.distributed (RCCL over xGMI on MI355X).

The reference drives data parallelism through ``horovod.torch`` (reference
bin/train_ce.py:83-131, bin/train_chain.py:106-145, bin/train_se.py:95-134,
data/dataloader.py:45-46,83-84).  This module offers the same call surface --
``init, size, rank, local_rank, broadcast_parameters, broadcast_optimizer_state,
DistributedOptimizer`` -- with one process per GPU and plain ``ncclAllReduce`` (RCCL):

* gradients are averaged with ONE all-reduce of the model's flat gradient buffer (85 MB) on the compute stream when
  backward is done (Horovod fuses tensors with a 64 MB / 5 ms heuristic instead).  PK2_HVD_OVERLAP=1 selects the
  bucketed variant -- one all-reduce per bucket (output layer, then LSTM layer 2, 1, 0: the order backward produces
  them) on a side HIP stream, overlapping the rest of backward -- which is NOT the default because on ROCm 7.2 a
  second active stream slows the graph-replayed step chains by more than the exchange costs (measured with a
  one-rank RCCL group: 35.4 -> 39.9 ms per step);
* the 1/size factor is folded into the optimiser kernel (no extra pass over the gradients);
* ``step()`` makes the compute stream wait on the side stream, so clipping acts on the averaged
  gradient (the reference clips local, possibly mid-flight gradients: SURVEY.md Appendix B#15).

Launch with ``python -m torch.distributed.run --nproc-per-node N ...`` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment).  Without those variables everything degrades to a
single process (size() == 1) and no process group is created.
"""
import os

import torch
import torch.distributed as dist

_state = dict(initialized=False, rank=0, size=1, local_rank=0, group=False)


def init(backend=None):
    if _state["initialized"]:
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # PK2_HVD_SINGLE_RANK_GROUP=1 (tests): create the process group and run every collective even with one
    # rank, so the RCCL + side-stream path can be exercised on a one-GPU box.
    forced = os.environ.get("PK2_HVD_SINGLE_RANK_GROUP") == "1" and "RANK" in os.environ
    if world > 1 or forced:
        if backend is None:    # PK2_HVD_BACKEND=gloo: tests that put several ranks on one GPU (RCCL refuses that)
            backend = os.environ.get("PK2_HVD_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        local = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        if not dist.is_initialized():
            dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=world)
        _state.update(rank=dist.get_rank(), size=dist.get_world_size(), local_rank=local, group=True)
    _state["initialized"] = True


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()
    _state.update(initialized=False, rank=0, size=1, local_rank=0, group=False)


def size():
    return _state["size"]


def rank():
    return _state["rank"]


def local_rank():
    return _state["local_rank"]


def local_device():
    """Index of this rank's GPU: the local rank (modulo the GPU count, which only matters when a test puts several
    ranks on one GPU)."""
    return _state["local_rank"] % max(1, torch.cuda.device_count())


def _collective():
    """True when gradients have to be exchanged (more than one rank, or the forced single-rank group)."""
    return _state["group"]


def broadcast_parameters(params, root_rank=0):
    """params: a state_dict or an iterable of (name, tensor) (reference bin/train_ce.py:127)."""
    if not _collective():
        return
    items = params.items() if hasattr(params, "items") else params
    for _, t in sorted(items, key=lambda kv: kv[0]):
        if torch.is_tensor(t):
            dist.broadcast(t, src=root_rank)


def broadcast_optimizer_state(optimizer, root_rank=0):
    """Broadcasts tensors held in the optimiser state (reference bin/train_ce.py:128).  State created
    lazily on the first step (the usual case at start-up) needs no exchange."""
    if not _collective():
        return
    sd = optimizer.state_dict()

    def walk(o):
        if torch.is_tensor(o):
            dist.broadcast(o, src=root_rank)
        elif isinstance(o, dict):
            for k in sorted(o, key=str):
                walk(o[k])
        elif isinstance(o, (list, tuple)):
            for v in o:
                walk(v)
    walk(sd.get("state", sd))


def allreduce_(tensor, average=True):
    if _collective():
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
        if average:
            tensor.div_(size())
    return tensor


class DistributedOptimizer:
    """hvd.DistributedOptimizer(optimizer, named_parameters=...) (reference bin/train_ce.py:131).

    Two kinds of wrapped optimiser:
    * a pykaldi2_amd.optim optimiser over a model with flat buffers: bucketed, stream-overlapped
      all-reduce driven by the model's bucket hook, averaging folded into the update kernel;
    * any torch.optim optimiser: gradients are all-reduced (averaged) per parameter inside step().
    """

    def __init__(self, optimizer, named_parameters=None):
        self._opt = optimizer
        self._named = list(named_parameters) if named_parameters is not None else None
        self._flat = hasattr(optimizer, "model") and hasattr(optimizer.model, "flat_parameters")
        self._handles = []
        self._side = None
        # PK2_HVD_OVERLAP=1: all-reduce every bucket on a side stream as soon as backward has produced it.  Default:
        # ONE all-reduce of the whole flat gradient buffer on the compute stream when backward is done -- on ROCm 7.2
        # a second stream that is active next to the graph-replayed LSTM / denominator step chains slows them by more
        # than the 84 MB exchange costs (one rank, RCCL, side stream: 35.4 -> 39.9 ms per step).
        self._overlap = os.environ.get("PK2_HVD_OVERLAP") == "1"
        if self._flat and _collective():
            if self._overlap:
                optimizer.model._bucket_hook = self._on_bucket
            optimizer.grad_scale = 1.0 / size()
        self._pending = set()
        self._reduced = False

    def __getattr__(self, name):
        return getattr(self._opt, name)

    # bucket finished on the compute stream -> all-reduce it on the side stream
    def _on_bucket(self, name, grad_slice):
        if grad_slice.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=grad_slice.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(grad_slice.device))
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                self._handles.append(dist.all_reduce(grad_slice, op=dist.ReduceOp.SUM, async_op=True))
        else:
            self._handles.append(dist.all_reduce(grad_slice, op=dist.ReduceOp.SUM, async_op=True))
        self._pending.add(name)

    def synchronize(self):
        if self._flat and _collective() and not self._overlap and not self._reduced:
            dist.all_reduce(self._opt.model.flat_parameters()[1], op=dist.ReduceOp.SUM)    # the flat gradient buffer
            self._reduced = True
        for h in self._handles:
            h.wait()
        self._handles = []
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        self._pending.clear()

    def zero_grad(self, *a, **k):
        self._reduced = False
        return self._opt.zero_grad(*a, **k)

    def measure_grad_norm(self, max_norm):
        self.synchronize()  # the norm is taken over the summed gradient
        return self._opt.measure_grad_norm(max_norm)

    def step(self, *a, **k):
        if _collective():
            if self._flat:
                self.synchronize()
            else:
                params = [p for g in self._opt.param_groups for p in g["params"] if p.grad is not None]
                for p in params:
                    dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
                    p.grad.div_(size())
        return self._opt.step(*a, **k)
