"""Loss operators with the reference's names and calling conventions (reference ops/ops.py).

  ChainObjtiveFunction.apply(loglikes, den_graph, supervision, chain_opts)   ops/ops.py:243-280
  ChainObjtiveBatch.apply(prediction, den_graph, supervisions, chain_opts)   the batched form
  CrossEntropy(logits, targets, ignore_index, reduction)                     nn.CrossEntropyLoss

Value / gradient convention kept from the reference: forward returns the
*objective* (log p_num - log p_den, to be maximised) and backward returns
minus its derivative regardless of ``grad_out`` (ops/ops.py:276-280), so
``loss.backward()`` followed by a descent step increases the objective.  Unlike
the reference, nothing is copied to the host and the saved tensor is not
modified in place.

MMIFunction / sMBRFunction (lattice-based, ops/ops.py:41-75,119-156) need an
on-the-fly WFST decoder; they are scoped for a later round (SURVEY.md 8(f)
rank 3) and raise NotImplementedError rather than silently doing something else.
"""
import numpy as np
import torch
from torch.autograd import Function

from . import _lib, chain


class ChainObjtiveFunction(Function):
    """Per-utterance LF-MMI (the reference's spelling).  loglikes: [T', P] CUDA f32."""

    @staticmethod
    def forward(ctx, loglikes, den_graph, supervision, chain_opts):
        out, grad = chain.compute_chain_objf_and_deriv(chain_opts, den_graph, [supervision],
                                                       loglikes.detach().unsqueeze(0))
        ctx.save_for_backward(grad.squeeze(0))
        return out[0, 0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        grad_input, = ctx.saved_tensors
        return -grad_input, None, None, None


class ChainObjtiveBatch(Function):
    """All utterances of a minibatch in one call: prediction [B, T', P] (any strides with a unit
    pdf stride), supervisions = list of chain.Supervision.  Returns the summed objective, i.e.
    what the reference's per-utterance loop accumulates (bin/train_chain.py:261-275)."""

    @staticmethod
    def forward(ctx, prediction, den_graph, supervisions, chain_opts):
        out, grad = chain.compute_chain_objf_and_deriv(chain_opts, den_graph, supervisions,
                                                       prediction.detach())
        ctx.save_for_backward(grad)
        ctx.per_sequence = out
        return out[0].sum()

    @staticmethod
    def backward(ctx, grad_out):
        grad_input, = ctx.saved_tensors
        return -grad_input, None, None, None


class _CrossEntropyFunction(Function):
    @staticmethod
    def forward(ctx, logits, targets, ignore_index, reduction):
        _lib.require_gpu()
        P = logits.shape[-1]
        x = logits.detach()
        # rows in memory order (works for the time-major views LSTMAM returns)
        if x.dim() == 3 and not x.is_contiguous() and x.transpose(0, 1).is_contiguous():
            x2 = x.transpose(0, 1).reshape(-1, P)
            tg = targets.transpose(0, 1).reshape(-1)
            layout = "tm"
        else:
            x2 = x.contiguous().view(-1, P)
            tg = targets.reshape(-1)
            layout = "bm"
        tg = tg.to(device=x.device, dtype=torch.int64).contiguous()
        rows = x2.shape[0]
        grad = torch.empty_like(x2)
        acc = torch.empty(2, dtype=torch.float32, device=x.device)
        cnt = acc[1:2].view(torch.int32)
        L = _lib.lib()
        _lib.check(L.pk2_softmax_ce_fwd_bwd(_lib.ptr(x2), P, _lib.ptr(tg), int(ignore_index), rows, P,
                                            _lib.ptr(acc), _lib.ptr(cnt), _lib.ptr(grad), P, None,
                                            _lib.stream_ptr(x.device)))
        if reduction == "mean":
            _lib.check(L.pk2_scale_by_count(_lib.ptr(grad), grad.numel(), 1.0, _lib.ptr(cnt),
                                            _lib.stream_ptr(x.device)))
            loss = acc[0] / cnt.to(torch.float32).clamp_min(1.0)[0]
        else:
            loss = acc[0].clone()
        g = grad.view(x.shape[1], x.shape[0], P).transpose(0, 1) if layout == "tm" else grad.view(x.shape)
        ctx.save_for_backward(g)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        g, = ctx.saved_tensors
        return g * grad_out, None, None, None


class CrossEntropyLoss(torch.nn.Module):
    """nn.CrossEntropyLoss(ignore_index=-100, reduction='mean'|'sum') as the reference uses it
    (bin/train_ce.py:134,189; bin/train_se.py:214,235), fused into one HIP kernel."""

    def __init__(self, ignore_index=-100, reduction="mean"):
        super().__init__()
        assert reduction in ("mean", "sum")
        self.ignore_index, self.reduction = ignore_index, reduction

    def forward(self, logits, targets):
        return _CrossEntropyFunction.apply(logits, targets, self.ignore_index, self.reduction)


def _lattice_op(name):
    class _Unavailable(Function):
        @staticmethod
        def forward(ctx, *args):
            raise NotImplementedError(
                "%s needs on-the-fly lattice generation (a WFST beam-search decoder over HCLG); it is "
                "scheduled after the LF-MMI path (SURVEY.md 8(f) rank 3)" % name)
    _Unavailable.__name__ = name
    return _Unavailable


MMIFunction = _lattice_op("MMIFunction")
sMBRFunction = _lattice_op("sMBRFunction")
