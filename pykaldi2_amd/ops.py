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

MMIFunction / sMBRFunction (lattice-based, ops/ops.py:41-75,119-156) decode on the device
(lattice.MappedLatticeFasterRecognizer) and run the lattice forward-backward there; LatticeBatchFunction
is the whole-minibatch form.
"""
import numpy as np
import os

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
        # (round 6: the summed objective and the negated gradient come out of the library call -- out[0].sum() and
        # -grad_input were a reduction, two fills and a [N, T, P] elementwise kernel of torch's in the middle of every step)
        N = len(supervisions)
        out, neg_grad = chain.compute_chain_objf_and_deriv(chain_opts, den_graph, supervisions,
                                                           prediction.detach(), operator_form=True)
        ctx.save_for_backward(neg_grad)
        ctx.per_sequence = out[:3 * N].view(3, N)
        return out[3 * N]

    @staticmethod
    def backward(ctx, grad_out):
        neg_grad, = ctx.saved_tensors
        return neg_grad, None, None, None


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
        # reduction='mean': the valid targets are counted by a one-workgroup launch in front, so the kernel writes the MEAN
        # loss's gradient as it is; backward applies the loss's incoming gradient in place by a launch that returns at
        # once when that gradient is exactly 1 (loss.backward()).  Rounds 1-6a scaled in a pass of its own over the
        # [rows, P] tensor (185 us of the 18.5 ms CE step for 2 x 470 MB; PK2_CE_SCALE_PASS=1 keeps it).
        ctx.prescaled = os.environ.get("PK2_CE_SCALE_PASS", "0") != "1"
        if reduction == "mean" and ctx.prescaled:
            _lib.check(L.pk2_softmax_ce_fwd_bwd_mean(_lib.ptr(x2), P, _lib.ptr(tg), int(ignore_index), rows, P,
                                                     _lib.ptr(acc), _lib.ptr(cnt), _lib.ptr(grad), P, _lib.stream_ptr(x.device)))
        else:
            _lib.check(L.pk2_softmax_ce_fwd_bwd(_lib.ptr(x2), P, _lib.ptr(tg), int(ignore_index), rows, P,
                                                _lib.ptr(acc), _lib.ptr(cnt), _lib.ptr(grad), P, None,
                                                _lib.stream_ptr(x.device)))
        if reduction == "mean":
            loss = acc[0] / cnt.to(torch.float32).clamp_min(1.0)[0]
        else:
            loss = acc[0].clone()
        ctx.save_for_backward(grad, cnt)
        ctx.mean = reduction == "mean"
        ctx.view = (layout, tuple(x.shape), P)
        ctx.applied = None          # the incoming gradient the saved buffer already carries (a second backward over a retained graph)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        grad, cnt = ctx.saved_tensors
        layout, shape, P = ctx.view
        go = grad_out.detach().to(device=grad.device, dtype=torch.float32).reshape(-1)[:1].contiguous()
        if ctx.prescaled:
            _lib.check(_lib.lib().pk2_scale_inplace_ratio(_lib.ptr(grad), grad.numel(), _lib.ptr(go),
                                                          _lib.ptr(ctx.applied) if ctx.applied is not None else None,
                                                          _lib.stream_ptr(grad.device)))
            ctx.applied = go
            out = grad
        else:
            out = torch.empty_like(grad)
            _lib.check(_lib.lib().pk2_scale_by_scalars(_lib.ptr(grad), _lib.ptr(out), grad.numel(), _lib.ptr(go),
                                                       _lib.ptr(cnt) if ctx.mean else None, _lib.stream_ptr(grad.device)))
        g = out.view(shape[1], shape[0], P).transpose(0, 1) if layout == "tm" else out.view(shape)
        return g, None, None, None


class CrossEntropyLoss(torch.nn.Module):
    """nn.CrossEntropyLoss(ignore_index=-100, reduction='mean'|'sum') as the reference uses it
    (bin/train_ce.py:134,189; bin/train_se.py:214,235), fused into one HIP kernel."""

    def __init__(self, ignore_index=-100, reduction="mean"):
        super().__init__()
        assert reduction in ("mean", "sum")
        self.ignore_index, self.reduction = ignore_index, reduction

    def forward(self, logits, targets):
        return _CrossEntropyFunction.apply(logits, targets, self.ignore_index, self.reduction)


class MMIFunction(Function):
    """Lattice MMI for one utterance with the reference's signature (ops/ops.py:41-75):
    ``MMIFunction.apply(loglikes[T, P], asr_decoder, trans_model, trans_ids)``.  Forward returns the lattice
    log-likelihood (as the reference does); backward returns -(numerator - denominator posteriors)
    regardless of grad_out.  asr_decoder = lattice.MappedLatticeFasterRecognizer."""

    @staticmethod
    def forward(ctx, loglikes, asr_decoder, trans_model, trans_ids):
        lat = asr_decoder.decode(loglikes.detach().contiguous())
        like, post = lat.mmi([trans_ids], 1.0, 0.2, True)     # lattice_scale(1.0, 0.2), drop_frames, cancel
        ctx.save_for_backward(post[0])
        return like[0].to(torch.float32)

    @staticmethod
    def backward(ctx, grad_out):
        post, = ctx.saved_tensors
        return -post, None, None, None


class sMBRFunction(Function):
    """sMBR / MPFE for one utterance (ops/ops.py:119-156):
    ``sMBRFunction.apply(loglikes, asr_decoder, trans_model, trans_ids, criterion, silence_phones)``.
    Forward returns the expected frame accuracy; backward its negated derivative."""

    @staticmethod
    def forward(ctx, loglikes, asr_decoder, trans_model, trans_ids, criterion, silence_phones):
        lat = asr_decoder.decode(loglikes.detach().contiguous())
        score, post = lat.mpe([trans_ids], criterion, silence_phones, True)
        ctx.save_for_backward(post[0])
        return score[0].to(torch.float32)

    @staticmethod
    def backward(ctx, grad_out):
        post, = ctx.saved_tensors
        return -post, None, None, None, None, None


class LatticeBatchFunction(Function):
    """All utterances of a minibatch in one decode + one forward-backward launch: replaces the reference's
    per-utterance Python loop (bin/train_se.py:237-249).  prediction: [N, Tmax, P] log-likelihoods (log prior
    already subtracted); returns the summed criterion."""

    @staticmethod
    def forward(ctx, prediction, lengths, asr_decoder, trans_model, trans_ids, criterion, silence_phones):
        lat = asr_decoder.decode_batch(prediction.detach(), lengths)
        if criterion == "mmi":
            val, post = lat.mmi(trans_ids, 1.0, 0.2, True)
        else:
            val, post = lat.mpe(trans_ids, criterion, silence_phones, True)
        ctx.save_for_backward(post)
        ctx.per_sequence = val
        ctx.lattice = lat
        return val.sum().to(torch.float32)

    @staticmethod
    def backward(ctx, grad_out):
        post, = ctx.saved_tensors
        return -post, None, None, None, None, None, None
