"""TransformerAM -- drop-in for reference models/transformer.py:69-94 running on libpk2hip.so.

Same constructor, same ``state_dict`` keys (``input_layer.*``, ``output_layer.*``, ``pos_encoder.pe``,
``transformer.layers.{i}.encoder_layer.{self_attn.in_proj_weight, self_attn.in_proj_bias,
self_attn.out_proj.*, linear1.*, linear2.*, norm1.*, norm2.*}``, ``transformer.layers.{i}.conv1d.*``,
``transformer.norm.*``) and the same default initialisation: the parameter containers are the torch
modules the reference instantiates, created in the reference's order (all encoder layers start as copies
of one initialised layer, like ``nn.TransformerEncoder``'s deep copies), but **none of their forward
methods is used**.  ``forward(x[T,B,D], src_mask, src_key_padding_mask) -> [T,B,P]``:

    Linear(80->C) -> n x [ post-norm encoder layer (MHA + ReLU FFN) -> Conv1d(C,C,k=3,pad=1) over time
    -> ReLU ] -> LayerNorm -> Linear(C->P);  positional encoding disabled (models/transformer.py:90).

Arithmetic: f32 MFMA GEMMs (`pk2_gemm_f32`, batched per (utterance, head) for QK^T and PV), masked
softmax, LayerNorm(+residual), ReLU, counter-based dropout -- all HIP kernels of libpk2hip.so.  With head size 64
(the reference's 512 / 8) the attention core is the fused kernel of csrc/attention.hip (scores never reach HBM, backward
by recomputation); other head sizes, or PK2_ATTN_FUSED=0, take the batched-GEMM form with the scores in HBM.
"""
import copy
import math
import os

import torch
import torch.nn as nn

from . import _lib
from .lstm import _gemm, _p


class PositionalEncoding(nn.Module):
    """Kept for checkpoint compatibility (buffer ``pe``); the reference never applies it."""

    def __init__(self, dim_model, dropout=0, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, dim_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, dim_model, 2).float() * (-math.log(10000.0) / dim_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe.unsqueeze(0).transpose(0, 1))


class _LayerParams(nn.Module):
    def __init__(self, dim_model, nheads, dim_feedforward, dropout, kernel_size, stride):
        super().__init__()
        self.encoder_layer = nn.TransformerEncoderLayer(dim_model, nheads, dim_feedforward, dropout)
        self.conv1d = nn.Conv1d(dim_model, dim_model, kernel_size, stride=stride, padding=1)


class _EncoderParams(nn.Module):
    def __init__(self, layer, nlayers, norm):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(nlayers)])
        self.norm = norm


def _bgemm(ta, tb, M, N, K, alpha, A, lda, sA0, sA1, B, ldb, sB0, sB1, beta, C, ldc, sC0, sC1, n0, n1):
    _lib.check(_lib.lib().pk2_gemm_f32_batched(int(ta), int(tb), M, N, K, alpha, A, lda, sA0, sA1, B, ldb, sB0, sB1,
                                               beta, C, ldc, sC0, sC1, n0, n1, _lib.stream_ptr()))


def _gemm_act(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias=None, beta=0.0, act=0, gate=None, ldg=0):
    """pk2_gemm_f32_act: the product with the ReLU (act 1) or the ReLU's backward mask (act 2: kept where gate > 0) in its
    epilogue instead of a row pass of its own behind it (PK2_TR_FUSE_RELU=0: the separate launches)."""
    _lib.check(_lib.lib().pk2_gemm_f32_act(int(ta), int(tb), M, N, K, 1.0, A, lda, B, ldb, beta, C, ldc, bias, act, gate, ldg,
                                           _lib.stream_ptr()))


def _gemm_seg(ta, tb, M, N, K, nseg, A, lda, segA, B, ldb, segB, C, ldc, bias=None, beta=0.0, act=0, gate=None, ldg=0):
    """pk2_gemm_f32_seg: sum over nseg products (A + j segA)(B + j segB) in one launch, one set of accumulators."""
    _lib.check(_lib.lib().pk2_gemm_f32_seg(int(ta), int(tb), M, N, K, nseg, 1.0, A, lda, segA, B, ldb, segB, beta, C, ldc, bias,
                                           act, gate, ldg, _lib.stream_ptr()))


def _fuse_conv():
    """Conv1d(k = 3) as ONE product over three row-shifted views of a zero-padded buffer (PK2_TR_FUSE_CONV=0: three products
    accumulated with beta = 1 and a ReLU pass, the form of rounds 2-5)."""
    return os.environ.get("PK2_TR_FUSE_CONV", "1") != "0"


def _fuse_relu():
    return os.environ.get("PK2_TR_FUSE_RELU", "1") != "0"


def _colsum(A, lda, M, N, out, beta=0.0):
    _lib.check(_lib.lib().pk2_colsum_f32(A, lda, M, N, beta, out, _lib.stream_ptr()))


def _dropout(x, p, seed, out=None):
    out = torch.empty_like(x) if out is None else out
    _lib.check(_lib.lib().pk2_dropout_f32(_p(x), _p(out), x.numel(), float(p), seed, _lib.stream_ptr()))
    return out


def _seed():
    return int(torch.empty((), dtype=torch.int64).random_().item()) & 0x7FFFFFFFFFFFFFFF


def _attention_bwd_unfused(L, sp, new, s, qkv, dcx, dqkv, T, B, C, H, d, drop):
    """Batched-GEMM form: dP = dctx V^T ; dV = Pd^T dctx ; dS = softmax'(P, dP) ; dQ = a dS K ; dK = a dS^T Q."""
    dP = new(B * H, T, T)
    _bgemm(0, 1, T, T, d, 1.0, _p(dcx), B * C, C, d, _p(qkv, 2 * C), B * 3 * C, 3 * C, d, 0.0, _p(dP), T,
           H * T * T, T * T, B, H)
    _bgemm(1, 0, T, d, T, 1.0, _p(s["Pd"]), T, H * T * T, T * T, _p(dcx), B * C, C, d, 0.0,
           _p(dqkv, 2 * C), B * 3 * C, 3 * C, d, B, H)
    if drop > 0:
        _dropout(dP, drop, s["seed_attn"], dP)
    _lib.check(L.pk2_softmax_bwd(_p(s["P"]), _p(dP), B * H, T, sp))
    sc = 1.0 / math.sqrt(d)
    _bgemm(0, 0, T, d, T, sc, _p(dP), T, H * T * T, T * T, _p(qkv, C), B * 3 * C, 3 * C, d, 0.0,
           _p(dqkv), B * 3 * C, 3 * C, d, B, H)
    _bgemm(1, 0, T, d, T, sc, _p(dP), T, H * T * T, T * T, _p(qkv), B * 3 * C, 3 * C, d, 0.0,
           _p(dqkv, C), B * 3 * C, 3 * C, d, B, H)


class _TransformerFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model, src_mask, key_pad, *params):
        m = model
        L = _lib.lib()
        sp = _lib.stream_ptr()
        T, B, Din = x.shape
        C, H, F, P = m.dim_model, m.nheads, m.dim_feedforward, m.output_size
        d = C // H
        R = T * B
        dev = x.device
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
        drop = m.dropout if m.training else 0.0
        fused = d == 64 and os.environ.get("PK2_ATTN_FUSED", "1") != "0"
        ctx.fused, ctx.masks = fused, (src_mask, key_pad)
        fuse_relu, fuse_conv = _fuse_relu(), _fuse_conv()

        def new_padded(rows, cols):
            """[B + rows + B, cols] with zero rows in front and behind (the convolution's padding in time: row r -+ B of
            the time-major matrix is frame t -+ 1 of the same utterance); returns the buffer and its middle."""
            buf = torch.zeros(rows + 2 * B, cols, device=dev, dtype=torch.float32)
            return buf, buf[B:B + rows]
        x = x.contiguous()
        h = new(R, C)
        _gemm(0, 1, R, C, Din, _p(x), Din, _p(m.input_layer.weight), Din, _p(h), C, bias=_p(m.input_layer.bias))
        saved = []
        # The convolution taps of ALL layers in product layout [L][3][Cout][Cin] with one strided copy: the layers' weights
        # sit at equal distances in the flat parameter buffer (same parameters in the same order per layer)
        Wp_all = None
        nl = len(m.transformer.layers)
        if nl > 1 and m._flat is not None:
            offs = [m._layout["transformer.layers.%d.conv1d.weight" % i][0] for i in range(nl)]
            step = offs[1] - offs[0]
            if step > 0 and all(offs[i] == offs[0] + i * step for i in range(nl)):
                Wp_all = torch.as_strided(m._flat, (nl, C, C, 3), (step, 3 * C, 3, 1), offs[0]).permute(0, 3, 1, 2).contiguous()
        for li_, lp in enumerate(m.transformer.layers):
            e = lp.encoder_layer
            a = e.self_attn
            s = dict(h_in=h)
            qkv = new(R, 3 * C)
            _gemm(0, 1, R, 3 * C, C, _p(h), C, _p(a.in_proj_weight), C, _p(qkv), 3 * C, bias=_p(a.in_proj_bias))
            s["seed_attn"] = _seed() if drop > 0 else None
            cx = new(R, C)
            Pm = Pd = lse = None
            if fused:
                lse = new(B * H, T)
                _lib.check(L.pk2_attention_fwd(_p(qkv), T, B, H, d, 1.0 / math.sqrt(d),
                                               _p(src_mask) if src_mask is not None else None,
                                               _p(key_pad) if key_pad is not None else None, float(drop),
                                               s["seed_attn"] or 0, _p(cx), _p(lse), sp))
            else:
                Pm = new(B * H, T, T)
                _bgemm(0, 1, T, T, d, 1.0 / math.sqrt(d), _p(qkv), B * 3 * C, 3 * C, d, _p(qkv, C), B * 3 * C, 3 * C, d,
                       0.0, _p(Pm), T, H * T * T, T * T, B, H)
                _lib.check(L.pk2_softmax_mask_fwd(_p(Pm), _p(src_mask) if src_mask is not None else None,
                                                  _p(key_pad) if key_pad is not None else None, B, H, T, sp))
                Pd = _dropout(Pm, drop, s["seed_attn"]) if drop > 0 else Pm
                _bgemm(0, 0, T, d, T, 1.0, _p(Pd), T, H * T * T, T * T, _p(qkv, 2 * C), B * 3 * C, 3 * C, d, 0.0,
                       _p(cx), B * C, C, d, B, H)
            ao = new(R, C)
            _gemm(0, 1, R, C, C, _p(cx), C, _p(a.out_proj.weight), C, _p(ao), C, bias=_p(a.out_proj.bias))
            s["seed1"] = _seed() if drop > 0 else None
            if drop > 0:
                _dropout(ao, drop, s["seed1"], ao)
            s1, x1, mu1, rs1 = new(R, C), new(R, C), new(R), new(R)
            _lib.check(L.pk2_layernorm_fwd(_p(ao), _p(h), _p(e.norm1.weight), _p(e.norm1.bias), R, C, e.norm1.eps,
                                           _p(s1), _p(x1), _p(mu1), _p(rs1), sp))
            f1 = new(R, F)
            if fuse_relu:
                _gemm_act(0, 1, R, F, C, _p(x1), C, _p(e.linear1.weight), C, _p(f1), F, bias=_p(e.linear1.bias), act=1)
            else:
                _gemm(0, 1, R, F, C, _p(x1), C, _p(e.linear1.weight), C, _p(f1), F, bias=_p(e.linear1.bias))
                _lib.check(L.pk2_relu_fwd(_p(f1), f1.numel(), sp))
            s["seedf"] = _seed() if drop > 0 else None
            f1d = _dropout(f1, drop, s["seedf"]) if drop > 0 else f1
            f2 = new(R, C)
            _gemm(0, 1, R, C, F, _p(f1d), F, _p(e.linear2.weight), F, _p(f2), C, bias=_p(e.linear2.bias))
            s["seed2"] = _seed() if drop > 0 else None
            if drop > 0:
                _dropout(f2, drop, s["seed2"], f2)
            s2, mu2, rs2 = new(R, C), new(R), new(R)
            x2p, x2 = new_padded(R, C) if fuse_conv else (None, new(R, C))
            _lib.check(L.pk2_layernorm_fwd(_p(f2), _p(x1), _p(e.norm2.weight), _p(e.norm2.bias), R, C, e.norm2.eps,
                                           _p(s2), _p(x2), _p(mu2), _p(rs2), sp))
            # Conv1d(k=3, pad=1) over time = three products over row-shifted slices (time-major: t-1 <-> row - B)
            Wp = Wp_all[li_] if Wp_all is not None else lp.conv1d.weight.detach().permute(2, 0, 1).contiguous()   # [3][Cout][Cin]
            y = new(R, C)
            if fuse_conv:      # tap j reads rows r + (j - 1) B: views j B rows into the padded buffer; bias and ReLU in the epilogue
                _gemm_seg(0, 1, R, C, C, 3, _p(x2p), C, B * C, _p(Wp), C, C * C, _p(y), C, bias=_p(lp.conv1d.bias), act=1)
            else:
                _gemm(0, 1, R, C, C, _p(x2), C, _p(Wp, C * C), C, _p(y), C, bias=_p(lp.conv1d.bias))
                if T > 1:
                    _gemm(0, 1, R - B, C, C, _p(x2), C, _p(Wp, 0), C, _p(y, B * C), C, beta=1.0)
                    _gemm(0, 1, R - B, C, C, _p(x2, B * C), C, _p(Wp, 2 * C * C), C, _p(y), C, beta=1.0)
                _lib.check(L.pk2_relu_fwd(_p(y), y.numel(), sp))
            s.update(qkv=qkv, P=Pm, Pd=Pd, lse=lse, cx=cx, s1=s1, x1=x1, mu1=mu1, rs1=rs1, f1=f1, f1d=f1d, s2=s2, x2=x2,
                     mu2=mu2, rs2=rs2, Wp=Wp, y=y)
            saved.append(s)
            h = y
        nf = m.transformer.norm
        hn, muf, rsf = new(R, C), new(R), new(R)
        _lib.check(L.pk2_layernorm_fwd(_p(h), None, _p(nf.weight), _p(nf.bias), R, C, nf.eps, None, _p(hn), _p(muf),
                                       _p(rsf), sp))
        logits = new(T, B, P)
        _gemm(0, 1, R, P, C, _p(hn), C, _p(m.output_layer.weight), C, _p(logits), P, bias=_p(m.output_layer.bias))
        ctx.model, ctx.saved, ctx.x, ctx.final = m, saved, x, (h, hn, muf, rsf)
        ctx.shape, ctx.drop = (T, B, Din), drop
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        m = ctx.model
        L = _lib.lib()
        sp = _lib.stream_ptr()
        T, B, Din = ctx.shape
        C, H, F, P = m.dim_model, m.nheads, m.dim_feedforward, m.output_size
        d = C // H
        R = T * B
        drop = ctx.drop
        dev = dlogits.device
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
        dlogits = dlogits.contiguous()
        g = m._grad_views()
        hL, hn, muf, rsf = ctx.final
        # ONE fill of the flat gradient buffer, then every parameter gradient is ADDED to it (beta = 1): a product whose K
        # = frames is cut into slices needs no launch that prepares C, a column sum none that clears its output, the
        # LayerNorm gradients none of their own (round 3 counted 62 + 70 + 62 such launches of ~5 us per step)
        m.flat_parameters()[1].zero_()
        # Round 5: the parameter gradients (a weight-gradient product + a column sum per linear layer, three products for
        # the convolution) hang off the chain  d out -> d in -> next layer; nothing but the optimiser waits for them, and a
        # product of this model fills a third of the chip (296 tiles of 64 x 64 in 768 slots).  They go to a side stream --
        # there is no persistent kernel in this model whose co-residency they could hurt (the BLSTM's recurrences lose more
        # than the overlap returns: lstm.py) -- fenced by events; backward ends with the compute stream waiting for them.
        # PK2_TR_SIDE_STREAM=0 keeps everything on one stream.
        main = torch.cuda.current_stream(dev)
        side = m._side_stream(dev) if os.environ.get("PK2_TR_SIDE_STREAM", "1") != "0" else None

        def on_side(fn, *tensors):
            """Runs fn on the side stream behind what the compute stream has enqueued so far; returns an event the compute
            stream must wait for before it OVERWRITES one of `tensors` (None on one stream)."""
            if side is None:
                fn()
                return None
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                fn()
                done = torch.cuda.Event()
                done.record(side)
            for t_ in tensors:
                t_.record_stream(side)
            return done

        def before_overwrite(done):
            if done is not None:
                main.wait_event(done)

        def lin_grads(dout, inp, in_dim, out_dim, wname, bname):
            def work():
                st = _lib.stream_ptr()
                # weight and bias gradient in one launch (the column sums ride in the product's operand loader)
                _lib.check(L.pk2_gemm_f32_tn_colsum(out_dim, in_dim, R, 1.0, _p(dout), out_dim, _p(inp), in_dim, 1.0, _p(g[wname]), in_dim,
                                                    _p(g[bname]), st))
            return on_side(work, dout, inp)

        lin_grads(dlogits, hn, C, P, "output_layer.weight", "output_layer.bias")
        dhn = new(R, C)
        _gemm(0, 0, R, C, P, _p(dlogits), P, _p(m.output_layer.weight), C, _p(dhn), C)
        nf = m.transformer.norm
        fuse_conv = _fuse_conv()

        def new_dh():
            """A gradient that becomes a convolution's output gradient: B zero rows in front and behind (see forward)."""
            if not fuse_conv:
                return new(R, C)
            return torch.zeros(R + 2 * B, C, device=dev, dtype=torch.float32)[B:B + R]
        dh = new_dh()
        _lib.check(L.pk2_layernorm_bwd(_p(dhn), _p(hL), _p(muf), _p(rsf), _p(nf.weight), R, C, _p(dh),
                                       _p(g["transformer.norm.weight"]), _p(g["transformer.norm.bias"]), sp))
        fuse_relu, dh_gated = _fuse_relu(), False
        for li in range(len(m.transformer.layers) - 1, -1, -1):
            lp = m.transformer.layers[li]
            e = lp.encoder_layer
            a = e.self_attn
            s = ctx.saved[li]
            pre = "transformer.layers.%d." % li
            # ReLU + Conv1d
            if not dh_gated:       # (else the product that finished dh applied this layer's ReLU mask in its epilogue)
                _lib.check(L.pk2_relu_bwd(_p(s["y"]), _p(dh), dh.numel(), sp))     # dh := dc
            dc = dh
            x2 = s["x2"]

            def conv_grads(dc=dc, x2=x2, pre=pre):
                st = _lib.stream_ptr()

                def tn(Mo, No, Ko, A, Bm, Cm):
                    _lib.check(L.pk2_gemm_f32(1, 0, Mo, No, Ko, 1.0, A, C, Bm, C, 1.0, Cm, C, None, st))
                dWp = torch.zeros(3, C, C, device=dev, dtype=torch.float32)
                # centre tap over all rows: its operand's column sums are the bias gradient (same launch)
                _lib.check(L.pk2_gemm_f32_tn_colsum(C, C, R, 1.0, _p(dc), C, _p(x2), C, 1.0, _p(dWp, C * C), C,
                                                    _p(g[pre + "conv1d.bias"]), st))
                if T > 1:
                    tn(C, C, R - B, _p(dc, B * C), _p(x2), _p(dWp, 0))
                    tn(C, C, R - B, _p(dc), _p(x2, B * C), _p(dWp, 2 * C * C))
                g[pre + "conv1d.weight"].copy_(dWp.permute(1, 2, 0))
            on_side(conv_grads, dc, x2)
            dx2 = new(R, C)
            if fuse_conv:      # tap j's transpose reads rows r + (1 - j) B of dc: one product over three views of its padded buffer
                _gemm_seg(0, 0, R, C, C, 3, _p(dc, B * C), C, -B * C, _p(s["Wp"]), C, C * C, _p(dx2), C)
            else:
                _gemm(0, 0, R, C, C, _p(dc), C, _p(s["Wp"], C * C), C, _p(dx2), C)
                if T > 1:
                    _gemm(0, 0, R - B, C, C, _p(dc, B * C), C, _p(s["Wp"], 0), C, _p(dx2), C, beta=1.0)
                    _gemm(0, 0, R - B, C, C, _p(dc), C, _p(s["Wp"], 2 * C * C), C, _p(dx2, B * C), C, beta=1.0)
            # LayerNorm 2 (+ residual)
            ds2 = new(R, C)
            _lib.check(L.pk2_layernorm_bwd(_p(dx2), _p(s["s2"]), _p(s["mu2"]), _p(s["rs2"]), _p(e.norm2.weight), R, C,
                                           _p(ds2), _p(g[pre + "encoder_layer.norm2.weight"]),
                                           _p(g[pre + "encoder_layer.norm2.bias"]), sp))
            df2 = _dropout(ds2, drop, s["seed2"]) if drop > 0 else ds2
            # FFN
            read_ds2 = lin_grads(df2, s["f1d"], F, C, pre + "encoder_layer.linear2.weight", pre + "encoder_layer.linear2.bias")
            df1 = new(R, F)
            if fuse_relu:         # (the ReLU mask and the dropout mask are both elementwise factors: their order is free)
                _gemm_act(0, 0, R, F, C, _p(df2), C, _p(e.linear2.weight), F, _p(df1), F, act=2, gate=_p(s["f1"]), ldg=F)
                if drop > 0:
                    _dropout(df1, drop, s["seedf"], df1)
            else:
                _gemm(0, 0, R, F, C, _p(df2), C, _p(e.linear2.weight), F, _p(df1), F)
                if drop > 0:
                    _dropout(df1, drop, s["seedf"], df1)
                _lib.check(L.pk2_relu_bwd(_p(s["f1"]), _p(df1), df1.numel(), sp))
            lin_grads(df1, s["x1"], C, F, pre + "encoder_layer.linear1.weight", pre + "encoder_layer.linear1.bias")
            dx1 = ds2 if drop == 0 else ds2    # residual branch: d x1 = d s2 (+ FFN path below)
            if df2 is ds2:
                before_overwrite(read_ds2)     # (the side stream reads d s2 as linear2's output gradient: the add below writes it)
            _gemm(0, 0, R, C, F, _p(df1), F, _p(e.linear1.weight), C, _p(dx1), C, beta=1.0)
            # LayerNorm 1 (+ residual)
            ds1 = new_dh()         # (ends up as dh: the output gradient of the convolution of the layer below)
            _lib.check(L.pk2_layernorm_bwd(_p(dx1), _p(s["s1"]), _p(s["mu1"]), _p(s["rs1"]), _p(e.norm1.weight), R, C,
                                           _p(ds1), _p(g[pre + "encoder_layer.norm1.weight"]),
                                           _p(g[pre + "encoder_layer.norm1.bias"]), sp))
            dao = _dropout(ds1, drop, s["seed1"]) if drop > 0 else ds1
            read_ds1 = lin_grads(dao, s["cx"], C, C, pre + "encoder_layer.self_attn.out_proj.weight",
                                 pre + "encoder_layer.self_attn.out_proj.bias")
            dcx = new(R, C)
            _gemm(0, 0, R, C, C, _p(dao), C, _p(a.out_proj.weight), C, _p(dcx), C)
            # attention: dP = dctx V^T ; dV = Pd^T dctx ; dS = softmax'(P, dP) ; dQ = a dS K ; dK = a dS^T Q
            qkv = s["qkv"]
            dqkv = new(R, 3 * C)
            if ctx.fused:
                src_mask, key_pad = ctx.masks
                _lib.check(L.pk2_attention_bwd(_p(qkv), _p(s["cx"]), _p(dcx), _p(s["lse"]), T, B, H, d, 1.0 / math.sqrt(d),
                                               _p(src_mask) if src_mask is not None else None,
                                               _p(key_pad) if key_pad is not None else None, float(drop),
                                               s["seed_attn"] or 0, _p(dqkv), _p(new(B * H, T)), sp))
            else:
                _attention_bwd_unfused(L, sp, new, s, qkv, dcx, dqkv, T, B, C, H, d, drop)
            lin_grads(dqkv, s["h_in"], C, 3 * C, pre + "encoder_layer.self_attn.in_proj_weight",
                      pre + "encoder_layer.self_attn.in_proj_bias")
            dh = ds1 if drop == 0 else ds1     # residual branch of the attention block
            if dao is ds1:
                before_overwrite(read_ds1)
            if fuse_relu and li > 0:     # dh is complete with this product: the layer below starts with ReLU'(its y) on it
                _gemm_act(0, 0, R, C, 3 * C, _p(dqkv), 3 * C, _p(a.in_proj_weight), C, _p(dh), C, beta=1.0, act=2,
                          gate=_p(ctx.saved[li - 1]["y"]), ldg=C)
                dh_gated = True
            else:
                _gemm(0, 0, R, C, 3 * C, _p(dqkv), 3 * C, _p(a.in_proj_weight), C, _p(dh), C, beta=1.0)
                dh_gated = False
            ctx.saved[li] = None
        lin_grads(dh, ctx.x.view(R, Din), Din, C, "input_layer.weight", "input_layer.bias")
        dx = None
        if ctx.x.requires_grad:
            dx = new(T, B, Din)
            _gemm(0, 0, R, Din, C, _p(dh), C, _p(m.input_layer.weight), Din, _p(dx), Din)
        if side is not None:
            main.wait_stream(side)
        params = dict(m.named_parameters())
        for name, v in g.items():
            p_ = params[name]
            if p_.grad is None or p_.grad.data_ptr() != v.data_ptr():
                p_.grad = v
        return (dx, None, None, None) + (None,) * len(m._param_names)


def _mask_to_device(mask, device, dtype):
    """A mask built on the host goes over pinned and non-blocking: a pageable copy is hipMemcpyAsync + a stream
    synchronisation, i.e. the host would wait at the head of every forward until the device has finished the previous
    step and then start enqueueing ~470 launches against an idle device (measured on the 12-layer LF-MMI step:
    tools/host_time_tr.py)."""
    if mask.is_cuda:
        return mask.to(device=device, dtype=dtype).contiguous()
    return mask.to(dtype).contiguous().pin_memory().to(device, non_blocking=True)


class TransformerAM(nn.Module):
    def __init__(self, dim_feat, dim_model, nheads, dim_feedforward, nlayers, dropout, output_size, kernel_size=3,
                 stride=1):
        super().__init__()
        assert kernel_size == 3 and stride == 1, "the HIP path implements the reference default Conv1d(k=3, stride=1, pad=1)"
        assert dim_model % nheads == 0
        self.dim_feat, self.dim_model, self.nheads = dim_feat, dim_model, nheads
        self.dim_feedforward, self.nlayers, self.dropout, self.output_size = dim_feedforward, nlayers, dropout, output_size
        # the reference's construction order (models/transformer.py:80-86) -> identical default initialisation
        self.pos_encoder = PositionalEncoding(dim_model, dropout)
        self.input_layer = nn.Linear(dim_feat, dim_model)
        self.output_layer = nn.Linear(dim_model, output_size)
        encoder_norm = nn.LayerNorm(dim_model)
        encoder_layer = _LayerParams(dim_model, nheads, dim_feedforward, dropout, kernel_size, stride)
        self.transformer = _EncoderParams(encoder_layer, nlayers, encoder_norm)
        self._param_names = [n for n, _ in self.named_parameters()]
        self._flat = self._gflat = self._layout = self._plist = None

    # ---- flat parameter / gradient buffers (fused optimiser, all-reduce) -------------------------------
    def _ensure_flat(self):
        # steady state: the cached parameter objects are still the module's and still views of the flat buffer (a walk
        # over named_parameters() costs 0.7 ms of host time, and a step asks three times)
        cached = self._plist
        if cached is not None and self._flat is not None and self.input_layer.weight is cached[0][0] \
                and self._flat.device == cached[0][0].device:
            base = self._flat.data_ptr()
            if all(q.data_ptr() == base + 4 * o for q, o in cached):
                return
        params = dict(self.named_parameters())
        first = params[self._param_names[0]]
        if self._flat is not None and self._flat.device == first.device and all(
                params[n].data_ptr() == self._flat.data_ptr() + 4 * o for n, (o, _) in self._layout.items()):
            self._plist = [(params[n], self._layout[n][0]) for n in self._param_names]
            return
        layout, off = {}, 0
        for n in self._param_names:
            layout[n] = (off, params[n].numel())
            off += (params[n].numel() + 63) // 64 * 64
        flat = torch.zeros(off, dtype=torch.float32, device=first.device)
        for n, (o, c) in layout.items():
            flat[o:o + c].copy_(params[n].data.reshape(-1))
            params[n].data = flat[o:o + c].view(params[n].shape)
        self._flat, self._layout, self._gflat = flat, layout, None
        self._plist = [(params[n], layout[n][0]) for n in self._param_names]

    def _side_stream(self, dev):
        st = getattr(self, "_side", None)
        if st is None or st.device != dev:
            st = self._side = torch.cuda.Stream(device=dev)
        return st

    def flat_parameters(self):
        self._ensure_flat()
        if self._gflat is None or self._gflat.device != self._flat.device:
            self._gflat = torch.zeros_like(self._flat)
        return self._flat, self._gflat

    def _grad_views(self):
        _, gflat = self.flat_parameters()
        params = dict(self.named_parameters())
        return {n: gflat[o:o + c].view(params[n].shape) for n, (o, c) in self._layout.items()}

    def forward(self, data, src_mask=None, src_key_padding_mask=None):
        """data [T,B,D]; src_mask [T,T] additive float (-inf = blocked) or None; src_key_padding_mask [B,T] bool,
        True = padding (reference bin/train_transformer_se.py:245-257)."""
        _lib.require_gpu()
        assert data.is_cuda and data.dtype == torch.float32 and data.dim() == 3
        self._ensure_flat()
        kp = None
        if src_key_padding_mask is not None:
            kp = _mask_to_device(src_key_padding_mask, data.device, torch.uint8)
        sm = None
        if src_mask is not None:
            sm = _mask_to_device(src_mask, data.device, torch.float32)
        params = [q for q, _ in self._plist]     # named_parameters() order (_ensure_flat)
        return _TransformerFunction.apply(data, self, sm, kp, *params)


def padded_forward(model, x, frames, look_ahead=-1):
    """The masks of reference bin/train_transformer_se.py:242-257 (= train_transformer_ce.py:188-201) around one
    forward: x [T, B, D] time-major, zero-padded; frames[b] valid frames of utterance b.  key_padding_mask is True
    on padding; with look_ahead > -1 frame t may attend to frames <= t + look_ahead.  Returns [B, T, P] (a
    transposed view of the model's [T, B, P] output)."""
    T, B = x.shape[0], x.shape[1]
    kpm = torch.ones(B, T, dtype=torch.bool)
    for b, n in enumerate(frames):
        kpm[b, :int(n)] = False
    src_mask = None
    if look_ahead > -1:
        keep = torch.tril(torch.ones(T, T), diagonal=look_ahead)
        src_mask = torch.zeros(T, T).masked_fill(keep == 0, float("-inf"))
    return model(x, src_mask, kpm).transpose(0, 1)
