"""LSTMAM -- drop-in for reference models/lstm.py:33-61 running on libpk2hip.so.

Same constructor, same ``state_dict`` keys (``lstm.weight_ih_l{k}[_reverse]``,
``lstm.weight_hh_l{k}[_reverse]``, ``lstm.bias_ih_l{k}[_reverse]``,
``lstm.bias_hh_l{k}[_reverse]``, ``output_layer.{weight,bias}``), same default
initialisation (parameters are created and initialised in the reference's order,
so ``torch.manual_seed(s)`` gives bit-identical weights), same semantics:
``forward(x[B,T,D]) -> logits[B,T,P]`` = Linear(LSTM(x)) with h0 = c0 = 0,
gate order i,f,g,o, every sequence run over all T frames (no packing).  The
reference's own forward has a typo (``self_output_layer``, models/lstm.py:59);
the intended computation is implemented.

All arithmetic is in libpk2hip.so: f32 MFMA GEMMs for the input / output
projections and every weight gradient, per-step recurrent MFMA kernels with the
gate math fused.  Parameters live in one flat buffer (and gradients in another)
so the optimiser and the gradient all-reduce each see a single contiguous
array; within a layer the two directions' matrices are adjacent, so both
directions share one GEMM.
"""
import math
import ctypes
import os

import torch
import torch.nn as nn

from . import _lib


def _gemm(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias=None, alpha=1.0, beta=0.0):
    _lib.check(_lib.lib().pk2_gemm_f32(int(ta), int(tb), M, N, K, alpha, A, lda, B, ldb, beta, C, ldc,
                                       bias, _lib.stream_ptr()))


def _p(t, off_floats=0):
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + 4 * off_floats)


class _LSTMParams(nn.Module):
    """Parameter container with nn.LSTM's names, creation order and initialisation."""

    def __init__(self, input_size, hidden_size, num_layers, bidirectional):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.num_layers, self.bidirectional = num_layers, bidirectional
        D = 2 if bidirectional else 1
        H = hidden_size
        for layer in range(num_layers):
            for d in range(D):
                in_size = input_size if layer == 0 else H * D
                sfx = "_reverse" if d == 1 else ""
                self.register_parameter("weight_ih_l%d%s" % (layer, sfx), nn.Parameter(torch.empty(4 * H, in_size)))
                self.register_parameter("weight_hh_l%d%s" % (layer, sfx), nn.Parameter(torch.empty(4 * H, H)))
                self.register_parameter("bias_ih_l%d%s" % (layer, sfx), nn.Parameter(torch.empty(4 * H)))
                self.register_parameter("bias_hh_l%d%s" % (layer, sfx), nn.Parameter(torch.empty(4 * H)))
        stdv = 1.0 / math.sqrt(H) if H > 0 else 0
        for w in self.parameters():  # nn.LSTM.reset_parameters
            nn.init.uniform_(w, -stdv, stdv)


class _LinearParams(nn.Module):
    """nn.Linear's parameters and initialisation."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features) if in_features > 0 else 0
        nn.init.uniform_(self.bias, -bound, bound)


class _LstmAmFunction(torch.autograd.Function):
    """x_tm [T,B,Din] (time-major) -> logits_tm [T,B,P]."""

    @staticmethod
    def forward(ctx, x_tm, model, *params):
        m = model
        T, B, Din = x_tm.shape
        H, D, Lr = m.hidden_size, m.num_dirs, m.num_layers
        dev = x_tm.device
        L = _lib.lib()
        sp = _lib.stream_ptr()
        rows = T * B
        saved = []
        seeds = []
        inp = x_tm.contiguous()
        for l in range(Lr):
            w_ih, w_hh, b_ih, b_hh = m._layer_views(l)
            in_size = inp.shape[-1]
            gx = torch.empty(T, B, D * 4 * H, device=dev, dtype=torch.float32)
            _gemm(0, 1, rows, D * 4 * H, in_size, _p(inp), in_size, _p(w_ih), in_size, _p(gx), D * 4 * H,
                  bias=_p(b_ih))
            y = torch.empty(T, B, D * H, device=dev, dtype=torch.float32)
            gates = torch.empty(D, T, B, 4 * H, device=dev, dtype=torch.float32)
            cells = torch.empty(D, T, B, H, device=dev, dtype=torch.float32)
            nws = L.pk2_lstm_fwd_workspace_floats(B, H, D)
            ws = torch.empty(nws, device=dev, dtype=torch.float32) if nws else None
            _lib.check(L.pk2_lstm_layer_fwd(_p(gx), _p(w_hh), _p(b_hh), B, T, H, D, _p(y), _p(gates), _p(cells),
                                            _p(ws) if ws is not None else None, sp))
            saved.append((inp, y, gates, cells))
            inp = y
            if m.dropout > 0 and m.training and l + 1 < Lr:
                # nn.LSTM semantics: dropout on the outputs of every layer but the last; the recurrence
                # itself keeps reading the undropped h (y).  The seed comes from torch's CPU generator.
                seed = int(torch.empty((), dtype=torch.int64).random_().item()) & 0x7FFFFFFFFFFFFFFF
                seeds.append(seed)
                inp = torch.empty_like(y)
                _lib.check(L.pk2_dropout_f32(_p(y), _p(inp), y.numel(), float(m.dropout), seed, sp))
            else:
                seeds.append(None)
        P = m.output_size
        logits = torch.empty(T, B, P, device=dev, dtype=torch.float32)
        _gemm(0, 1, rows, P, D * H, _p(inp), D * H, _p(m.output_layer.weight), D * H, _p(logits), P,
              bias=_p(m.output_layer.bias))
        ctx.model = m
        ctx.saved = saved
        ctx.seeds = seeds
        ctx.shape = (T, B, Din)
        ctx.need_dx = x_tm.requires_grad
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        m = ctx.model
        T, B, Din = ctx.shape
        H, D, Lr, P = m.hidden_size, m.num_dirs, m.num_layers, m.output_size
        L = _lib.lib()
        sp = _lib.stream_ptr()
        rows = T * B
        dev = dlogits.device
        dlogits = dlogits.contiguous()
        gflat = m._grad_flat()
        # ONE memset of the flat gradient buffer, then every gradient product accumulates (beta = 1): a split-K GEMM no longer
        # needs its own zeroing launch (gemm_prescale_kernel), a column sum no scale_vec launch, and the bias gradients can
        # be added from inside the backward recurrence (pk2_lstm_layer_bwd_bias)
        gflat.zero_()
        views = m._grad_views(gflat)
        y_last = ctx.saved[-1][1]
        # The serial chain is  dlogits -> dy -> recurrence(l=L-1) -> dy -> recurrence(l=L-2) ...; the
        # weight / bias gradients hang off it and could overlap it on a side stream (PK2_SIDE_STREAM=1).
        # Measured on MI355X / ROCm 7.2: a second active stream slows the graph-replayed step chains of
        # the main stream by ~25 % (52.6 vs 48.5 ms per step), more than the overlap returns, so the
        # default keeps everything on the caller's stream.
        main = torch.cuda.current_stream(dev)
        side = m._side_stream(dev) if os.environ.get("PK2_SIDE_STREAM") == "1" else main

        def on_side(fn, *tensors):
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                fn()
            for t_ in tensors:
                t_.record_stream(side)

        # output layer: dy = dlogits W (critical path), dW = dlogits^T y, db = colsum(dlogits)
        gw, gb = views["output_layer.weight"], views["output_layer.bias"]
        dy = torch.empty(T, B, D * H, device=dev, dtype=torch.float32)
        _gemm(0, 0, rows, D * H, P, _p(dlogits), P, _p(m.output_layer.weight), D * H, _p(dy), D * H)

        def out_grads():
            # weight and bias gradient in one launch (round 6: the column sums ride in the product's operand loader)
            _lib.check(L.pk2_gemm_f32_tn_colsum(P, D * H, rows, 1.0, _p(dlogits), P, _p(y_last), D * H, 1.0, _p(gw), D * H, _p(gb),
                                                _lib.stream_ptr()))
            m._bucket_ready("output_layer")
        on_side(out_grads, dlogits, y_last)
        scratch = torch.empty(L.pk2_lstm_bwd_scratch_floats(B, H, D), device=dev, dtype=torch.float32)
        dx = None
        for l in range(Lr - 1, -1, -1):
            inp, y, gates, cells = ctx.saved[l]
            in_size = inp.shape[-1]
            w_ih, w_hh, b_ih, b_hh = m._layer_views(l)
            gw_ih, gw_hh, gb_ih, gb_hh = m._layer_views(l, gflat)
            dgx = torch.empty(T, B, D * 4 * H, device=dev, dtype=torch.float32)
            bias_done = ctypes.c_int32(0)       # (the one-launch recurrence sums its d gates over the frames on the way)
            _lib.check(L.pk2_lstm_layer_bwd_bias(_p(dy), _p(w_hh), _p(gates), _p(cells), B, T, H, D, _p(dgx), _p(scratch),
                                                 _p(gb_ih), _p(gb_hh), ctypes.byref(bias_done), sp))
            G = D * 4 * H

            def layer_grads(dgx=dgx, inp=inp, y=y, in_size=in_size, gw_ih=gw_ih, gw_hh=gw_hh, gb_ih=gb_ih,
                            gb_hh=gb_hh, l=l, bias_done=bool(bias_done.value)):
                # dW_ih (both directions at once) = dgx^T inp; the bias gradients (b_ih and b_hh receive the same sum) with it
                # unless the one-launch recurrence has summed them already
                if not bias_done:
                    _lib.check(L.pk2_gemm_f32_tn_colsum(G, in_size, rows, 1.0, _p(dgx), G, _p(inp), in_size, 1.0, _p(gw_ih), in_size,
                                                        _p(gb_ih), _lib.stream_ptr()))
                    gb_hh.copy_(gb_ih)
                else:
                    _gemm(1, 0, G, in_size, rows, _p(dgx), G, _p(inp), in_size, _p(gw_ih), in_size, beta=1.0)
                # dW_hh[d] = sum_t dg_d[t]^T h_d[t-1] (reverse direction: h_d[t+1]); time-major => row shift by B
                if T > 1:
                    k = (T - 1) * B
                    if D == 2:
                        # both directions in one batched launch (matrix 1 = matrix 0 + these strides: the reverse
                        # direction pairs dg[t] with h[t+1])
                        _lib.check(L.pk2_gemm_f32_batched(1, 0, 4 * H, H, k, 1.0, _p(dgx, B * G), G, 4 * H - B * G, 0,
                                                          _p(y), D * H, B * D * H + H, 0, 1.0, _p(gw_hh), H,
                                                          4 * H * H, 0, 2, 1, _lib.stream_ptr()))
                    else:
                        _gemm(1, 0, 4 * H, H, k, _p(dgx, B * G), G, _p(y), D * H, _p(gw_hh), H, beta=1.0)
                m._bucket_ready("lstm.l%d" % l)
            # critical path first (next layer's dy), weight gradients on the side stream
            if l > 0 or ctx.need_dx:
                dprev = torch.empty(T, B, in_size, device=dev, dtype=torch.float32)
                ev = torch.cuda.Event()
                ev.record(main)
                _gemm(0, 0, rows, in_size, G, _p(dgx), G, _p(w_ih), in_size, _p(dprev), in_size)
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    layer_grads()
                for t_ in (dgx, inp, y):
                    t_.record_stream(side)
                if l > 0:
                    dy = dprev
                    if ctx.seeds[l - 1] is not None:  # same mask and scale as the forward dropout of layer l-1
                        _lib.check(L.pk2_dropout_f32(_p(dy), _p(dy), dy.numel(), float(m.dropout), ctx.seeds[l - 1], sp))
                else:
                    dx = dprev
            else:
                on_side(layer_grads, dgx, inp, y)
        main.wait_stream(side)
        # Parameter gradients are published directly as views of the flat gradient buffer
        # (p.grad = gradient of THIS backward, which is what zero_grad -> backward -> step needs);
        # autograd gets None for them so nothing is copied or double-counted.
        params = dict(m.named_parameters())
        for name in m._param_names:
            p, v = params[name], views[name]
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v
        ctx.saved = None
        return (dx, None) + (None,) * len(m._param_names)


class LSTMAM(nn.Module):
    def __init__(self, input_size, output_size, hidden_size, num_layers, dropout, bidirectional):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.dropout = dropout
        self.bidirectional = bidirectional
        self.num_dirs = 2 if bidirectional else 1
        # same creation order as the reference (output layer first, then the LSTM)
        self.output_layer = _LinearParams(hidden_size * self.num_dirs, output_size)
        self.lstm = _LSTMParams(input_size, hidden_size, num_layers, bidirectional)
        self._param_names = [n for n, _ in self.named_parameters()]
        self._flat = None
        self._gflat = None
        self._layout = None
        self._bucket_hook = None

    # ---- flat parameter / gradient storage -------------------------------------------
    def _flat_order(self):
        """Flat layout: output layer, then per layer [w_ih f,r][w_hh f,r][b_ih f,r][b_hh f,r]."""
        order = ["output_layer.weight", "output_layer.bias"]
        sfx = ["", "_reverse"][:self.num_dirs]
        for l in range(self.num_layers):
            for kind in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                for s in sfx:
                    order.append("lstm.%s_l%d%s" % (kind, l, s))
        return order

    def _ensure_flat(self):
        params = dict(self.named_parameters())
        first = params["output_layer.weight"]
        if self._flat is not None and self._flat.device == first.device and all(
                params[n].data_ptr() == self._flat.data_ptr() + 4 * off for n, (off, _) in self._layout.items()):
            return
        order = self._flat_order()
        layout, off = {}, 0
        for n in order:
            cnt = params[n].numel()
            layout[n] = (off, cnt)
            off += (cnt + 63) // 64 * 64  # 256-byte aligned segments
        flat = torch.zeros(off, dtype=torch.float32, device=first.device)
        for n in order:
            o, cnt = layout[n]
            flat[o:o + cnt].copy_(params[n].data.reshape(-1))
            params[n].data = flat[o:o + cnt].view(params[n].shape)
        self._flat, self._layout = flat, layout
        self._gflat = None
        self._buckets = {}
        names = order
        def span(prefixes):
            sel = [layout[n] for n in names if any(n.startswith(p) for p in prefixes)]
            lo = min(o for o, _ in sel)
            hi = max((o + c + 63) // 64 * 64 for o, c in sel)
            return lo, hi
        self._buckets["output_layer"] = span(["output_layer."])
        for l in range(self.num_layers):
            self._buckets["lstm.l%d" % l] = span(["lstm.%s_l%d" % (k, l) for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")])

    def flat_parameters(self):
        """(flat_param, flat_grad) 1-D tensors covering every parameter (fused optimiser / all-reduce)."""
        self._ensure_flat()
        return self._flat, self._grad_flat()

    def _grad_flat(self):
        self._ensure_flat()
        if self._gflat is None or self._gflat.device != self._flat.device:
            self._gflat = torch.zeros_like(self._flat)
        return self._gflat

    def _grad_views(self, gflat):
        params = dict(self.named_parameters())
        return {n: gflat[o:o + c].view(params[n].shape) for n, (o, c) in self._layout.items()}

    def _layer_views(self, l, flat=None):
        """(w_ih [D*4H,in], w_hh [D*4H,H], b_ih [D*4H], b_hh [D*4H]) of layer l, both directions adjacent."""
        flat = self._flat if flat is None else flat
        D, H = self.num_dirs, self.hidden_size
        in_size = self.input_size if l == 0 else H * D
        def seg(kind, cols):
            o, _ = self._layout["lstm.%s_l%d" % (kind, l)]
            n = D * 4 * H * cols
            return flat[o:o + n].view(D * 4 * H, cols) if cols > 1 else flat[o:o + D * 4 * H]
        return seg("weight_ih", in_size), seg("weight_hh", H), seg("bias_ih", 1), seg("bias_hh", 1)

    def _side_stream(self, dev):
        """The stream of the weight-gradient products under PK2_SIDE_STREAM=1; with PK2_SIDE_CU_PER_XCD=k a stream whose
        kernels only get the first k CUs of every XCD (pk2_stream_create_cu_mask), so that they share fewer CUs with the
        one-launch recurrences of the main stream (measured, round 5)."""
        st = getattr(self, "_side", None)
        if st is None or st.device != dev:
            k = int(os.environ.get("PK2_SIDE_CU_PER_XCD", "0"))
            if k > 0:
                h = ctypes.c_void_p()
                with torch.cuda.device(dev):
                    _lib.check(_lib.lib().pk2_stream_create_cu_mask(k, ctypes.byref(h)))
                st = torch.cuda.ExternalStream(h.value, device=dev)
                self._side_handle = h        # (lives as long as the model)
            else:
                st = torch.cuda.Stream(device=dev)
            self._side = st
        return st

    def _bucket_ready(self, name):
        if self._bucket_hook is not None:
            lo, hi = self._buckets[name]
            self._bucket_hook(name, self._gflat[lo:hi])

    # ---- forward ---------------------------------------------------------------------
    def forward_time_major(self, x_tm):
        """x_tm [T,B,D] -> logits [T,B,P] (time-major, the kernels' native layout)."""
        _lib.require_gpu()
        assert x_tm.is_cuda and x_tm.dtype == torch.float32
        self._ensure_flat()
        D, H = self.num_dirs, self.hidden_size
        for l in range(self.num_layers):  # adjacency of the two directions is what the GEMMs rely on
            in_size = self.input_size if l == 0 else H * D
            assert (4 * H * in_size) % 64 == 0 and (4 * H * H) % 64 == 0 and (4 * H) % 64 == 0
        params = [p for _, p in self.named_parameters()]
        return _LstmAmFunction.apply(x_tm, self, *params)

    def forward(self, data):
        """data [B,T,D] -> logits [B,T,P], contiguous like the reference's (callers do
        ``prediction.view(-1, P)``, bin/train_ce.py:189).  The copy out of the kernels' time-major
        layout is skipped by callers that use forward_time_major()."""
        x_tm = data.transpose(0, 1).contiguous()
        return self.forward_time_major(x_tm).transpose(0, 1).contiguous()
