"""Fused optimisers over the model's flat parameter / gradient buffers (libpk2hip.so).

Adam(amsgrad=True) and SGD(momentum) with the reference's call pattern
(reference bin/train_ce.py:123,195-196; bin/train_chain.py:138,281-288; bin/train_se.py:127,253-257):

    optimizer.zero_grad(); loss.backward()
    norm = clip_grad_norm_(model, max_norm)      # device scalar, no host sync
    optimizer.step()

``clip_grad_norm_`` only measures; the clip coefficient min(1, max_norm/(norm+1e-6)) is applied
inside the update kernel, which reads the norm from device memory.  ``param_groups[0]['lr']`` can
be rewritten between steps (Noam schedule, bin/train_chain.py:281-284).
"""
import torch

from . import _lib


class _FlatOptimizer:
    def __init__(self, model, lr, weight_decay=0.0):
        self.model = model
        self.param_groups = [dict(lr=lr, weight_decay=weight_decay, params=list(model.parameters()))]
        self._norm = None
        self._max_norm = 0.0
        self._ws = None
        self.grad_scale = 1.0
        self.step_count = 0
        # several ranks: hvd.DistributedOptimizer checks the guard of ALL ranks itself (same step on every rank) and
        # switches this local, unsynchronised check off -- a rank that raises alone leaves its peers in the next all-reduce
        self.guard_check = True

    def zero_grad(self, set_to_none=True):
        # gradients are overwritten by the next backward (LSTMAM publishes p.grad itself)
        self._norm = None
        self._max_norm = 0.0

    def _flat(self):
        return self.model.flat_parameters()

    # ---- torch.optim-format state dicts ---------------------------------------------------------------------------
    # The reference's checkpoints hold torch.optim state dicts (bin/train_ce.py:160-165: {'model', 'optimizer',
    # 'epoch'}): per-parameter state keyed by the index of the parameter in model.parameters() order plus
    # param_groups.  The flat buffers here are views of the same parameters, so both directions are a slicing job.
    def _segments(self):
        """[(offset, count, shape)] of every parameter inside the flat buffers, in model.parameters() order."""
        self._flat()
        layout = self.model._layout
        return [(layout[n][0], layout[n][1], tuple(p.shape)) for n, p in self.model.named_parameters()]

    def _to_torch_state(self, tensors, step):
        """tensors: {key: flat tensor or None} -> {index: {key: per-parameter copy, 'step': tensor}}"""
        out = {}
        for i, (o, c, shape) in enumerate(self._segments()):
            st = {k: v[o:o + c].view(shape).clone() for k, v in tensors.items() if v is not None}
            if step is not None:
                st["step"] = torch.tensor(float(step))
            out[i] = st
        return out

    def _from_torch_state(self, state, keys):
        """Inverse: gathers per-parameter state tensors into flat buffers; returns ({key: flat}, step)."""
        p, _ = self._flat()
        flat = {k: torch.zeros_like(p) for k in keys}
        step = 0
        for i, (o, c, shape) in enumerate(self._segments()):
            st = state.get(i, state.get(str(i)))
            if st is None:
                continue
            for k in keys:
                if k in st and st[k] is not None:
                    flat[k][o:o + c].copy_(torch.as_tensor(st[k]).reshape(-1).to(p.device))
            if "step" in st:
                step = max(step, int(float(st["step"])))
        return flat, step

    @staticmethod
    def _is_torch_format(sd):
        st = sd.get("state")
        return isinstance(st, dict) and "param_groups" in sd and all(isinstance(k, int) or str(k).isdigit() for k in st)

    def measure_grad_norm(self, max_norm):
        p, g = self._flat()
        L = _lib.lib()
        if getattr(self, "_norm_buf", None) is None or self._norm_buf.device != g.device:
            self._norm_buf = torch.empty(1, dtype=torch.float32, device=g.device)
        if self._ws is None or self._ws.device != g.device:
            self._ws = torch.empty(L.pk2_grad_norm_workspace_bytes(g.numel()), dtype=torch.uint8, device=g.device)
        _lib.check(L.pk2_grad_norm(_lib.ptr(g), g.numel(), _lib.ptr(self._norm_buf), _lib.ptr(self._ws),
                                   self._ws.numel(), _lib.stream_ptr(g.device)))
        self._norm = self._norm_buf
        self._max_norm = float(max_norm)
        # the norm of the averaged gradient is what is clipped when grad_scale != 1
        return self._norm if self.grad_scale == 1.0 else self._norm * self.grad_scale


def clip_grad_norm_(optimizer, max_norm):
    """Device-side replacement for nn.utils.clip_grad_norm_(model.parameters(), max_norm): returns
    the total norm (0-dim CUDA tensor); the scaling happens inside optimizer.step()."""
    return optimizer.measure_grad_norm(max_norm)[0]


class Adam(_FlatOptimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        super().__init__(model, lr, weight_decay)
        self.betas, self.eps, self.amsgrad = betas, eps, amsgrad
        self.state = None

    def step(self):
        if self.guard_check:
            _lib.check_persist_guard("Adam.step")     # (no synchronisation; raises at the first step after a time-out)
        p, g = self._flat()
        if self.state is None or self.state["exp_avg"].device != p.device:
            self.state = dict(exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p),
                              max_exp_avg_sq=torch.zeros_like(p) if self.amsgrad else None)
        self.step_count += 1
        grp = self.param_groups[0]
        # clipping acts on the *scaled* gradient: ||s*g|| = s*||g||  ->  compare max_norm/s against ||g||
        max_norm = self._max_norm / self.grad_scale if self._norm is not None and self._max_norm > 0 else 0.0
        _lib.check(_lib.lib().pk2_adam_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(self.state["exp_avg"]),
                                            _lib.ptr(self.state["exp_avg_sq"]),
                                            _lib.ptr(self.state["max_exp_avg_sq"]), p.numel(), float(grp["lr"]),
                                            float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                            float(grp["weight_decay"]), self.step_count, float(max_norm),
                                            _lib.ptr(self._norm) if self._norm is not None else None,
                                            float(self.grad_scale), _lib.stream_ptr(p.device)))

    def state_dict(self):
        """torch.optim.Adam's format (what the reference's checkpoints hold and torch.optim.Adam.load_state_dict takes)."""
        grp = self.param_groups[0]
        n = len(self._segments())
        state = {} if self.state is None else self._to_torch_state(
            dict(exp_avg=self.state["exp_avg"], exp_avg_sq=self.state["exp_avg_sq"],
                 max_exp_avg_sq=self.state["max_exp_avg_sq"]), self.step_count)
        return dict(state=state, param_groups=[dict(lr=grp["lr"], betas=tuple(self.betas), eps=self.eps,
                                                    weight_decay=grp["weight_decay"], amsgrad=self.amsgrad,
                                                    maximize=False, foreach=None, capturable=False, differentiable=False,
                                                    fused=None, params=list(range(n)))])

    def load_state_dict(self, sd):
        if not self._is_torch_format(sd):      # round-1 checkpoints of this repo: flat tensors
            self.state, self.step_count = sd["state"], sd["step"]
            self.param_groups[0].update(sd["param_groups"][0])
            return
        g = sd["param_groups"][0]
        self.param_groups[0].update(lr=g["lr"], weight_decay=g.get("weight_decay", 0.0))
        self.betas, self.eps = tuple(g.get("betas", self.betas)), g.get("eps", self.eps)
        self.amsgrad = bool(g.get("amsgrad", self.amsgrad))
        if not sd["state"]:
            self.state, self.step_count = None, 0
            return
        keys = ["exp_avg", "exp_avg_sq"] + (["max_exp_avg_sq"] if self.amsgrad else [])
        flat, self.step_count = self._from_torch_state(sd["state"], keys)
        self.state = dict(exp_avg=flat["exp_avg"], exp_avg_sq=flat["exp_avg_sq"], max_exp_avg_sq=flat.get("max_exp_avg_sq"))


class SGD(_FlatOptimizer):
    def __init__(self, model, lr, momentum=0.0, weight_decay=0.0):
        super().__init__(model, lr, weight_decay)
        self.momentum = momentum
        self.buf = None

    def step(self):
        if self.guard_check:
            _lib.check_persist_guard("SGD.step")
        p, g = self._flat()
        first = self.buf is None
        if self.momentum != 0 and first:
            self.buf = torch.zeros_like(p)
        self.step_count += 1
        grp = self.param_groups[0]
        max_norm = self._max_norm / self.grad_scale if self._norm is not None and self._max_norm > 0 else 0.0
        _lib.check(_lib.lib().pk2_sgd_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(self.buf), p.numel(),
                                           float(grp["lr"]), float(self.momentum), float(grp["weight_decay"]),
                                           1 if first else 0, float(max_norm),
                                           _lib.ptr(self._norm) if self._norm is not None else None,
                                           float(self.grad_scale), _lib.stream_ptr(p.device)))

    def state_dict(self):
        """torch.optim.SGD's format (momentum_buffer per parameter)."""
        grp = self.param_groups[0]
        n = len(self._segments())
        state = {} if self.buf is None else self._to_torch_state(dict(momentum_buffer=self.buf), None)
        return dict(state=state, param_groups=[dict(lr=grp["lr"], momentum=self.momentum, dampening=0,
                                                    weight_decay=grp["weight_decay"], nesterov=False, maximize=False,
                                                    foreach=None, differentiable=False, fused=None,
                                                    params=list(range(n)))])

    def load_state_dict(self, sd):
        if not self._is_torch_format(sd):
            self.buf, self.step_count = sd["buf"], sd["step"]
            self.param_groups[0].update(sd["param_groups"][0])
            return
        g = sd["param_groups"][0]
        self.param_groups[0].update(lr=g["lr"], weight_decay=g.get("weight_decay", 0.0))
        self.momentum = g.get("momentum", self.momentum)
        if not sd["state"]:
            self.buf, self.step_count = None, 0
            return
        flat, _ = self._from_torch_state(sd["state"], ["momentum_buffer"])
        self.buf, self.step_count = flat["momentum_buffer"], 1      # the buffer exists: not the first step
