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

    def zero_grad(self, set_to_none=True):
        # gradients are overwritten by the next backward (LSTMAM publishes p.grad itself)
        self._norm = None
        self._max_norm = 0.0

    def _flat(self):
        return self.model.flat_parameters()

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
        return dict(state=self.state, step=self.step_count,
                    param_groups=[{k: v for k, v in self.param_groups[0].items() if k != "params"}])

    def load_state_dict(self, sd):
        self.state, self.step_count = sd["state"], sd["step"]
        self.param_groups[0].update(sd["param_groups"][0])


class SGD(_FlatOptimizer):
    def __init__(self, model, lr, momentum=0.0, weight_decay=0.0):
        super().__init__(model, lr, weight_decay)
        self.momentum = momentum
        self.buf = None

    def step(self):
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
        return dict(buf=self.buf, step=self.step_count,
                    param_groups=[{k: v for k, v in self.param_groups[0].items() if k != "params"}])

    def load_state_dict(self, sd):
        self.buf, self.step_count = sd["buf"], sd["step"]
        self.param_groups[0].update(sd["param_groups"][0])
