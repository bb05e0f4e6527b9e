"""One-pass SGD with momentum for the relation head (``sgc_sgd_momentum_step``).

Same update as ``torch.optim.SGD(params, lr, momentum, weight_decay)`` with dampening 0 and no Nesterov momentum - the optimizer
the reference builds in ``train_test.py`` - but one kernel per parameter tensor that reads the gradient, the weight and the momentum
buffer once and writes the weight and the buffer once (5 x 4 B per parameter; the foreach implementation makes three passes,
9 x 4 B).  At 277 M parameters that is 2.4 ms -> 1.1 ms per step.  Subclasses ``torch.optim.Optimizer`` so that ``param_groups``
(the reference rescales ``lr`` inside its loop), ``state_dict`` and ``zero_grad`` behave as usual.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, momentum: float = 0.0, weight_decay: float = 0.0):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("lr, momentum and weight_decay must be non-negative")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self._lib = _lib.load()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        f = ctypes.c_float
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise RuntimeError("FusedSGD handles contiguous f32 parameters on the GPU (the HIP path has no fallback)")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                st = self.state[p]
                first = "momentum_buffer" not in st
                if first:
                    st["momentum_buffer"] = torch.empty_like(p, memory_format=torch.contiguous_format)
                _lib.check(self._lib.sgc_sgd_momentum_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(st["momentum_buffer"]),
                                                           ctypes.c_long(p.numel()), f(group["lr"]), f(group["momentum"]),
                                                           f(group["weight_decay"]), int(first), _lib.stream_ptr()),
                           "sgc_sgd_momentum_step")
                # the kernel wrote p behind autograd's back: bump its version counter (the classifier re-derives its 16-bit weight
                # copies when a parameter's version changes, and autograd's saved-tensor checks rely on it too)
                torch.autograd.graph.increment_version(p)
        return loss
