"""One-pass SGD with momentum for the relation head (``sgc_sgd_momentum_step``).

Same update as ``torch.optim.SGD(params, lr, momentum, weight_decay)`` with dampening 0 and no Nesterov momentum - the optimizer
the reference builds in ``train_test.py`` - but one kernel per LARGE parameter tensor (one launch for all the small ones together) that reads the gradient, the weight and the
momentum buffer once and writes the weight and the buffer once (5 x 4 B per parameter; the foreach implementation makes three passes,
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
        self._fc1_engine = None

    def fuse_fc1(self, model):
        """Called by ``pair_loop.train_minibatch`` before the step: if this optimizer updates ``model.fc1.weight`` (the reference's
        [4096, 1024 * 64] layer on the tiled engine), return the engine whose backward may hand the gradient over in GEMM order - the
        update then runs as ``sgc_sgd_fc1_fused`` (gradient un-permuted on the fly, f16 compute copy of the forward written by the same
        pass: 6.4 instead of 10.7 GB of traffic per step) - else None."""
        fc1 = getattr(model, "fc1", None)
        if fc1 is None or tuple(fc1.weight.shape) != (4096, 65536) or not fc1.weight.is_cuda:
            return None
        if not any(p is fc1.weight for group in self.param_groups for p in group["params"]):
            return None
        eng = model.engine()
        if not hasattr(eng, "fc1_grad_gemm_order") or type(eng).__name__ != "RelHeadEngine":
            return None
        self._fc1_engine = eng
        return eng

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        f = ctypes.c_float
        SMALL = 1 << 20                      # tensors up to this many elements share one launch (sgc_sgd_momentum_multi), 32 at a time
        for group in self.param_groups:
            small = []
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
                if getattr(p, "_sgc_grad_gemm_order", False):
                    # fc1.weight with its gradient in GEMM order (engine.fc1_grad_gemm_order): fused un-permute + update + f16 copy
                    eng = self._fc1_engine
                    if eng is None or tuple(p.shape) != (4096, 65536):
                        raise RuntimeError("fc1.weight.grad is in GEMM order but this optimizer was not prepared by fuse_fc1()")
                    w1p = eng.ws.get("w1p", p.numel(), torch.float16)
                    _lib.check(self._lib.sgc_sgd_fc1_fused(_lib.ptr(p), _lib.ptr(g), _lib.ptr(st["momentum_buffer"]), 4096, f(group["lr"]),
                                                           f(group["momentum"]), f(group["weight_decay"]), int(first), _lib.ptr(w1p),
                                                           _lib.stream_ptr()), "sgc_sgd_fc1_fused")
                    torch.autograd.graph.increment_version(p)
                    eng._w1p_fresh = (p.data_ptr(), p._version)
                    p.grad = None                       # it was a view of the engine's scratch, in an order nobody else should read
                    p._sgc_grad_gemm_order = False
                    continue
                if p.numel() <= SMALL:
                    small.append((p, g, st["momentum_buffer"], first))
                else:
                    _lib.check(self._lib.sgc_sgd_momentum_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(st["momentum_buffer"]),
                                                               ctypes.c_long(p.numel()), f(group["lr"]), f(group["momentum"]),
                                                               f(group["weight_decay"]), int(first), _lib.stream_ptr()),
                               "sgc_sgd_momentum_step")
                # the kernel writes p behind autograd's back: bump its version counter (the classifier re-derives its 16-bit weight
                # copies when a parameter's version changes, and autograd's saved-tensor checks rely on it too)
                torch.autograd.graph.increment_version(p)
            for k in range(0, len(small), 32):
                part = small[k:k + 32]
                n = len(part)
                P, L = ctypes.c_void_p * n, ctypes.c_long * n
                mask = sum(1 << t for t, e in enumerate(part) if e[3])
                _lib.check(self._lib.sgc_sgd_momentum_multi(n, P(*[e[0].data_ptr() for e in part]), P(*[e[1].data_ptr() for e in part]),
                                                            P(*[e[2].data_ptr() for e in part]), L(*[e[0].numel() for e in part]),
                                                            f(group["lr"]), f(group["momentum"]), f(group["weight_decay"]),
                                                            ctypes.c_uint(mask), _lib.stream_ptr()), "sgc_sgd_momentum_multi")
        return loss
