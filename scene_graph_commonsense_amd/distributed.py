"""Data-parallel glue: one process per GPU, images sharded across ranks, gradients all-reduced over RCCL.

The reference wraps the classifier in DDP over gloo and lets ``losses.backward()`` all-reduce the 1.1 GB of
f32 gradients (``train_test.py:28,72-80,276``).  Here the backward is explicit, so the all-reduce is too:
``fc1.weight``'s gradient (97 % of the bytes) is handed to RCCL as soon as the fc1 weight-gradient GEMM has
been enqueued and runs on RCCL's stream under the remaining conv3/conv2/conv1 backward; the other parameters
go in one flat bucket at the end.  There is no data-path collective (images and their pairs are independent).
"""
from __future__ import annotations

import os
from typing import Dict, List

import torch
import torch.distributed as dist


def init_from_env(backend: str = None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)          # bind the rank to its GPU before RCCL creates its communicator
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


class GradReducer:
    """Mean-reduce one step's gradients across ranks, with an early asynchronous launch for the large tensors.

    ``hook(name, grad)`` is called by the backward as soon as a large gradient has been enqueued (``fc1.weight``: 97 % of the
    bytes): RCCL all-reduces it in place on its own stream while the rest of the backward runs.  ``finish_grads(grads)``
    reduces every other tensor of the step in one flat bucket, waits for the early ones and divides by the world size.  The
    step's gradients are reduced BEFORE they are accumulated into ``param.grad`` (``model.training_step``), so gradient
    accumulation over several steps sums mean gradients exactly as DDP + autograd would and never reads a tensor RCCL is
    still writing."""

    def __init__(self, world: int, early=("fc1.weight",), force_collectives: bool = False):
        self.world = world
        self.active = world > 1 or force_collectives   # forced: a one-rank RCCL group still runs every collective (tests)
        self.early = set(early)
        self.pending: List = []          # (work handle, tensor) of the early reductions in flight
        self.done = set()
        self.exposed_events: List = []   # (start, end) event pairs around the waits on the compute stream

    def hook(self, name: str, grad: torch.Tensor):
        if not self.active or name not in self.early or name in self.done:
            return
        self.pending.append((dist.all_reduce(grad, op=dist.ReduceOp.SUM, async_op=True), grad))
        self.done.add(name)

    def _drain(self):
        for work, g in self.pending:
            timed = g.is_cuda
            if timed:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            work.wait()                  # orders the compute stream after RCCL's: later kernels see the reduced tensor
            if timed:
                b.record()
                self.exposed_events.append((a, b))
            g.div_(self.world)
        self.pending.clear()
        self.done.clear()

    def pop_exposed_ms(self) -> float:
        """Milliseconds the compute stream waited for the early all-reduces since the last call (after a device synchronise).  The
        flat bucket of the small tensors is a synchronous collective on top of that (34 MB)."""
        t = sum(a.elapsed_time(b) for a, b in self.exposed_events)
        self.exposed_events = []
        return float(t)

    def _flat(self, tensors: List[torch.Tensor]):
        if not tensors:
            return
        flat = torch.cat([g.reshape(-1) for g in tensors])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(self.world)
        off = 0
        for g in tensors:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def finish_grads(self, grads: Dict[str, torch.Tensor]):
        """Reduce the step's gradient dict in place: everything not handed to ``hook`` goes in one flat bucket."""
        if self.active:
            for k in grads:
                if not grads[k].is_contiguous():
                    grads[k] = grads[k].contiguous()
            self._flat([g for n, g in grads.items() if n not in self.done])
        self._drain()

    def finish(self, named_params):
        """Same on ``param.grad`` (for callers that wrote the step's gradients there themselves)."""
        if self.active:
            self._flat([p.grad for n, p in named_params if p.grad is not None and n not in self.done])
        self._drain()


def allreduce_counters(values: torch.Tensor) -> torch.Tensor:
    """Sum integer hit/target counters of the evaluator across ranks (a few hundred bytes)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM)
    return values


# ------------------------------------------------------------------------------------------------ sharded data parallelism
def _hip_sgd_update(p, g, m, lr, momentum, weight_decay, first):
    """One-pass SGD-momentum update of a contiguous f32 slice through the HIP kernel (``sgc_sgd_momentum_step``); GPU only."""
    import ctypes
    from . import _lib
    if not p.is_cuda:
        raise RuntimeError("ShardedSGD updates parameters with the HIP kernel: GPU tensors only (no CPU fallback)")
    f = ctypes.c_float
    _lib.check(_lib.load().sgc_sgd_momentum_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), ctypes.c_long(p.numel()), f(lr), f(momentum),
                                                 f(weight_decay), int(first), _lib.stream_ptr()), "sgc_sgd_momentum_step")


def _hip_sgd_fc1_update(p, g, m, lr, momentum, weight_decay, first):
    """The same update for whole rows of fc1.weight with the gradient in GEMM order (``sgc_sgd_fc1_fused`` without the f16 copy: the
    compute copies are made from the gathered masters)."""
    import ctypes
    from . import _lib
    if not p.is_cuda or p.numel() % 65536:
        raise RuntimeError("fused fc1 update: whole rows of a GPU tensor")
    f = ctypes.c_float
    _lib.check(_lib.load().sgc_sgd_fc1_fused(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), p.numel() // 65536, f(lr), f(momentum), f(weight_decay),
                                             int(first), None, _lib.stream_ptr()), "sgc_sgd_fc1_fused")


class _Piece:
    """One shard this rank owns: ``length`` elements at ``offset`` of a flat f32 buffer (a big parameter's storage or the small
    parameters' bucket), with its momentum buffer and the step's (accumulated) mean gradient."""
    __slots__ = ("key", "offset", "length", "bucket_off", "bucket_len", "mom", "acc", "first")

    def __init__(self, key, offset, length, bucket_off, bucket_len):
        self.key, self.offset, self.length, self.bucket_off, self.bucket_len = key, offset, length, bucket_off, bucket_len
        self.mom, self.acc, self.first = None, None, True


class ShardedSGD:
    """Gradient reducer AND optimizer of the data-parallel relation head: reduce-scatter of the f32 gradients, SGD-momentum on
    the shard this rank owns, all-gather of the updated f32 parameters (the reference: DDP all-reduce + a full
    ``torch.optim.SGD`` step on every rank, ``train_test.py:72-80,100,276``).

    Against all-reduce + full update (``GradReducer`` + ``optim.FusedSGD``, still available: ``bench.py --dp-mode allreduce``)
    the wire bytes are the same (a ring all-reduce IS a reduce-scatter + an all-gather) but (a) the optimizer pass touches 1/W
    of the parameters and the momentum buffer is 1/W of 1.1 GB per rank, (b) the two halves are separate collectives: the
    reduce-scatter overlaps the rest of the backward exactly like the all-reduce did, and the all-gather of a row block starts
    as soon as that block's shard is updated (asynchronous collectives on RCCL's stream, waited for once at the end of
    ``step``).  ``fc1.weight`` (97 % of the bytes) is cut into ``buckets`` row blocks that are reduce-scattered / gathered as
    separate collectives, so the optimizer works on the first block while the last is on the wire.  The f32 master parameters
    stay REPLICATED and bit-identical on every rank (checkpoints, ``state_dict`` and the 16-bit re-cast work as before).
    ``defer_gather=True`` + ``attach(model)``: the wait for fc1.weight's gather moves to the point of the NEXT forward where its
    16-bit copies are made (right before fc1, ~12 ms into the step).  Not done: gathering 16-bit compute copies instead of f32
    (halves the all-gather, but every rank would hold stale masters outside its shard).

    Interface: ``hook`` / ``finish_grads`` (what ``model.training_step(reducer=...)`` calls; ``owns_grads`` tells it not to
    write ``param.grad`` - no rank holds the full mean gradient), ``zero_grad`` / ``step`` / ``param_groups`` (what the training
    loops call on an optimizer).  Gradient accumulation over several ``finish_grads`` sums MEAN gradient shards.  Update rule =
    ``torch.optim.SGD(lr, momentum, weight_decay)`` (dampening 0, no Nesterov), the kernel of ``optim.FusedSGD``."""

    owns_grads = True

    def __init__(self, named_params, world: int = 1, rank: int = 0, lr: float = 1e-3, momentum: float = 0.0, weight_decay: float = 0.0,
                 big=("fc1.weight",), buckets: int = 8, update_fn=None, group=None, force_collectives: bool = False,
                 defer_gather: bool = False):
        self.named = [(n, p) for n, p in named_params]
        self.world, self.rank, self.group = int(world), int(rank), group
        self.collective = self.world > 1 or bool(force_collectives)   # forced: a one-rank RCCL group still runs every collective (tests)
        # deferred: ``step`` returns while the all-gathers of the big parameters are still on the wire; whoever reads those parameters
        # next calls ``wait_gathers`` first - the classifier does (``attach``: the 16-bit copies of fc1.weight are made right before fc1
        # runs, ``engine.Weights``), so the gather overlaps the next forward up to fc1.  Anything else that reads the parameters on the
        # compute stream (checkpoints, ``state_dict``) must call ``wait_gathers()`` itself; ``torch.cuda.synchronize()`` also suffices.
        self.defer_gather = bool(defer_gather)
        self.param_groups = [dict(lr=lr, momentum=momentum, weight_decay=weight_decay, params=[p for _, p in self.named])]
        self.update = update_fn or _hip_sgd_update
        self.big = [n for n, p in self.named if n in set(big) and p.numel() % (self.world * 4) == 0]
        self.small = [(n, p) for n, p in self.named if n not in self.big]
        self.pieces: Dict[str, List[_Piece]] = {}
        W = self.world
        for n, p in self.named:
            if n not in self.big:
                continue
            if not p.is_contiguous():
                raise RuntimeError("ShardedSGD needs contiguous parameters")
            nb = max(1, int(buckets))
            while nb > 1 and p.numel() % (nb * W * 4) != 0:
                nb //= 2
            bl = p.numel() // nb
            self.pieces[n] = [_Piece(n, k * bl + self.rank * (bl // W), bl // W, k * bl, bl) for k in range(nb)]
        self.small_numel = sum(p.numel() for _, p in self.small)
        self.small_pad = (self.small_numel + 4 * W - 1) // (4 * W) * (4 * W)
        sl = self.small_pad // W
        self.pieces["__small__"] = [_Piece("__small__", self.rank * sl, sl, 0, self.small_pad)] if self.small else []
        self.pending: List = []            # (work, shard tensor, piece) reduce-scatters in flight
        self.hooked = set()
        self.gathers: List = []            # (work, keep-alive, is a big parameter's) all-gathers in flight
        self._small_flat = None
        self._keep: List = []
        self.exposed_events: List = []     # (start, end) event pairs around every wait on the compute stream
        self._native_rs = None
        # True (set by ``fuse_fc1``, sticky): EVERY fc1.weight gradient handed to this object is in GEMM order [4096][window*1024 +
        # channel] - the backward skips its transposition to the reference order (0.4 ms in front of the first reduce-scatter), the
        # collectives move the rows as they are (the permutation is inside a row, the shards are whole rows) and the shard update
        # un-permutes on the fly (``sgc_sgd_fc1_fused``)
        self.fc1_gemm_order = False
        self.update_fc1 = _hip_sgd_fc1_update

    def fuse_fc1(self, model):
        """Called by ``attach`` (and asked again by ``pair_loop.train_minibatch``); returns the engine whose backward keeps fc1.weight's
        gradient in GEMM order, or None (shapes other than the reference's, shards that are not whole rows, an injected CPU update rule,
        a module this object was not attached to).  Once on, ``model.training_step(reducer=self)`` sets the engine's switch itself."""

        eng = self._fc1_fusable(model)
        if eng is not None:
            self.fc1_gemm_order = True
        return eng

    def _fc1_fusable(self, model):
        """The engine ``fuse_fc1`` would switch, or None; changes nothing (``attach`` asks every rank before any of them switches)."""
        fc1 = getattr(model, "fc1", None)
        if fc1 is None or "fc1.weight" not in self.big or tuple(fc1.weight.shape) != (4096, 65536) or not fc1.weight.is_cuda:
            return None
        if self.update is not _hip_sgd_update or dict(self.named).get("fc1.weight") is not fc1.weight:
            return None
        if any(pc.length % 65536 or pc.offset % 65536 for pc in self.pieces["fc1.weight"]):
            return None
        eng = model.engine()
        if type(eng).__name__ != "RelHeadEngine":
            return None
        if not self.fc1_gemm_order and any(pc.acc is not None for pc in self.pieces["fc1.weight"]):
            return None                    # a reference-order gradient is being accumulated: not in the middle of it
        return eng

    # ------------------------------------------------------------------ collectives
    def _reduce_scatter(self, out, inp, async_op):
        """out = this rank's 1/W of sum over ranks of inp.  Backends without a native reduce-scatter for this tensor type (gloo
        with GPU tensors in the one-GPU tests) fall back to all-reduce + slice: same values."""
        if not self.collective:
            out.copy_(inp)
            return None
        if self._native_rs is not False:
            try:
                w = dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
                self._native_rs = True
                return w
            except (RuntimeError, NotImplementedError):
                if self._native_rs:
                    raise
                self._native_rs = False
        tmp = inp.clone()
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group)
        n = out.numel()
        out.copy_(tmp[self.rank * n:(self.rank + 1) * n])
        return None

    def _all_gather(self, out, src):
        """out (flat, W x src.numel()) = the ranks' ``src`` in rank order; returns a work handle or None when already complete.
        Same fallback rule as ``_reduce_scatter`` (list form of all-gather + copies)."""
        if getattr(self, "_native_ag", None) is not False:
            try:
                w = dist.all_gather_into_tensor(out, src, group=self.group, async_op=True)
                self._native_ag = True
                return w
            except (RuntimeError, NotImplementedError):
                if getattr(self, "_native_ag", None):
                    raise
                self._native_ag = False
        parts = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(parts, src.contiguous(), group=self.group)
        n = src.numel()
        for k, t in enumerate(parts):
            out[k * n:(k + 1) * n].copy_(t)
        return None

    def _wait(self, work):
        if work is None:
            return
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            work.wait()
            b.record()
            self.exposed_events.append((a, b))
        else:
            work.wait()

    def pop_exposed_ms(self) -> float:
        """Milliseconds the compute stream spent waiting for collectives since the last call (call after a device synchronise)."""
        t = sum(a.elapsed_time(b) for a, b in self.exposed_events)
        self.exposed_events = []
        return float(t)

    # ------------------------------------------------------------------ reducer half
    def hook(self, name: str, grad: torch.Tensor):
        """Called by the backward when a big gradient has been enqueued: reduce-scatter its row blocks asynchronously."""
        if name not in self.pieces or name in self.hooked or name == "__small__":
            return
        g = grad.reshape(-1)
        for pc in self.pieces[name]:
            if not self.collective:
                self.pending.append((None, g[pc.offset:pc.offset + pc.length], pc))
                continue
            out = torch.empty(pc.length, dtype=g.dtype, device=g.device)
            w = self._reduce_scatter(out, g[pc.bucket_off:pc.bucket_off + pc.bucket_len], async_op=True)
            self.pending.append((w, out, pc))
        self._keep.append(grad)                                   # the collective reads the gradient until it is waited for
        self.hooked.add(name)

    def finish_grads(self, grads: Dict[str, torch.Tensor]):
        """Reduce-scatter everything of this step that ``hook`` has not taken, wait, and add the MEAN shards to the accumulators."""
        for n in self.big:
            if n not in self.hooked:
                self.hook(n, grads[n] if grads[n].is_contiguous() else grads[n].contiguous())
        if self.small:
            dev, dt = grads[self.small[0][0]].device, grads[self.small[0][0]].dtype
            flat = torch.zeros(self.small_pad, dtype=dt, device=dev)
            off = 0
            for n, p in self.small:
                flat[off:off + p.numel()].copy_(grads[n].reshape(-1))
                off += p.numel()
            pc = self.pieces["__small__"][0]
            out = torch.empty(pc.length, dtype=dt, device=dev)
            self._reduce_scatter(out, flat, async_op=False)
            self.pending.append((None, out, pc))
        inv = 1.0 / self.world
        for work, shard, pc in self.pending:
            self._wait(work)
            if pc.acc is None:        # world 1: the shard is a view of the engine's gradient buffer (reused next step) - own it
                pc.acc = shard.clone() if not self.collective else shard.mul_(inv)
            else:
                pc.acc.add_(shard, alpha=inv)
        self.pending.clear()
        self.hooked.clear()
        self._keep = []

    # ------------------------------------------------------------------ optimizer half
    def zero_grad(self, set_to_none: bool = True):
        for pcs in self.pieces.values():
            for pc in pcs:
                pc.acc = None
        for _, p in self.named:
            p.grad = None

    def attach(self, model):
        """Let ``model`` (a classifier of ``model.py``) wait for this optimizer's parameter gathers before it reads ``fc1.weight``.
        With ``defer_gather`` ``step()`` returns while RCCL may still be writing ``fc1.weight``'s storage; the forward's weight-copy
        refresh waits through ``model.weight_sync``; every OTHER reader on the compute stream goes through ``state_dict()``
        (checkpoints, EMA snapshots, ``load_state_dict`` round trips), which gets a pre-hook that waits too.  Code that reads
        ``model.parameters()`` directly between ``step()`` and the next forward must call ``wait_gathers()`` itself."""
        model.__dict__["weight_sync"] = self.wait_gathers
        from .engine import TUNING
        # fc1.weight's gradient in GEMM order from now on, on every path of this model that feeds this object - but only if EVERY rank
        # can (the flag changes how the bytes of the shared reduce-scatter are read: a rank with another SGC_* environment, an injected
        # update rule or row blocks that are not whole rows must not mix its order into the collective)
        want = bool(TUNING.fused_sgd) and self._fc1_fusable(model) is not None
        agreed = want
        if self.collective and dist.is_available() and dist.is_initialized():
            fc1 = getattr(model, "fc1", None)
            on_dev = dist.get_backend(self.group) == "nccl" and fc1 is not None and fc1.weight.is_cuda
            flag = torch.tensor([int(want), -int(want)], dtype=torch.int32, device=fc1.weight.device if on_dev else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)          # (min, -max)
            agreed = bool(int(flag[0]))
            if int(flag[0]) != -int(flag[1]):
                import warnings
                warnings.warn("ShardedSGD.attach: the ranks disagree on the fused fc1 update (this rank: %s) - every rank keeps the "
                              "reference column order" % want)
        if agreed:
            self.fuse_fc1(model)
        if hasattr(model, "register_state_dict_pre_hook") and not getattr(model, "_sgc_gather_hook", False):
            opt = self
            model.register_state_dict_pre_hook(lambda module, prefix, keep_vars: opt.wait_gathers())
            model.__dict__["_sgc_gather_hook"] = True
        return self

    def wait_gathers(self, small_only: bool = False):
        """Block the compute stream until the parameter all-gathers of the last ``step`` have landed (``small_only``: only the
        flat bucket of the small parameters - what ``step`` itself waits for with ``defer_gather``)."""
        keep = []
        for work, src, big in self.gathers:
            if small_only and big:
                keep.append((work, src, big))
            else:
                self._wait(work)
        self.gathers[:] = keep
        if self._small_flat is not None:
            off = 0
            with torch.no_grad():
                for n, p in self.small:
                    p.view(-1).copy_(self._small_flat[off:off + p.numel()])
                    off += p.numel()
            self._small_flat = None

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closures are not supported")
        self.wait_gathers()
        g0 = self.param_groups[0]
        lr, mom, wd = float(g0["lr"]), float(g0["momentum"]), float(g0["weight_decay"])
        named = dict(self.named)
        # ---- big parameters: update this rank's piece of every row block in place, gather the block
        for n in self.big:
            flat = named[n].view(-1)
            for pc in self.pieces[n]:
                if pc.acc is None:
                    continue
                mine = flat[pc.offset:pc.offset + pc.length]
                if pc.mom is None:
                    pc.mom = torch.empty_like(mine)
                (self.update_fc1 if (self.fc1_gemm_order and n == "fc1.weight") else self.update)(mine, pc.acc, pc.mom, lr, mom, wd, pc.first)
                pc.first = False
                if self.collective:
                    src = mine.clone()        # a copy of the shard (1/W of the block): no aliasing of a collective's input and output
                    self.gathers.append((self._all_gather(flat[pc.bucket_off:pc.bucket_off + pc.bucket_len], src), src, True))
        # ---- small parameters: one flat bucket
        if self.small and self.pieces["__small__"][0].acc is not None:
            pc = self.pieces["__small__"][0]
            dev = pc.acc.device
            flat = torch.zeros(self.small_pad, dtype=pc.acc.dtype, device=dev)
            off = 0
            for n, p in self.small:
                flat[off:off + p.numel()].copy_(p.view(-1))
                off += p.numel()
            mine = flat[pc.offset:pc.offset + pc.length]
            if pc.mom is None:
                pc.mom = torch.empty_like(mine)
            self.update(mine, pc.acc, pc.mom, lr, mom, wd, pc.first)
            pc.first = False
            if self.collective:
                src = mine.clone()
                self.gathers.append((self._all_gather(flat, src), src, False))
            self._small_flat = flat
        for _, p in self.named:               # parameters change behind autograd's back: version-based caches must notice
            torch.autograd.graph.increment_version(p)
        # the classifier re-derives its 16-bit copies right after: the small parameters' gather must have landed; fc1.weight's may
        # stay in flight until its copies are made (``defer_gather`` + ``attach``)
        self.wait_gathers(small_only=self.defer_gather)
        return None
