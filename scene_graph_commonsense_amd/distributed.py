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
    """Mean-reduce gradients across ranks with early launch for the large tensors."""

    def __init__(self, world: int, early=("fc1.weight",)):
        self.world = world
        self.early = set(early)
        self.pending: List = []
        self.done = set()

    def hook(self, name: str, grad: torch.Tensor):
        if self.world <= 1 or name not in self.early:
            return
        self.pending.append((dist.all_reduce(grad, op=dist.ReduceOp.SUM, async_op=True), grad))
        self.done.add(name)

    def finish(self, named_params):
        """All-reduce everything not reduced yet (one flat bucket), wait, and divide by world size."""
        if self.world <= 1:
            self.pending.clear(); self.done.clear()
            return
        rest = [p.grad for n, p in named_params if p.grad is not None and n not in self.done]
        if rest:
            flat = torch.cat([g.reshape(-1) for g in rest])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(self.world)
            off = 0
            for g in rest:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
        for work, g in self.pending:
            work.wait()
            g.div_(self.world)
        self.pending.clear(); self.done.clear()


def allreduce_counters(values: torch.Tensor) -> torch.Tensor:
    """Sum integer hit/target counters of the evaluator across ranks (a few hundred bytes)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM)
    return values
