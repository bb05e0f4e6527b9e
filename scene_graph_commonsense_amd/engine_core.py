"""Shared definitions of the engine modules: outputs / workspace containers, the loss coefficients on the host, ``Tuning`` (the switches of
the product path) and the deferred-check pool.  ``engine.py`` re-exports every name, so ``engine.TUNING`` / ``engine.tuning`` / ``engine.Workspace``
are these objects.  The orchestration itself: ``engine.py`` (the class), ``engine_weights.py``, ``engine_plan.py``, ``engine_fwd.py``, ``engine_bwd.py``.
"""
from __future__ import annotations

import contextlib
import ctypes
import math
import os
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .synthetic import HeadConfig

ELEM_F16, ELEM_BF16 = 0, 1
XC = 384            # packed input channels (2*128+1 = 257 zero-padded to a multiple of the K tile)


def _c_long(v):
    return ctypes.c_long(int(v))


def conv_k_layout(w: torch.Tensor) -> torch.Tensor:
    """[N, C, 3, 3] conv weight -> [N, 9*C] with K ordered (C/64 chunk, tap, 64 channels): the K order of the
    implicit-GEMM A operand (csrc/gemm_nt.h), chosen so the nine taps of a channel chunk are consecutive."""
    N, C = w.shape[0], w.shape[1]
    return w.reshape(N, C // 64, 64, 9).permute(0, 1, 3, 2).reshape(N, 9 * C)


@dataclass
class PairOutputs:
    relation: torch.Tensor                  # [P, R] log-probs (hier) or raw logits (flat)
    super_relation: Optional[torch.Tensor]  # [P, 3]
    connectivity: torch.Tensor              # [P] raw logit
    hidden: torch.Tensor                    # [P, 512] post-ReLU (post-dropout) fc2 output
    cand_conf: torch.Tensor                 # [P, 3] (hier) or [P, 1]
    cand_pred: torch.Tensor                 # [P, 3] int32 / [P, 1]


class Weights(dict):
    """The 16-bit compute copies of the parameters.  An entry can be DEFERRED: ``defer(key, make)`` registers the function that
    builds it and the first ``w[key]`` of the step runs it.  The two copies of ``fc1.weight`` (97 % of the parameter bytes) are made
    this way, right before the first kernel that reads them: with ``distributed.ShardedSGD(defer_gather=True)`` the all-gather of
    the updated ``fc1.weight`` is still on the wire when the next step starts and ``make`` first waits for it, so the gather
    overlaps everything the forward does before fc1 (flatten, conv1, conv2, conv3: ~12 of 48 ms at the benchmark's size)."""

    def __init__(self):
        super().__init__()
        self.deferred = {}
        # fc1.weight's gradient leaves the backward in GEMM order [4096][window*1024 + channel] instead of the reference's
        # [4096][channel*64 + window]: set for the duration of one ``pair_loop.train_minibatch`` call whose optimizer consumes that order
        # (``optim.FusedSGD`` / ``distributed.ShardedSGD``); kept HERE because every engine of a module (image-group lanes, the
        # augmented view's) shares this object - never set while a caller may look at ``fc1.weight.grad``
        self.fc1_grad_gemm_order = False

    def defer(self, key, make):
        self.deferred[key] = make
        dict.pop(self, key, None)

    def __getitem__(self, key):
        make = self.deferred.pop(key, None)
        if make is not None:
            dict.__setitem__(self, key, make())
        return dict.__getitem__(self, key)


class Workspace:
    """Grow-only cache of device buffers keyed by name (no allocation inside the steady-state step)."""

    def __init__(self, device):
        self.device = device
        self.bufs: Dict[str, torch.Tensor] = {}

    def get(self, name, numel, dtype, zero=False):
        t = self.bufs.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = torch.empty(int(numel), dtype=dtype, device=self.device)
            self._zero(t)              # created zeroed: padded tensors keep their zero halo, kernels write interiors only
            self.bufs[name] = t
        elif zero:
            self._zero(t[:numel])
        return t[:numel]

    @staticmethod
    def _zero(t):
        if t.numel():
            _lib.check(_lib.load().sgc_fill_zero(_lib.ptr(t), _c_long(t.numel() * t.element_size()), _lib.stream_ptr()), "sgc_fill_zero")

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.bufs.values())


# ---------------------------------------------------------------------------------- host-side training helpers
def csr_by(index: np.ndarray, n: int):
    """ptr/list of pair ids grouped by object id (stable, so sums run in pair order)."""
    order = np.argsort(index, kind="stable").astype(np.int32)
    ptr = np.searchsorted(index[order], np.arange(n + 1)).astype(np.int32)
    return ptr, order


def loss_coefficients(cfg: HeadConfig, step: np.ndarray, n_steps: int, directed: np.ndarray, class_weight: np.ndarray,
                      lambda_connectivity: float = 0.1, lambda_not_connected: float = 1.0):
    """Fold the reference's per-step loss bookkeeping into per-pair coefficients.

    Reference: each direction-step t adds  loss_rel_t + lambda_c * loss_conn_t  to running sums that are
    themselves added to ``losses`` after every step (``train_test.py:219-233``), so step t carries the weight
    (T - t).  Inside a step (``train_utils.py:64-94,116-157``): BCE(conn, 1) averaged over the connected pairs
    REPLACES lambda_nc * BCE(conn, 0) averaged over the others whenever a connected pair exists; the relation
    term is mean NLL on the super-category plus, per super-category, a class-weighted mean NLL.
    Returns float32/int32 arrays (tgt, a, b, c, y): loss_i = -a*super[st] - b*rel[t] + c*BCE(conn, y).
    """
    P = step.shape[0]
    w = (n_steps - step).astype(np.float64)
    conn = directed >= 0
    n_conn = np.bincount(step[conn], minlength=n_steps).astype(np.float64)
    n_all = np.bincount(step, minlength=n_steps).astype(np.float64)
    n_nc = n_all - n_conn
    has = n_conn[step] > 0
    c = np.zeros(P)
    c[conn] = w[conn] * lambda_connectivity / n_conn[step[conn]]
    sel = (~conn) & (~has)
    c[sel] = w[sel] * lambda_connectivity * lambda_not_connected / np.maximum(n_nc[step[sel]], 1)
    a = np.zeros(P)
    b = np.zeros(P)
    t = np.where(conn, directed, 0)
    cw = class_weight.astype(np.float64)[t]
    if cfg.hierarchical:
        ng, npos = cfg.num_geometric, cfg.num_possessive
        seg = np.where(t < ng, 0, np.where(t < ng + npos, 1, 2))
        a[conn] = w[conn] / n_conn[step[conn]]
        key = step * 3 + seg
        wsum = np.bincount(key[conn], weights=cw[conn], minlength=3 * n_steps)
        b[conn] = w[conn] * cw[conn] / wsum[key[conn]]
    else:
        wsum = np.bincount(step[conn], weights=cw[conn], minlength=n_steps)
        b[conn] = w[conn] * cw[conn] / wsum[step[conn]]
    return (directed.astype(np.int32), a.astype(np.float32), b.astype(np.float32), c.astype(np.float32),
            conn.astype(np.float32))


class TrainContext:
    pass


@dataclass
class Tuning:
    """The switches of the product path, in ONE place, read once at import.  Defaults are the measured best.
    Environment (four documented variables; tests and tools flip the fields directly, e.g. ``with engine.tuning(shared_fc1=False)``):

      SGC_SHARED_LEVEL         0 per-pair kernels | 1 conv3 over shared windows | 2 + fc1 over the same windows | 3 (default) + the
                               per-object maps shared with the image's background map (second level) and the linear pairs
      SGC_SHARED_MAX_FRACTION  share of pair-specific windows above which a scene goes to the per-pair kernels (default 0.5:
                               profiles/r03_box_sweep.txt - the step time crosses near 0.65 but the workspace reaches 170 GB at 0.5)
      SGC_BWD_STREAMS          0: weight-gradient chain on the caller's stream (single-stream profiles, tools/collect_profiles.sh)
      SGC_TUNING               "field=value,field=value": any field below by name (A/B tools: tools/ab_env.sh SGC_TUNING gemms_apart=1
                               gemms_apart=0), e.g. shared_bwd=0 (per-pair backward under a shared forward), gemms_apart=0 (round 2's order
                               of the two backward chains), shared_linear=0, shared_conv2=0, patch_dgrad=0 / patch_wgrad=0 (the column
                               forms of the conv3 window backward: im2col / col2im)
    Decided and no longer switchable: sparse-MFMA conv3 weight gradient, un-pool fused into the conv3 data gradient, im2col + plain
    GEMM (not the gathered TN block) for the column form of the weight gradient over the listed windows."""
    shared_conv3: bool = True
    shared_fc1: bool = True
    shared_objects: bool = True
    shared_bwd: bool = True
    shared_max_fraction: float = 0.5
    bwd_streams: bool = True
    gemms_apart: bool = True          # two-stream backward: keep the big GEMMs of the two chains from running side by side
    shared_linear: bool = True        # pairs whose regions of influence on the 16-grid are disjoint: X windows combined, not convolved
    shared_conv2: bool = True         # conv2 halves computed on the objects' own regions, the rest copied from the image's background half
    patch_dgrad: bool = True          # conv3 data gradient over the listed windows in patch form (20 rows per window; off: 36 columns + col2im)
    patch_wgrad: bool = True          # conv3 weight gradient over the listed windows from 4 x 4 patches (16 rows per window; off: im2col, 36)
    plan_kernels: bool = True         # row plan of the shared windows by placement kernels (off: torch.sort / searchsorted / gathers, rounds 2-3)
    weight_kernels: bool = True       # 16-bit weight layouts by one gather + cast launch each (off: torch view / permute / flip / cat chains)
    fc1_own_sums: bool = True         # fc1 assembly reads S'_j[R_j] pre-summed per object (off: four corner vectors per pair; same bits)
    sparse_wgrad: bool = True         # conv3 weight gradient over the real pairs' listed windows on the sparse matrix cores (off: dense block)
    fc1_x16: bool = True              # fc1's pair-specific products leave the grouped GEMM as f16 rows (the per-object rows stay f32): -0.8 ms
                                      # per step, hidden error 7.0e-4 -> 7.03e-4 (profiles/r05_fc1_x16_ab.txt).  One more rounding in front of
                                      # fc1's ReLU: like every other one it flips units whose pre-activation is within the forward tolerance of
                                      # zero (held per unit by tests/test_backward_gpu.py::test_backward_matches_reference_fingerprints)
    assemble_by_subject: bool = True  # fc1 assembly walks the pairs sorted by subject (the subject's prefix table stays in the L2s; same bits)
    conv2_bwd_regions: bool = True    # conv2 data gradient only on the cells where an object's gradient can be non-zero (its pseudo-pair's pixel
                                      # rectangle + 1 cell; off: whole 32x32 maps; same bits)
    sparse_dgrad: bool = True         # conv3 data gradient over the real pairs' listed windows on the sparse matrix cores (off: dense patch form)
    gather_wgrad: bool = True         # sparse conv3 weight gradient over the listed windows reads its second operand straight from the forward's f16
                                      # maps through the window list (f16 -> bf16 in registers; off: 3.7 GB patch copy first, +1.0 ms; same bits)
    wgrad_xcd_k: bool = True          # sparse conv3 weight gradient: every XCD owns K ranges (all 36 tiles of a channel half) instead of one M tile
                                      # for every K range: B leaves the fabric once instead of four times (32 split-K slabs instead of 7)
    small_wgrads_main: bool = True    # two-stream backward: the conv2 / conv1 weight gradients run in line on the caller's stream instead of behind the
                                      # side stream's long sparse conv3 weight gradient (the caller's stream waited 2.7 ms for that tail)
    fused_sgd: bool = True            # train_minibatch + optim.FusedSGD: fc1.weight's gradient stays in GEMM order, one pass un-permutes, updates and
                                      # writes the f16 copy (off: transposition + update + transposition; same bits)

    @classmethod
    def from_env(cls):
        lvl = int(os.environ.get("SGC_SHARED_LEVEL", "3"))
        t = cls(shared_conv3=lvl >= 1, shared_fc1=lvl >= 2, shared_objects=lvl >= 3, shared_linear=lvl >= 3, shared_conv2=lvl >= 1,
                shared_max_fraction=float(os.environ.get("SGC_SHARED_MAX_FRACTION", "0.5")),
                bwd_streams=os.environ.get("SGC_BWD_STREAMS", "1") != "0")
        for item in filter(None, os.environ.get("SGC_TUNING", "").split(",")):
            k, _, v = item.partition("=")
            k = k.strip()
            if k not in cls.__dataclass_fields__:
                raise ValueError("SGC_TUNING: unknown field %r (fields: %s)" % (k, ", ".join(cls.__dataclass_fields__)))
            setattr(t, k, float(v) if k == "shared_max_fraction" else v.strip() not in ("0", "false", "False", ""))
        return t


TUNING = Tuning.from_env()


@contextlib.contextmanager
def tuning(**overrides):
    """Temporarily override fields of ``TUNING`` (tests, A/B tools)."""
    old = {k: getattr(TUNING, k) for k in overrides}
    for k, v in overrides.items():
        setattr(TUNING, k, v)
    try:
        yield TUNING
    finally:
        for k, v in old.items():
            setattr(TUNING, k, v)


def shared_fc1_enabled() -> bool:
    """fc1 as a grouped window-major GEMM (``TUNING.shared_fc1``; off: one [pairs, 65536] GEMM over assembled rows)."""
    return TUNING.shared_fc1


def shared_objects_enabled() -> bool:
    """Second level (``TUNING.shared_objects``): a pseudo-pair (object, background) computes only the windows of the object's
    rectangle, the rest comes from the image's all-background map (off: every pseudo-pair is a full conv3 map)."""
    return TUNING.shared_objects


def shared_conv3_enabled(hint=None, n_pairs=0) -> bool:
    """conv3 over shared windows (``TUNING.shared_conv3``).  ``hint`` (the host's count of pair-specific windows,
    ``DeviceScene.shared_windows``): when more than ``TUNING.shared_max_fraction`` of all windows are pair-specific (most boxes
    cover most of the image) the per-pair kernels are used - the column buffers of the shared backward grow with that count
    (9.2 KB per window pixel, twice); measured: ``bench.py`` sensitivity sweep / profiles/r03_box_sweep.txt."""
    if not TUNING.shared_conv3:
        return False
    n = hint.get("windows") if isinstance(hint, dict) else hint
    if n is not None and n_pairs > 0 and n > TUNING.shared_max_fraction * 64 * n_pairs:
        return False
    return True


class _CheckRing:
    """Pinned int32 words for the deferred consistency checks of all engines of this process (``RelHeadEngine._post_check``): a pool of
    one-word views of pinned blocks; a word goes back to the pool when ``verify_checks`` has looked at it, the pool grows by a block
    when it is empty (engines that are dropped with checks pending simply never return theirs)."""
    BLOCK = 64
    free = []

    @classmethod
    def take(cls):
        if not cls.free:
            block = torch.zeros(cls.BLOCK, dtype=torch.int32).pin_memory()
            cls.free = [block[i:i + 1] for i in range(cls.BLOCK)]
        return cls.free.pop()


__all__ = [
    "Dict",
    "ELEM_BF16",
    "ELEM_F16",
    "HeadConfig",
    "Optional",
    "PairOutputs",
    "TUNING",
    "TrainContext",
    "Tuning",
    "Weights",
    "Workspace",
    "XC",
    "_CheckRing",
    "_c_long",
    "_lib",
    "contextlib",
    "conv_k_layout",
    "csr_by",
    "ctypes",
    "dataclass",
    "loss_coefficients",
    "math",
    "np",
    "os",
    "shared_conv3_enabled",
    "shared_fc1_enabled",
    "shared_objects_enabled",
    "torch",
    "tuning",
]
