"""Host orchestration of the fused pairwise relation path on one MI355X.

All compute is in the HIP kernels of ``csrc/`` reached through the C-ABI of ``include/sgc_relhead.h``;
torch supplies device memory, the current HIP stream and (elsewhere) torch.distributed.  The restructuring
identities are SURVEY §7.1:

  conv1 is 1x1 and the bbox mask is per pixel   -> conv1 runs once per image and role (``sgc_conv1_tanh``)
  conv2 is linear over a channel concat          -> U_i + V_j from per-object convs (``sgc_conv2_object``)
  everything after that ReLU is per pair         -> expansion, conv3, fc1, fc2, head over all pairs at once
  one-hot label concat                           -> per-object 512-vectors gathered in fc2's epilogue
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


def make_engine(cfg: HeadConfig, device="cuda:0") -> "RelHeadEngine":
    """The engine for ``cfg``: the tiled MFMA kernels for the reference's sizes (hidden_dim 128, feature_size 32 - every shipped
    configuration), the generic f32 trunk (``engine_generic.GenericTrunkEngine``) for any other ``input_dim`` / ``feature_size`` the
    reference's constructor accepts (``model.py:110-111``)."""
    if cfg.hidden_dim == 128 and cfg.feature_size == 32:
        return RelHeadEngine(cfg, device)
    from .engine_generic import GenericTrunkEngine
    # per pair: conv2 + conv3 multiply-adds of the reference graph at this size (model.py:141-146)
    macs = 9.0 * (2 * cfg.hidden_dim) * (4 * cfg.hidden_dim) * cfg.feature_size ** 2 + 9.0 * (4 * cfg.hidden_dim) * (8 * cfg.hidden_dim) * (cfg.feature_size // 2) ** 2
    if macs > 5e7 and not _GENERIC_WARNED:
        import warnings
        _GENERIC_WARNED.append(True)
        warnings.warn("relation head built with input_dim=%d, feature_size=%d: the tiled gfx950 kernels exist for 128 / 32 only; this size "
                      "runs on the generic f32 trunk (one thread per output element, %.2g MACs per pair - correct, but orders of magnitude "
                      "off the MFMA roofline; csrc/kernels_generic.hip)" % (cfg.hidden_dim, cfg.feature_size, macs), RuntimeWarning, stacklevel=2)
    return GenericTrunkEngine(cfg, device)


_GENERIC_WARNED = []


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


class RelHeadEngine:
    """Forward and backward of the relation head over explicit pair lists (one instance = one GPU, one workspace)."""

    def _check_sizes(self, cfg: HeadConfig):
        if cfg.hidden_dim != 128 or cfg.feature_size != 32:
            raise NotImplementedError("the tiled gfx950 kernels are specialised to hidden_dim=128, feature_size=32; other sizes run on "
                                      "engine_generic.GenericTrunkEngine (engine.make_engine picks it)")
        if cfg.num_relations + 4 > 64:
            raise NotImplementedError("head kernel holds one output row per wavefront lane (<= 60 relations)")

    def __init__(self, cfg: HeadConfig, device="cuda:0"):
        self._check_sizes(cfg)
        self.cfg = cfg
        self.device = torch.device(device)
        self.lib = _lib.load()
        self.ws = Workspace(self.device)          # buffers a training context keeps until its backward has run
        self.scratch = self.ws                    # transient buffers; a child engine shares its parent's (see ``child``)
        self.w: Dict[str, torch.Tensor] = Weights()
        self.T = (1.0, 1.0, 1.0)
        self.timers = None          # optional {name: [(start_event, end_event), ...]} filled by bench.py
        self._checks = []           # deferred device-side consistency checks: (event, pinned flag, message), see ``_post_check``
        self._w1p_fresh = None      # (data_ptr, version) of the fc1.weight whose f16 copy the fused optimizer step has already written

    # ------------------------------------------------------------------ deferred consistency checks
    def _post_check(self, bad: torch.Tensor, message: str):
        """``bad`` (device bool/int scalar, non-zero = inconsistent) is copied to pinned memory behind the work enqueued so far and
        looked at LATER (``verify_checks``: at the next forward, at the end of an epoch / an evaluation pass, or explicitly) - a
        host-side ``int(tensor)`` here would stall the launch queue of every training step for a condition that never holds in the
        drivers' own use.  The pinned words come from one process-wide ring (``_CheckRing``): no allocation per plan."""
        word = _CheckRing.take()
        word.copy_(bad.reshape(1).to(torch.int32), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._checks.append((ev, word, message))

    def verify_checks(self, block: bool = False):
        """Raise if a posted check failed (``block``: wait for the pending ones first)."""
        keep, failed = [], None
        for ev, word, message in self._checks:
            if block:
                ev.synchronize()
            if not ev.query():
                keep.append((ev, word, message))
                continue
            if int(word[0]) != 0 and failed is None:
                failed = message
            _CheckRing.free.append(word)
        if failed is not None:
            keep = []                 # (their words are not reused: copies into them may still be in flight)
        self._checks[:] = keep
        if failed is not None:
            raise RuntimeError(failed)

    @property
    def fc1_grad_gemm_order(self) -> bool:
        return bool(getattr(self.w, "fc1_grad_gemm_order", False))

    @fc1_grad_gemm_order.setter
    def fc1_grad_gemm_order(self, v: bool):
        self.w.fc1_grad_gemm_order = bool(v)

    def child(self) -> "RelHeadEngine":
        """An engine that shares this one's weights and transient scratch but owns the buffers of its training context: the
        per-step ``forward()`` of the drop-in modules keeps many contexts alive until ``losses.backward()``
        (``train_test.py:189-276``: one classifier call per direction-step, one backward per minibatch)."""
        c = type(self).__new__(type(self))
        c.cfg, c.device, c.lib, c.w, c.T, c.timers = self.cfg, self.device, self.lib, self.w, self.T, None
        c._checks = self._checks
        c.head_rows = getattr(self, "head_rows", None)
        c._w1p_fresh = None
        c.ws, c.scratch = Workspace(self.device), self.scratch
        c._side_stream = getattr(self, "_side_stream", None)
        return c

    def _timed(self, name, fn):
        """Run one launch, bracketing it with HIP events on the current stream when bench timers are on."""
        if self.timers is None:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self.timers.setdefault(name, []).append((a, b))
        return r

    # ------------------------------------------------------------------ weights
    def load_weights(self, sd: Dict[str, torch.Tensor], fc1_sync=None):
        """Build the 16-bit compute copies (layouts of csrc/kernels_fwd.hip) from the f32 master weights.  ``fc1_sync``: called
        before ``fc1.weight`` is read (``Weights``: that copy is made at its first use in the step)."""
        cfg, dev = self.cfg, self.device
        g = lambda k: sd[k].detach().to(dev, torch.float32)
        w = self.w
        self._load_trunk_weights(sd, g, fc1_sync)
        self._load_head_weights(sd, g)

    def _permute_cast(self, src, dst, kind, dims, sstr, dstr=None, src_off=0, dst_off=0):
        """dst[dst_off + i.dstr] = cast(src[src_off + i.sstr]) over ``dims`` (``sgc_permute_cast``; dst contiguous when ``dstr`` is None)."""
        n = len(dims)
        if dstr is None:
            dstr, acc = [0] * n, 1
            for k in range(n - 1, -1, -1):
                dstr[k], acc = acc, acc * dims[k]
        _lib.check(self.lib.sgc_permute_cast(_lib.ptr(src), _lib.ptr(dst), kind, n, (ctypes.c_int * n)(*dims), (ctypes.c_long * n)(*sstr),
                                             (ctypes.c_long * n)(*dstr), _c_long(src_off), _c_long(dst_off), self._st()), "sgc_permute_cast")

    def _load_trunk_weights(self, sd, g, fc1_sync):
        w = self.w
        if TUNING.weight_kernels:
            # every layout below = one gather + cast launch from the f32 master (the torch forms in the else branch are the definition)
            w1r = self.ws.get("w1r", 2 * 128 * XC, torch.float16)                  # created zeroed; the channel padding stays zero
            c2, c3 = g("conv2_1.weight").contiguous(), g("conv3_1.weight").contiguous()
            w2r = self.ws.get("w2r", 2 * 512 * 1152, torch.float16)
            for r, name in enumerate(("conv1_1.weight", "conv1_2.weight")):
                self._permute_cast(g(name).contiguous(), w1r, 0, [128, 257], [257, 1], [XC, 1], dst_off=r * 128 * XC)
                self._permute_cast(c2, w2r, 0, [512, 2, 9, 64], [2304, 576, 1, 9], src_off=r * 128 * 9, dst_off=r * 512 * 1152)
            w3r = self.ws.get("w3r", 1024 * 4608, torch.float16)
            self._permute_cast(c3, w3r, 0, [1024, 8, 9, 64], [4608, 576, 1, 9])
            w["w1r"], w["w2r"], w["w3r"] = w1r.view(2, 128, XC), w2r.view(2, 512, 1152), w3r.view(1024, 4608)
            w["b1"] = torch.stack([g("conv1_1.bias"), g("conv1_2.bias")]).contiguous()
            w["cst"] = torch.tanh(w["b1"]).half().contiguous()                  # tanh(conv1(0)) outside the box
            w["b2"] = g("conv2_1.bias").contiguous()
            w["b3"] = g("conv3_1.bias").contiguous()
        else:
            self._load_trunk_weights_torch(g)

        fc1_param = sd["fc1.weight"]

        def make_w1p():
            if fc1_sync is not None:
                fc1_sync()
            fresh = getattr(self, "_w1p_fresh", None)
            if fresh is not None and fresh == (fc1_param.data_ptr(), fc1_param._version) and "w1p" in self.ws.bufs:
                return self.ws.bufs["w1p"][:fc1_param.numel()]       # written by the fused optimizer step (sgc_sgd_fc1_fused) for this version
            with torch.no_grad():
                return self._transpose_cast(g("fc1.weight").contiguous(), "w1p", torch.float16, 0, 4096, 16, 65536, 4096, 64, 65536, 64, 1024)
        w.defer("w1p", make_w1p)
        w["bf1"] = g("fc1.bias").contiguous()

    def _load_trunk_weights_torch(self, g):
        w = self.w
        w1r = self.ws.get("w1r", 2 * 128 * XC, torch.float16).view(2, 128, XC)     # created zeroed; the channel padding stays zero
        w1r[0, :, :257] = g("conv1_1.weight").view(128, 257).half()
        w1r[1, :, :257] = g("conv1_2.weight").view(128, 257).half()
        w["w1r"] = w1r
        w["b1"] = torch.stack([g("conv1_1.bias"), g("conv1_2.bias")]).contiguous()
        w["cst"] = torch.tanh(w["b1"]).half().contiguous()                  # tanh(conv1(0)) outside the box
        c2 = g("conv2_1.weight")
        w["w2r"] = torch.stack([conv_k_layout(c2[:, r * 128:(r + 1) * 128]) for r in (0, 1)]).half().contiguous()
        w["b2"] = g("conv2_1.bias").contiguous()
        w["w3r"] = conv_k_layout(g("conv3_1.weight")).half().contiguous()
        w["b3"] = g("conv3_1.bias").contiguous()

    def _load_head_weights(self, sd, g):
        """fc2 and the head: independent of hidden_dim / feature_size (fc1 always ends in 4096 features)."""
        cfg, w = self.cfg, self.w
        fc2 = g("fc2.weight")
        w["fc2_full"] = fc2
        if TUNING.weight_kernels:
            w2m = self.ws.get("w2m", 512 * 4096, torch.float16)
            self._permute_cast(fc2.contiguous(), w2m, 0, [512, 4096], [int(fc2.shape[1]), 1])
            w["w2m"] = w2m.view(512, 4096)
        else:
            w["w2m"] = fc2[:, :4096].half().contiguous()
        w["bf2"] = g("fc2.bias").contiguous()
        R = cfg.num_relations
        if cfg.hierarchical:
            rows = [g("fc3_1.weight"), g("fc3_2.weight"), g("fc3_3.weight"), g("fc5.weight"), g("fc4.weight")]
            bias = [g("fc3_1.bias"), g("fc3_2.bias"), g("fc3_3.bias"), g("fc5.bias"), g("fc4.bias")]
        else:
            rows = [g("fc3.weight"), g("fc4.weight")]
            bias = [g("fc3.bias"), g("fc4.bias")]
        Wc = self.ws.get("head_rows", 64 * 512, torch.float32).view(64, 512)         # created zeroed; rows beyond the head stay zero
        bc = self.ws.get("head_bias", 64, torch.float32)
        rc = torch.cat(rows)
        Wc[:rc.shape[0]] = rc
        bc[:rc.shape[0]] = torch.cat(bias)
        w["head_wt"] = Wc.t().contiguous()
        w["head_b"] = bc
        w["head_w"] = Wc                             # row-major copy for the head backward
        self.head_rows = rc.shape[0]

    # ------------------------------------------------------------------ helpers
    def _transpose_cast(self, src, name, dtype, kind, na, nb, sa_s, sb_s, ss_i, sa_d, sb_d, ds_j):
        dst = self.ws.get(name, src.numel(), dtype)
        _lib.check(self.lib.sgc_transpose_cast(_lib.ptr(src), _lib.ptr(dst), kind, na, nb, _c_long(sa_s), _c_long(sb_s),
                                               _c_long(ss_i), _c_long(sa_d), _c_long(sb_d), _c_long(ds_j), self._st()),
                   "sgc_transpose_cast")
        return dst

    def _st(self):
        return _lib.stream_ptr()

    def label_vectors(self, cats: torch.Tensor, super_mh: Optional[torch.Tensor]):
        """Per-object 512-vectors replacing the one-hot/multi-hot concat of ``model.py:152-168`` (``sgc_label_vectors``)."""
        cfg, fc2 = self.cfg, self.w["fc2_full"]
        n_obj = int(cats.shape[0])
        mh = super_mh if (super_mh is not None and cfg.dataset == "vg") else None
        lsub = torch.empty(n_obj, 512, dtype=torch.float32, device=self.device)
        lobj = torch.empty(n_obj, 512, dtype=torch.float32, device=self.device)
        cats = cats if cats.dtype == torch.int64 else cats.long()
        _lib.check(self.lib.sgc_label_vectors(_lib.ptr(fc2), int(fc2.shape[1]), 4096, _lib.ptr(cats), _lib.ptr(mh), n_obj, cfg.num_classes,
                                              cfg.num_super_classes if mh is not None else 0, _lib.ptr(lsub), _lib.ptr(lobj), self._st()),
                   "sgc_label_vectors")
        return lsub, lobj

    def loss_coefficients_device(self, step_ptr: torch.Tensor, n_steps: int, directed: torch.Tensor, class_weight: torch.Tensor,
                                 lambda_connectivity: float = 0.1, lambda_not_connected: float = 1.0):
        """Device form of ``loss_coefficients`` (``sgc_loss_coefficients``: one thread per direction-step, same double arithmetic
        in the same order).  ``directed`` [P] int32, ``class_weight`` [R] f32, both on the device."""
        cfg = self.cfg
        P = int(directed.shape[0])
        tgt = torch.empty(P, dtype=torch.int32, device=self.device)
        co = torch.empty(4, P, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.sgc_loss_coefficients(_lib.ptr(step_ptr), int(n_steps), _lib.ptr(directed), _lib.ptr(class_weight),
                                                  cfg.num_geometric if cfg.hierarchical else cfg.num_relations, cfg.num_possessive,
                                                  int(cfg.hierarchical), ctypes.c_double(lambda_connectivity),
                                                  ctypes.c_double(lambda_not_connected), _lib.ptr(tgt), _lib.ptr(co[0]), _lib.ptr(co[1]),
                                                  _lib.ptr(co[2]), _lib.ptr(co[3]), self._st()), "sgc_loss_coefficients")
        return (tgt, co[0], co[1], co[2], co[3])

    def connectivity_stats(self, conn: torch.Tensor, directed: torch.Tensor, raw: torch.Tensor, included: Optional[torch.Tensor] = None):
        """[5] int64 device counters (not connected, connected, predicted connected, precision numerator, recall numerator) of
        ``train_utils.py:66-87,176-184`` summed over the pairs (``included`` u8: only the steps the overlap filter kept)."""
        out = torch.empty(5, dtype=torch.int64, device=self.device)
        _lib.check(self.lib.sgc_connectivity_stats(_lib.ptr(conn), _lib.ptr(directed), _lib.ptr(raw), _lib.ptr(included),
                                                   int(conn.shape[0]), _lib.ptr(out), self._st()), "sgc_connectivity_stats")
        return out

    # ------------------------------------------------------------------ stages
    def image_maps(self, f0: torch.Tensor, f1: Optional[torch.Tensor], roles=(0, 1), tag="img"):
        """conv1 + tanh per image and role: returns {role: a_img [n_img*1024,128] f16}."""
        lib, ws = self.lib, self.ws
        n_img = f0.shape[0]
        C0 = f0.shape[1]
        C1 = 0 if f1 is None else f1.shape[1]
        x = ws.get("x_" + tag, n_img * 1024 * XC, torch.float16)
        _lib.check(lib.sgc_pack_image_nhwc(_lib.ptr(f0), C0, _lib.ptr(f1), C1, _lib.ptr(x), n_img, 1024, XC, self._st()),
                   "sgc_pack_image_nhwc")
        out = {}
        for r in roles:
            a = ws.get("a_img_%s_%d" % (tag, r), n_img * 1024 * 128, torch.float16)
            _lib.check(lib.sgc_conv1_tanh(_lib.ptr(x), _lib.ptr(self.w["w1r"][r]), _lib.ptr(self.w["b1"][r]), _lib.ptr(a),
                                          n_img * 1024, XC, self._st()), "sgc_conv1_tanh")
            out[r] = a
        self._x = x
        return out

    def object_halves(self, a_img, obj_img: torch.Tensor, bbox: torch.Tensor, roles=(0, 1), with_bg=False, regions=None):
        """Per-object masked maps and conv2 halves U (role 0) / V (role 1, carries the bias).
        ``with_bg``: one object with an EMPTY box per image is appended (index n_obj + image) - the constant map tanh(b1) every
        masked map equals outside its box; its halves are the background of ``conv3_shared``.  (One per image rather than one in
        all: the backward sums the background's gradient per image, so a step over B images stays the sum of B one-image steps.)
        ``regions`` (host count of the 2x2-pixel windows of all objects' D16 rectangles, ``DeviceScene.conv2_windows``; needs
        ``with_bg``): conv2 runs on those windows only - outside them an object's half IS its image's background half (the map is
        the constant tanh(b1) outside the box), copied row by row: same bits, 17 % of the rows on the benchmark's boxes."""
        lib, ws = self.lib, self.ws
        n_real = int(obj_img.shape[0])
        n_img = 0
        if with_bg:
            n_img = int(a_img[roles[0]].numel()) // (1024 * 128)
            obj_img = torch.cat([obj_img, torch.arange(n_img, dtype=obj_img.dtype, device=obj_img.device)])
            bbox = torch.cat([bbox, bbox.new_zeros(n_img, 4)])
        n_obj = obj_img.shape[0]
        by_region = bool(with_bg and regions and TUNING.shared_conv2 and n_real > 0)
        if by_region:
            cnt = torch.empty(n_real, dtype=torch.int32, device=self.device)
            _lib.check(lib.sgc_conv2_regions_count(_lib.ptr(bbox), n_real, _lib.ptr(cnt), self._st()), "sgc_conv2_regions_count")
            incl = torch.cumsum(cnt, 0, dtype=torch.int32)
            rlist = self.scratch.get("conv2_regions", int(regions) + 64, torch.int32)
            _lib.check(lib.sgc_conv2_regions_fill(_lib.ptr(bbox), n_real, _lib.ptr(incl), _lib.ptr(rlist), self._st()), "sgc_conv2_regions_fill")
            rn = incl[n_real - 1:]
        res = {}
        for r in roles:
            a_pad = ws.get("a_pad_%d" % r, n_obj * 34 * 34 * 128, torch.float16)
            _lib.check(lib.sgc_object_masked_maps(_lib.ptr(a_img[r]), _lib.ptr(obj_img), _lib.ptr(bbox),
                                                  _lib.ptr(self.w["cst"][r]), _lib.ptr(a_pad), n_obj, 32, 128,
                                                  self._st()), "sgc_object_masked_maps")
            uv = self.scratch.get("uv_%d" % r, n_obj * 1024 * 512, torch.float16)
            bias = _lib.ptr(self.w["b2"]) if r == 1 else None
            if by_region:
                def run(a_pad=a_pad, uv=uv, bias=bias, r=r):
                    _lib.check(lib.sgc_conv2_object(_lib.ptr(a_pad[n_real * 34 * 34 * 128:]), _lib.ptr(self.w["w2r"][r]), bias,
                                                    _lib.ptr(uv[n_real * 1024 * 512:]), n_img, self._st()), "sgc_conv2_object")
                    _lib.check(lib.sgc_conv2_object_regions(_lib.ptr(a_pad), _lib.ptr(self.w["w2r"][r]), bias, _lib.ptr(rlist), _lib.ptr(rn),
                                                            int(regions), _lib.ptr(uv), self._st()), "sgc_conv2_object_regions")
                    _lib.check(lib.sgc_conv2_fill_background(_lib.ptr(bbox), _lib.ptr(obj_img), n_real, _lib.ptr(uv), self._st()),
                               "sgc_conv2_fill_background")
                self._timed("conv2_fwd", run)
            else:
                self._timed("conv2_fwd", lambda: _lib.check(lib.sgc_conv2_object(_lib.ptr(a_pad), _lib.ptr(self.w["w2r"][r]), bias, _lib.ptr(uv),
                                                                                n_obj, self._st()), "sgc_conv2_object"))
            res[r] = uv
        return res

    def expand(self, U, V, sub_idx, obj_idx, P, z, z_bf=None, amz=None, dense=None, pixrect=None):
        """Pair expansion: dense LDS-staged kernel when the pair list is "all ordered pairs of every image"
        (dense = (img_ptr, pid, max_n)), generic pair-list kernel otherwise.  ``pixrect`` ([P] packed rectangles from
        ``shared_plan``): only the pixels conv3 over shared windows reads are written (dense kernel only)."""
        lib = self.lib
        if dense is not None and 0 < dense[2] <= 150:
            img_ptr, pid, max_n = dense
            self._timed("expand_dense", lambda: _lib.check(lib.sgc_pair_expand_dense_windows(
                _lib.ptr(U), _lib.ptr(V), _lib.ptr(img_ptr), _lib.ptr(pid), int(pid.shape[1]), int(img_ptr.shape[0]) - 1, max_n,
                _lib.ptr(z), _lib.ptr(z_bf), _lib.ptr(amz), _lib.ptr(pixrect), self._st()), "sgc_pair_expand_dense_windows"))
        elif z_bf is None and amz is None:
            self._timed("expand", lambda: _lib.check(lib.sgc_pair_expand(_lib.ptr(U), _lib.ptr(V), _lib.ptr(sub_idx), _lib.ptr(obj_idx),
                                                                       _lib.ptr(z), P, ELEM_F16, self._st()), "sgc_pair_expand"))
        else:
            self._timed("expand_train", lambda: _lib.check(lib.sgc_pair_expand_train(
                _lib.ptr(U), _lib.ptr(V), _lib.ptr(sub_idx), _lib.ptr(obj_idx), _lib.ptr(z), _lib.ptr(z_bf), _lib.ptr(amz), P,
                self._st()), "sgc_pair_expand_train"))

    FULL_PIXRECT = (16 << 5) | (16 << 15)          # packed pixel rectangle covering the whole 16x16 map

    def shared_plan(self, bbox, sub_idx, obj_idx, P, bound=None, keep=False, n_obj=0, n_img=0, objects=False, obj_img=None):
        """The window list of a pair list (``csrc/kernels_shared.hip``).  Pair index space: [P real pairs][2*n_obj pseudo-pairs
        (o, bg), (bg, o)][n_img all-background maps].  ``gather`` = pair*64 + window of every listed window (the X windows of the
        real pairs and - second level, ``objects`` - the windows R_o of the pseudo-pairs), ``incl`` inclusive prefix counts over
        the pair index space, ``pixrect`` the packed pixel rectangle in which a pair's z / routing codes / dz exist (whole map for
        pseudo-pairs without the second level and for the background maps).  ``bound`` = what the host knows
        (``model._shared_hint``): then nothing is read back; with ``keep`` (training: the backward's GEMMs need exact sizes) one
        sync reads the counts otherwise.  ``keep``: the lists live in buffers this engine owns."""
        lib = self.lib
        hint = bound if isinstance(bound, dict) else ({"windows": bound} if bound is not None else {})
        own = self.ws if keep else self.scratch
        n2 = 2 * n_obj
        Pt = P + n2 + n_img
        if (objects and TUNING.shared_linear and obj_img is not None and hint.get("linear_windows") and hint.get("windows") is not None
                and hint.get("object_windows") is not None and P > 0):
            return self._shared_plan_linear(bbox, sub_idx, obj_idx, P, hint, own, n_obj, n_img, obj_img)
        cnt = self.scratch.get("xw_count", Pt, torch.int32)
        pixrect = own.get("xw_pixrect", Pt, torch.int32)
        _lib.check(lib.sgc_shared_windows_count(_lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), P, _lib.ptr(cnt), _lib.ptr(pixrect),
                                                self._st()), "sgc_shared_windows_count")
        if Pt > P:
            cnt[P:].zero_()
            pixrect[P:].fill_(self.FULL_PIXRECT)
            if objects:
                _lib.check(lib.sgc_shared_objects_count(_lib.ptr(bbox), n_obj, _lib.ptr(cnt[P:]), _lib.ptr(pixrect[P:]), self._st()),
                           "sgc_shared_objects_count")
        incl = torch.cumsum(cnt, 0, dtype=torch.int32)
        gather = own.get("xw_gather", (P + (n2 if objects else 0)) * 64, torch.int32)
        _lib.check(lib.sgc_shared_windows_fill(_lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), P, _lib.ptr(incl), _lib.ptr(gather),
                                               self._st()), "sgc_shared_windows_fill")
        if objects:
            _lib.check(lib.sgc_shared_objects_fill(_lib.ptr(bbox), n_obj, P, _lib.ptr(incl), _lib.ptr(gather), self._st()), "sgc_shared_objects_fill")
        e_real, e_obj = hint.get("windows"), (hint.get("object_windows") if objects else 0)
        if (e_real is None or e_obj is None) and keep:
            e_real = int(incl[P - 1]) if P else 0
            e_obj = int(incl[Pt - 1]) - e_real
        exact = e_real is not None and e_obj is not None
        total = (e_real + e_obj) if exact else (P + (n2 if objects else 0)) * 64
        if not exact and e_real is not None:
            total = min(total, int(e_real) + n2 * 64)
        self._xw = (gather, incl[:P] if P else incl)
        self._xw_total = incl[Pt - 1:]
        self._xw_linear = None
        return dict(gather=gather, incl=incl, n_total=incl[Pt - 1:], pixrect=pixrect, bound=total, entries=total if exact else None,
                    entries_real=e_real if exact else None, window_entries=hint.get("per_window"), objects=objects, P=P, n_obj=n_obj,
                    n_img=n_img)

    def _shared_plan_linear(self, bbox, sub_idx, obj_idx, P, hint, own, n_obj, n_img, obj_img):
        """``shared_plan`` with the LINEAR pairs split off (csrc/kernels_shared.hip, sixth identity; full scenes only: the host knows
        every count).  Three lists over the same pair index space: ALL X windows (what fc1 multiplies: ``gather_all`` / ``incl_all``,
        the window-major destinations are per entry of this list), the CONV list (pairs that convolve their own windows + the
        per-object entries: under the plan's usual names ``gather`` / ``incl`` / ``n_total`` / ``pixrect``, so the conv3 kernels of
        both directions run on it unchanged) and the LINEAR list (``lin``: windows combined from per-object pre-activations)."""
        lib, dev = self.lib, self.device
        n2 = 2 * n_obj
        Pt = P + n2 + n_img
        e_all, e_obj, e_lin = int(hint["windows"]), int(hint["object_windows"]), int(hint["linear_windows"])
        cnt = self.scratch.get("xw_count3", 3 * Pt, torch.int32).view(3, Pt)
        pixrect = own.get("xw_pixrect", Pt, torch.int32)
        _lib.check(lib.sgc_shared_windows_count3(_lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), P, _lib.ptr(cnt[0]), _lib.ptr(cnt[1]),
                                                 _lib.ptr(cnt[2]), _lib.ptr(pixrect), self._st()), "sgc_shared_windows_count3")
        cnt[:, P:].zero_()
        pixrect[P:].fill_(self.FULL_PIXRECT)
        _lib.check(lib.sgc_shared_objects_count(_lib.ptr(bbox), n_obj, _lib.ptr(cnt[0, P:]), _lib.ptr(pixrect[P:]), self._st()),
                   "sgc_shared_objects_count")
        cnt[1, P:P + n2] = cnt[0, P:P + n2]
        if TUNING.plan_kernels:
            incl = own.get("xw_incl3", 3 * Pt, torch.int32).view(3, Pt)         # [3][Pt]: all / conv / linear
            _lib.check(lib.sgc_scan_rows(_lib.ptr(cnt), _lib.ptr(incl), 3, Pt, self._st()), "sgc_scan_rows")
            incl_all, incl_c, incl_l = incl[0], incl[1], incl[2]
        else:
            incl = torch.cumsum(cnt, 1, dtype=torch.int32)
            incl_all, incl_c, incl_l = incl[0].contiguous(), incl[1].contiguous(), incl[2].contiguous()
        # the host's counts size every buffer and list below; they are TRUSTED (no read-back) but checked: the device's own counts of
        # the boxes / pair lists actually passed must equal them (a scene whose boxes were edited after ``flatten_scene`` would
        # otherwise misplace rows silently).  Looked at by ``verify_checks`` at the next forward.
        self._post_check((incl_all[P - 1] != e_all) | (incl_l[P - 1] != e_lin) | ((incl_all[Pt - 1] - incl_all[P - 1]) != e_obj),
                         "shared-window plan: the scene's host-side window counts (windows=%d, linear_windows=%d, object_windows=%d) do not "
                         "match the boxes / pair lists on the device - was the scene modified after flatten_scene()?" % (e_all, e_lin, e_obj))
        e_c = e_all - e_lin
        gather_all = own.get("xw_gather_all", e_all + e_obj + 64, torch.int32)
        gather_c = own.get("xw_gather", e_c + e_obj + 64, torch.int32)
        gather_l = own.get("xw_gather_lin", e_lin + 64, torch.int32)
        _lib.check(lib.sgc_shared_windows_fill(_lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), P, _lib.ptr(incl_all), _lib.ptr(gather_all),
                                               self._st()), "sgc_shared_windows_fill")
        _lib.check(lib.sgc_shared_windows_fill_class(_lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), P, _lib.ptr(incl_c), _lib.ptr(gather_c),
                                                     1, self._st()), "sgc_shared_windows_fill_class")
        _lib.check(lib.sgc_shared_windows_fill_class(_lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), P, _lib.ptr(incl_l), _lib.ptr(gather_l),
                                                     2, self._st()), "sgc_shared_windows_fill_class")
        for inc, ga in ((incl_all, gather_all), (incl_c, gather_c)):
            _lib.check(lib.sgc_shared_objects_fill(_lib.ptr(bbox), n_obj, P, _lib.ptr(inc), _lib.ptr(ga), self._st()), "sgc_shared_objects_fill")
        # the linear windows ordered by (image, window) for the background side of the backward (stable: sums in list order):
        # one placement kernel instead of sort + searchsorted + gathers (sgc_bucket_place, bit-identical: tests/test_scene_gpu.py)
        # (two-level placement, sgc_bucket_place_seg: cost linear in the list at any minibatch size; the torch sort below is only the
        # reference the kernel is tested against, ``TUNING.plan_kernels`` off)
        if TUNING.plan_kernels:
            order = own.get("xw_lin_order", e_lin + 64, torch.int32)
            seg = own.get("xw_lin_seg", 64 * n_img + 1, torch.int32)
            self._bucket_place(gather_l, e_lin, sub_idx, obj_img, 1, 64 * n_img, None, order, seg, 1)
        else:
            code_l = gather_l[:e_lin].long()
            keys = obj_img.long()[sub_idx.long()[code_l >> 6]] * 64 + (code_l & 63)
            skeys, order = torch.sort(keys, stable=True)
            seg = torch.searchsorted(skeys, torch.arange(64 * n_img + 1, device=dev)).to(torch.int32).contiguous()
            order = order.to(torch.int32).contiguous()
        lin = dict(gather=gather_l, n=incl_l[P - 1:P].contiguous(), max=e_lin, order=order, seg=seg,
                   drow=own.get("xw_lin_drow", e_lin + 64, torch.int32))
        self._xw = (gather_all, incl_all[:P])
        self._xw_total = incl_c[Pt - 1:]
        self._xw_linear = (e_lin, e_obj)          # bench accounting: windows combined instead of convolved, per-object entries
        return dict(gather=gather_c, incl=incl_c, n_total=incl_c[Pt - 1:], pixrect=pixrect, bound=e_c + e_obj, entries=e_c + e_obj,
                    entries_real=e_c, window_entries=hint.get("per_window"), objects=True, P=P, n_obj=n_obj, n_img=n_img,
                    gather_all=gather_all, incl_all=incl_all, entries_all=e_all + e_obj, entries_real_all=e_all, lin=lin)

    def window_major_rows(self, plan, P, n2):
        """Window-major row space of the shared fc1 (``csrc/kernels_shared.hip``): device group offsets, tile -> group table and the
        row ``dest[e]`` of every listed window (X entries behind the per-object rows of their group; a pseudo-pair's own windows
        ARE per-object rows).  Per-window entry counts come from the host when it knows them (full scenes,
        ``DeviceScene.window_entries``); a pair subset costs one read-back."""
        from .pairs import window_major_layout
        dev = self.device
        split = "gather_all" in plan                  # linear pairs split off: the rows are those of the list of ALL X windows
        gather = plan["gather_all"] if split else plan["gather"]
        counts = plan.get("window_entries")
        if counts is None:
            E = int((plan["incl_all"] if split else plan["incl"])[P - 1]) if P else 0
            counts = torch.bincount((gather[:E] & 63).long(), minlength=64).cpu().numpy()
        E = int(np.asarray(counts).sum())
        goff, tile_group = window_major_layout(counts, n2)
        # ONE host-to-device copy for the four small tables: group offsets, first X row of every group, group ends, tile -> group
        gend_h = (goff[:64].astype(np.int64) + n2 + np.asarray(counts, dtype=np.int64)).astype(np.int32)
        tables = np.concatenate([goff.astype(np.int32), np.zeros(3, dtype=np.int32), (goff[:64].astype(np.int64) + n2).astype(np.int32), gend_h,
                                 np.asarray(tile_group, dtype=np.int32)])          # 68 + 64 + 64 + tiles: every table 16-byte aligned
        tab_d = torch.from_numpy(tables).to(dev)
        goff_d, xbase, gend, tile_group_d = tab_d[:65], tab_d[68:132], tab_d[132:196], tab_d[196:]
        Et = E
        if split:
            Et = plan["entries_all"]
        elif plan.get("objects"):
            Et = plan["entries"] if plan["entries"] is not None else int(plan["n_total"][0])
        dest = torch.empty(max(Et, 1), dtype=torch.int32, device=dev)
        # row of X entry e = first X row of its window's group + its rank among that window's entries in list order (what a stable sort
        # by window gives; sgc_bucket_place computes the ranks directly)
        kern = TUNING.plan_kernels
        if E > 0 and kern:
            self._bucket_place(gather, E, None, None, 0, 64, xbase, dest, None, 0)
        elif E > 0:
            cex = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
            skeys, order = torch.sort((gather[:E] & 63).long(), stable=True)
            base = torch.from_numpy(goff[:64].astype(np.int64) + n2 - cex).to(dev)
            dest[order] = (base[skeys] + torch.arange(E, device=dev)).int()
        if Et > E and kern:                          # the pseudo-pairs' own windows: row goff[w] + ps
            _lib.check(self.lib.sgc_window_rows_objects(_lib.ptr(gather[E:]), Et - E, _lib.ptr(goff_d), P, _lib.ptr(dest[E:]), self._st()),
                       "sgc_window_rows_objects")
        elif Et > E:
            code = gather[E:Et].long()
            dest[E:Et] = (goff_d[:64].long()[code & 63] + (code >> 6) - P).int()
        dest_conv = dest
        if split:
            # destination of every entry of the CONV list: the same (pair, window) sits at  first(pair, all) + its rank in the pair's
            # rectangle  in the list of all X windows (a pair is in the conv list with all of its windows or with none)
            Ec = plan["entries"]
            if kern:
                dest_conv = torch.empty(max(Ec, 1), dtype=torch.int32, device=dev)
                _lib.check(self.lib.sgc_window_rows_conv(_lib.ptr(plan["gather"]), Ec, _lib.ptr(plan["incl"]), _lib.ptr(plan["incl_all"]),
                                                         _lib.ptr(dest), _lib.ptr(dest_conv), self._st()), "sgc_window_rows_conv")
            else:
                pair_k = (plan["gather"][:Ec] >> 6).long()
                first = lambda inc: torch.cat([inc.new_zeros(1), inc[:-1]]).long()
                dest_conv = dest[torch.arange(Ec, device=dev) - first(plan["incl"])[pair_k] + first(plan["incl_all"])[pair_k]].contiguous()
        return dict(goff=goff_d, goff_host=goff, gend=gend, tile_group=tile_group_d, dest=dest, dest_conv=dest_conv,
                    rows=int(goff[64]), E=E, E_total=Et, n2=n2)

    def _bucket_place(self, codes, n, sub_idx, obj_img, img_key, n_keys, base, out, seg, mode):
        """Stable placement of a window list by key (``sgc_bucket_place_seg``: two-level kernels, scratch from the workspace)."""
        lib = self.lib
        lib.sgc_bucket_place_scratch_ints.restype = ctypes.c_long
        need = int(lib.sgc_bucket_place_scratch_ints(int(n), int(n_keys)))
        scratch = self.scratch.get("bucket_cnt", max(need, 1), torch.int32)
        _lib.check(lib.sgc_bucket_place_seg(_lib.ptr(codes), int(n), _lib.ptr(sub_idx), _lib.ptr(obj_img), int(img_key), int(n_keys), _lib.ptr(base),
                                            _lib.ptr(out), _lib.ptr(seg), int(mode), _lib.ptr(scratch), _c_long(need), self._st()),
                   "sgc_bucket_place_seg")

    def fc1_shared(self, wm, ywm, bbox, sub_idx, obj_idx, incl, P, n_obj, h1, dropout, seed, order=None):
        """fc1 + ReLU (+ dropout) from the window-major rows: grouped GEMM, per-object 2-D prefix sums, per-pair assembly."""
        lib, sc = self.lib, self.scratch
        w1p = self.w["w1p"]                          # deferred copy: made here (after the wait for fc1.weight's all-gather, if one is in flight)
        owm = sc.get("owm", wm["rows"] * int(lib.sgc_fc1_products_pitch()), torch.float32)
        oxh = None
        if TUNING.fc1_x16:
            oxh = sc.get("oxh", wm["rows"] * 4096, torch.float16)
            self._timed("fc1_fwd_windows", lambda: _lib.check(lib.sgc_fc1_windows_gemm_x16(
                _lib.ptr(ywm), _lib.ptr(w1p), _lib.ptr(wm["tile_group"]), _lib.ptr(wm["goff"]), wm["n2"], _lib.ptr(owm), _lib.ptr(oxh), wm["rows"],
                self._st()), "sgc_fc1_windows_gemm_x16"))
        else:
            self._timed("fc1_fwd_windows", lambda: _lib.check(lib.sgc_fc1_windows_gemm(
                _lib.ptr(ywm), _lib.ptr(w1p), _lib.ptr(wm["tile_group"]), _lib.ptr(owm), wm["rows"], self._st()), "sgc_fc1_windows_gemm"))
        S = sc.get("fc1_S", wm["n2"] * 81 * 4096, torch.float32)
        self._timed("fc1_fwd_integral", lambda: _lib.check(lib.sgc_fc1_integral(_lib.ptr(owm), _lib.ptr(wm["goff"]), wm["n2"], _lib.ptr(S), self._st()),
                                                           "sgc_fc1_integral"))
        own = None
        if TUNING.fc1_own_sums:            # S'_j[R_j] per object: read once per pair instead of four corners
            own = sc.get("fc1_own", max(n_obj, 1) * 4096, torch.float32)
            _lib.check(lib.sgc_fc1_own_rect_sums(_lib.ptr(S), _lib.ptr(bbox), n_obj, _lib.ptr(own), self._st()), "sgc_fc1_own_rect_sums")
        if order is not None and (not TUNING.assemble_by_subject or int(order.shape[0]) != P):
            order = None
        if oxh is not None:
            self._timed("fc1_fwd_assemble", lambda: _lib.check(lib.sgc_fc1_assemble_x16(
                _lib.ptr(S), _lib.ptr(oxh), _lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), _lib.ptr(incl), _lib.ptr(wm["dest"]), n_obj,
                _lib.ptr(self.w["bf1"]), int(dropout), ctypes.c_uint(seed), _lib.ptr(h1), P, _lib.ptr(own), _lib.ptr(order), self._st()),
                "sgc_fc1_assemble_x16"))
        else:
            self._timed("fc1_fwd_assemble", lambda: _lib.check(lib.sgc_fc1_assemble_ordered(
                _lib.ptr(S), _lib.ptr(owm), _lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), _lib.ptr(incl), _lib.ptr(wm["dest"]), n_obj,
                _lib.ptr(self.w["bf1"]), int(dropout), ctypes.c_uint(seed), _lib.ptr(h1), P, _lib.ptr(own), _lib.ptr(order), self._st()),
                "sgc_fc1_assemble_ordered"))

    def conv3_shared(self, plan, z, U, V, bbox, obj_img, sub_idx, obj_idx, P, y, am, y_bf, keep=None, wm=None):
        """conv3 + ReLU + pool with the per-object part computed once per object (``csrc/kernels_shared.hip``): U / V hold
        n_obj + n_img objects, the last n_img the empty-box backgrounds of the images; ``plan`` from ``shared_plan``.
        ``z`` [P + 2 n_obj (+ n_img)] padded maps: the real pairs' expansion is there already, the pseudo-pairs' (and background
        maps') is written here; ``keep=(z_bf, amz)`` (training): same-shaped bf16 copy and routing codes for the backward.
        ``wm`` (``window_major_rows``): ``y`` / ``y_bf`` are the window-major buffers of the shared fc1 and nothing is assembled
        per pair; with ``plan['objects']`` the pseudo-pairs are computed on their own windows only (second level).
        Returns what the backward needs."""
        lib, sc = self.lib, self.scratch
        own = self.ws if keep is not None else sc
        n_obj, n_img = int(obj_img.shape[0]), plan["n_img"]
        n2 = 2 * n_obj
        objects = bool(plan["objects"]) and wm is not None
        n_tail = n2 + (n_img if objects else 0)                                  # + the all-background map of every image
        bg_codes = raw_n = None
        if TUNING.plan_kernels and obj_img.dtype == torch.int32:
            tabs = sc.get("ps_tables", 2 * n_tail + 64 * n_img + 4, torch.int32)
            ps_sub, ps_obj = tabs[:n_tail], tabs[n_tail:2 * n_tail]
            bg_codes, raw_n = tabs[2 * n_tail:2 * n_tail + 64 * n_img], tabs[2 * n_tail + 64 * n_img:2 * n_tail + 64 * n_img + 1]
            _lib.check(lib.sgc_pseudo_pair_tables(_lib.ptr(obj_img), n_obj, n_img, P, int(objects), _lib.ptr(ps_sub), _lib.ptr(ps_obj),
                                                  _lib.ptr(bg_codes), _lib.ptr(raw_n), self._st()), "sgc_pseudo_pair_tables")
        else:
            ar = torch.arange(n_obj, dtype=torch.int32, device=self.device)
            bg = obj_img.to(torch.int32) + n_obj                                 # every object's background = its image's
            ps_sub, ps_obj = torch.cat([ar, bg]), torch.cat([bg, ar])
            if objects:
                bgs = torch.arange(n_obj, n_obj + n_img, dtype=torch.int32, device=self.device)
                ps_sub, ps_obj = torch.cat([ps_sub, bgs]), torch.cat([ps_obj, bgs])
        zt = z[P * 18 * 18 * 512:]
        if keep is None:
            _lib.check(lib.sgc_pair_expand(_lib.ptr(U), _lib.ptr(V), _lib.ptr(ps_sub), _lib.ptr(ps_obj), _lib.ptr(zt), n_tail, ELEM_F16, self._st()),
                       "sgc_pair_expand")
        else:
            _lib.check(lib.sgc_pair_expand_train(_lib.ptr(U), _lib.ptr(V), _lib.ptr(ps_sub), _lib.ptr(ps_obj), _lib.ptr(zt),
                                                 _lib.ptr(keep[0][(P - keep[2]) * 18 * 18 * 512:]), _lib.ptr(keep[1][P * 256 * 256:]), n_tail, self._st()),
                       "sgc_pair_expand_train")
        gather, incl = plan["gather"], plan["incl"]
        out = dict(plan, n2=n2, wm=wm)
        if wm is not None:
            am_ps = am[P * 65536:] if am is not None else None                   # routing codes of the pseudo-pairs: behind the real pairs'
            if objects:
                # second level: the pseudo-pairs' windows R_o are entries of the window list; the other rows are the background maps'
                zb = zt[n2 * 18 * 18 * 512:]
                y_bg = sc.get("y_bg", n_img * 65536, torch.float16)
                ybf_bg = sc.get("ybf_bg", n_img * 65536, torch.bfloat16) if y_bf is not None else None
                am_bg = own.get("am_bg", n_img * 65536, torch.uint8) if am is not None else None
                self._timed("conv3_fwd_objects", lambda: _lib.check(lib.sgc_conv3_relu_pool(
                    _lib.ptr(zb), _lib.ptr(self.w["w3r"]), _lib.ptr(self.w["b3"]), _lib.ptr(y_bg), _lib.ptr(am_bg), _lib.ptr(ybf_bg), n_img,
                    self._st()), "sgc_conv3_relu_pool"))
                _lib.check(lib.sgc_shared_objects_fill_rows(_lib.ptr(bbox), _lib.ptr(obj_img), n_obj, _lib.ptr(wm["goff"]), _lib.ptr(y_bg),
                                                            _lib.ptr(ybf_bg), _lib.ptr(am_bg), _lib.ptr(y), _lib.ptr(y_bf), _lib.ptr(am_ps),
                                                            self._st()), "sgc_shared_objects_fill_rows")
                out["am_bg"] = am_bg
            else:
                self._timed("conv3_fwd_objects", lambda: _lib.check(lib.sgc_conv3_relu_pool_wm(
                    _lib.ptr(zt), _lib.ptr(self.w["w3r"]), _lib.ptr(self.w["b3"]), _lib.ptr(wm["goff"]), _lib.ptr(y), _lib.ptr(am_ps), _lib.ptr(y_bf),
                    n2, self._st()), "sgc_conv3_relu_pool_wm"))
            lin = plan.get("lin") if objects else None
            if lin is None:
                self._timed("conv3_fwd_windows", lambda: _lib.check(lib.sgc_conv3_relu_pool_windows_wm(
                    _lib.ptr(z), _lib.ptr(self.w["w3r"]), _lib.ptr(self.w["b3"]), _lib.ptr(gather), _lib.ptr(plan["n_total"]), _lib.ptr(wm["dest_conv"]),
                    plan["bound"], _lib.ptr(y), _lib.ptr(am), _lib.ptr(y_bf), self._st()), "sgc_conv3_relu_pool_windows_wm"))
            else:
                # linear pairs: their X windows are combined from the pre-activations of the per-object entries (the tail of the list:
                # the same launch stores their accumulators) and of the images' background maps (a 64 n_img-window launch of their own)
                n_pe = plan["entries"] - plan["entries_real"]
                n_raw = n_pe + 64 * n_img
                raw = sc.get("raw_pre", n_raw * 4 * 1024, torch.float32)
                self._timed("conv3_fwd_windows", lambda: _lib.check(lib.sgc_conv3_relu_pool_windows_wm_raw(
                    _lib.ptr(z), _lib.ptr(self.w["w3r"]), _lib.ptr(self.w["b3"]), _lib.ptr(gather), _lib.ptr(plan["n_total"]), _lib.ptr(wm["dest_conv"]),
                    plan["bound"], _lib.ptr(y), _lib.ptr(am), _lib.ptr(y_bf), _lib.ptr(raw), plan["entries_real"], self._st()),
                    "sgc_conv3_relu_pool_windows_wm_raw"))
                if bg_codes is None:
                    bg_codes = ((P + n2 + torch.arange(n_img, device=self.device, dtype=torch.int32))[:, None] * 64
                                + torch.arange(64, device=self.device, dtype=torch.int32)[None, :]).reshape(-1).contiguous()
                    raw_n = torch.full((1,), 64 * n_img, dtype=torch.int32, device=self.device)
                self._timed("conv3_fwd_raw", lambda: _lib.check(lib.sgc_conv3_windows_raw(
                    _lib.ptr(z), _lib.ptr(self.w["w3r"]), _lib.ptr(bg_codes), _lib.ptr(raw_n), 64 * n_img, _lib.ptr(raw[n_pe * 4096:]), self._st()),
                    "sgc_conv3_windows_raw"))
                self._timed("conv3_fwd_linear", lambda: _lib.check(lib.sgc_windows_linear_forward(
                    _lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), _lib.ptr(obj_img), n_obj, P, _lib.ptr(lin["gather"]), _lib.ptr(lin["n"]),
                    lin["max"], _lib.ptr(plan["incl_all"]), _lib.ptr(wm["dest"]), _lib.ptr(raw), _c_long(n_pe), _lib.ptr(self.w["b3"]),
                    _lib.ptr(y), _lib.ptr(y_bf), _lib.ptr(am), _lib.ptr(lin["drow"]), self._st()), "sgc_windows_linear_forward"))
            out["am_ps"] = am_ps
            return out
        am_ps = own.get("am_ps", n2 * 65536, torch.uint8) if am is not None else None
        y_ps = sc.get("y_ps", n2 * 65536, torch.float16)
        ybf_ps = sc.get("ybf_ps", n2 * 65536, torch.bfloat16) if y_bf is not None else None
        self._timed("conv3_fwd_objects", lambda: _lib.check(lib.sgc_conv3_relu_pool(
            _lib.ptr(zt), _lib.ptr(self.w["w3r"]), _lib.ptr(self.w["b3"]), _lib.ptr(y_ps), _lib.ptr(am_ps), _lib.ptr(ybf_ps), n2,
            self._st()), "sgc_conv3_relu_pool"))
        self._timed("conv3_fwd_windows", lambda: _lib.check(lib.sgc_conv3_relu_pool_windows(
            _lib.ptr(z), _lib.ptr(self.w["w3r"]), _lib.ptr(self.w["b3"]), _lib.ptr(gather), _lib.ptr(plan["n_total"]), plan["bound"], _lib.ptr(y),
            _lib.ptr(am), _lib.ptr(y_bf), self._st()), "sgc_conv3_relu_pool_windows"))
        self._timed("conv3_fwd_assemble", lambda: _lib.check(lib.sgc_shared_windows_assemble(
            _lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), P, n_obj, _lib.ptr(y_ps), _lib.ptr(am_ps), _lib.ptr(ybf_ps),
            _lib.ptr(y), _lib.ptr(am), _lib.ptr(y_bf), self._st()), "sgc_shared_windows_assemble"))
        out["am_ps"] = am_ps
        return out

    def pair_trunk(self, U, V, sub_idx, obj_idx, lsub, lobj, train=False, seeds=(0, 0), keep_argmax=False,
                   iou_mask=None, dense=None, shared=None, pair_order=None) -> PairOutputs:
        lib, ws, cfg = self.lib, self.ws, self.cfg
        P = int(sub_idx.shape[0])
        Ppad = (P + 63) // 64 * 64
        plan, Pt = None, P
        if shared is not None:
            n_obj = int(shared[1].shape[0])
            n_img = int(U.numel()) // (1024 * 512) - n_obj                 # the background objects behind the real ones
            wm_mode = shared_fc1_enabled()
            plan = self.shared_plan(shared[0], sub_idx, obj_idx, P, shared[2], n_obj=n_obj, n_img=n_img,
                                    objects=wm_mode and shared_objects_enabled(), obj_img=shared[1])
            Pt = P + 2 * n_obj + n_img
        z = ws.get("z_pad", Pt * 18 * 18 * 512, torch.float16)     # border stays zero: only interiors are written
        # the row plan (a sort, a dozen small launches, four blocking host-to-device copies) goes BEFORE the pair expansion: its
        # launches are then behind the host when the 1.4 ms expansion kernel starts, instead of leaving the GPU idle between them
        wm = self.window_major_rows(plan, P, 2 * n_obj) if shared is not None and wm_mode else None
        self.expand(U, V, sub_idx, obj_idx, P, z, dense=dense, pixrect=None if plan is None else plan["pixrect"])
        am = ws.get("argmax", Pt * 65536, torch.uint8) if keep_argmax else None
        h1 = ws.get("h1", Ppad * 4096, torch.float16)
        if shared is not None and wm_mode:
            # conv3 and fc1 over shared windows: the rows fc1 multiplies are written window-major, y [P, 65536] never exists
            ywm = ws.get("ywm", wm["rows"] * 1024, torch.float16)
            self.conv3_shared(plan, z, U, V, shared[0], shared[1], sub_idx, obj_idx, P, ywm, am, None, wm=wm)
            self.fc1_shared(wm, ywm, shared[0], sub_idx, obj_idx, plan.get("incl_all", plan["incl"]), P, n_obj, h1, train, seeds[0], order=pair_order)
        else:
            y = ws.get("y", Ppad * 65536, torch.float16)
            if shared is not None:
                self.conv3_shared(plan, z, U, V, shared[0], shared[1], sub_idx, obj_idx, P, y, am, None)
            else:
                self._timed("conv3_fwd", lambda: _lib.check(lib.sgc_conv3_relu_pool(_lib.ptr(z), _lib.ptr(self.w["w3r"]), _lib.ptr(self.w["b3"]),
                                                                                   _lib.ptr(y), _lib.ptr(am), _lib.ptr(None), P, self._st()),
                                                            "sgc_conv3_relu_pool"))
            w1p = self.w["w1p"]                               # deferred copy (Weights): made here, outside the timed launch
            self._timed("fc1_fwd", lambda: _lib.check(lib.sgc_fc1_relu(_lib.ptr(y), _lib.ptr(w1p), _lib.ptr(self.w["bf1"]), _lib.ptr(h1), P, 65536,
                                                                       int(train), ctypes.c_uint(seeds[0]), self._st()), "sgc_fc1_relu"))
        p = ws.get("p", Ppad * 512, torch.float32)
        self._timed("fc2_fwd", lambda: _lib.check(lib.sgc_fc2_labels_relu(_lib.ptr(h1), _lib.ptr(self.w["w2m"]), _lib.ptr(self.w["bf2"]), _lib.ptr(lsub),
                                           _lib.ptr(lobj), _lib.ptr(sub_idx), _lib.ptr(obj_idx), _lib.ptr(p), P,
                                           int(train), ctypes.c_uint(seeds[1]), self._st()), "sgc_fc2_labels_relu"))
        return self.head(p, P, iou_mask)

    def head(self, p, P, iou_mask=None) -> PairOutputs:
        lib, cfg, dev = self.lib, self.cfg, self.device
        R = cfg.num_relations
        hier = cfg.hierarchical
        nc = 3 if hier else 1
        rel = torch.empty(P, R, dtype=torch.float32, device=dev)
        sup = torch.empty(P, 3, dtype=torch.float32, device=dev) if hier else None
        conn = torch.empty(P, dtype=torch.float32, device=dev)
        cconf = torch.empty(P, nc, dtype=torch.float32, device=dev)
        cpred = torch.empty(P, nc, dtype=torch.int32, device=dev)
        T = self.T
        f = ctypes.c_float
        _lib.check(lib.sgc_bayes_head(_lib.ptr(p), _lib.ptr(self.w["head_wt"]), _lib.ptr(self.w["head_b"]), P,
                                      cfg.num_geometric if hier else R, cfg.num_possessive if hier else 0,
                                      cfg.num_semantic if hier else 0, int(hier), f(T[0]), f(T[1]), f(T[2]),
                                      _lib.ptr(rel), _lib.ptr(sup), _lib.ptr(conn), _lib.ptr(cconf), _lib.ptr(cpred),
                                      _lib.ptr(iou_mask), self._st()), "sgc_bayes_head")
        return PairOutputs(rel, sup, conn, p[:P * 512].view(P, 512), cconf, cpred)

    # ------------------------------------------------------------------ fused entry
    def forward_pairs(self, image_feature, image_depth, obj_img, bbox, cats, super_mh, sub_idx, obj_idx, train=False,
                      seeds=(0, 0), keep_argmax=False, iou_mask=None, dense=None, select=None, shared_windows=None, pair_order=None) -> PairOutputs:
        """One call per minibatch: image maps -> per-object halves -> all pairs.
        ``select`` ([P] bool / uint8 device tensor): run the per-pair trunk (expansion, conv3, fc1, fc2, head) ONLY for the selected
        pairs and scatter the results into full-size outputs; the other pairs get confidence -inf (exactly what the overlap filter
        gives them in the evaluator, ``evaluator.py:131-134``), prediction 0, zero log-probs and hidden vectors."""
        self.verify_checks()
        a_img = self.image_maps(image_feature, image_depth)
        share = shared_conv3_enabled(shared_windows, int(sub_idx.shape[0]))
        uv = self.object_halves(a_img, obj_img, bbox, with_bg=share,
                                regions=shared_windows.get("conv2_windows") if (share and isinstance(shared_windows, dict)) else None)
        shared = (bbox, obj_img, shared_windows) if share else None
        lsub, lobj = self.label_vectors(cats, super_mh)
        self._lsub, self._lobj = lsub, lobj
        if select is None:
            return self.pair_trunk(uv[0], uv[1], sub_idx, obj_idx, lsub, lobj, train, seeds, keep_argmax, iou_mask, dense, shared, pair_order)
        sel = select.bool()
        if share and shared_windows is not None:
            # With conv3 / fc1 over shared windows a pair whose boxes do not overlap has (almost) no pair-specific window: skipping it
            # saves nothing, while a pair SUBSET loses the host's window counts (read-backs) and the dense expansion.  Compute every
            # pair and blank the unselected ones - the same outputs (measured 21.4 vs 34.2 ms per 8x64 minibatch).
            out = self.pair_trunk(uv[0], uv[1], sub_idx, obj_idx, lsub, lobj, train, seeds, keep_argmax, iou_mask, dense, shared, pair_order)
            drop = ~sel
            out.relation[drop] = 0
            if out.super_relation is not None:
                out.super_relation[drop] = 0
            out.connectivity[drop] = 0
            hidden = out.hidden.clone()
            hidden[drop] = 0
            out.cand_conf[drop] = -math.inf
            out.cand_pred[drop] = 0
            return PairOutputs(out.relation, out.super_relation, out.connectivity, hidden, out.cand_conf, out.cand_pred)
        idx = torch.nonzero(sel).flatten()
        P, Ps = int(sub_idx.shape[0]), int(idx.shape[0])
        cfg, dev = self.cfg, self.device
        nc = 3 if cfg.hierarchical else 1
        full = PairOutputs(torch.zeros(P, cfg.num_relations, device=dev), torch.zeros(P, 3, device=dev) if cfg.hierarchical else None,
                           torch.zeros(P, device=dev), torch.zeros(P, 512, device=dev),
                           torch.full((P, nc), -math.inf, device=dev), torch.zeros(P, nc, dtype=torch.int32, device=dev))
        if Ps == 0:
            return full
        dense_s = None
        if dense is not None:
            img_ptr, pid, max_n = dense
            rank = (torch.cumsum(sel.int(), 0) - 1).int()
            ok = pid >= 0
            pc = pid.clamp(min=0).long()
            dense_s = (img_ptr, torch.where(ok & sel[pc], rank[pc], torch.full_like(pid, -1)).contiguous(), max_n)
        if shared is not None and isinstance(shared[2], dict):
            # the counts describe the full pair list: for a subset only the object part stays exact, the rest is an upper bound
            shared = (shared[0], shared[1], None)
        out = self.pair_trunk(uv[0], uv[1], sub_idx[idx].contiguous(), obj_idx[idx].contiguous(), lsub, lobj, train, seeds, keep_argmax,
                              None if iou_mask is None else iou_mask[idx].contiguous(), dense_s, shared)
        full.relation[idx] = out.relation
        if full.super_relation is not None:
            full.super_relation[idx] = out.super_relation
        full.connectivity[idx] = out.connectivity
        full.hidden[idx] = out.hidden
        full.cand_conf[idx] = out.cand_conf
        full.cand_pred[idx] = out.cand_pred
        return full

    def compat_forward(self, hs, ho, c1, c2, mh1, mh2, train=False, seeds=(0, 0)) -> PairOutputs:
        """The reference's per-step call on PRE-MASKED inputs (``model.py:170``): row k of ``hs`` / ``ho`` [b,257,32,32] is the subject /
        object crop of pair k (inference: no context is kept)."""
        dev = self.device
        b = int(hs.shape[0])
        a_s = self.image_maps(hs, None, roles=(0,), tag="cs")
        a_o = self.image_maps(ho, None, roles=(1,), tag="co")
        F = self.cfg.feature_size
        full = torch.tensor([[0, F, 0, F]], dtype=torch.int32, device=dev).repeat(b, 1).contiguous()
        ids = torch.arange(b, dtype=torch.int32, device=dev)
        U = self.object_halves({0: a_s[0]}, ids, full, roles=(0,))[0]
        V = self.object_halves({1: a_o[1]}, ids, full, roles=(1,))[1]
        lsub, _ = self.label_vectors(c1, mh1)
        _, lobj = self.label_vectors(c2, mh2)
        return self.pair_trunk(U, V, ids, ids, lsub, lobj, train=train, seeds=seeds)

    # ====================================================================== training (forward + backward)
    def prep_bwd_weights(self, sd, fc1_sync=None):
        """bf16 transposed / flipped weight copies for the data-gradient GEMMs."""
        dev = self.device
        g = lambda k: sd[k].detach().to(dev, torch.float32)
        w = self.w
        if TUNING.weight_kernels:
            fc2 = w["fc2_full"].contiguous()
            w2mT = self.ws.get("w2mT", 4096 * 512, torch.bfloat16)
            self._permute_cast(fc2, w2mT, 1, [4096, 512], [1, int(fc2.shape[1])])
            w["w2mT"] = w2mT.view(4096, 512)
        else:
            w["w2mT"] = w["fc2_full"][:, :4096].t().contiguous().to(torch.bfloat16)
        self._prep_bwd_trunk_weights(sd, g, fc1_sync)

    # (own pixel q, tap k) combinations that reach coordinate c of a window's 4 x 4 input patch: see sgc_windows_dgrad_patches
    _PATCH_OPTS = staticmethod(lambda c: [(0, 0)] if c == 0 else ([(1, 2)] if c == 3 else [(0, c), (1, c - 1)]))

    def _prep_bwd_trunk_weights(self, sd, g, fc1_sync):
        w = self.w

        def make_w1pT():
            if fc1_sync is not None:
                fc1_sync()
            with torch.no_grad():
                return self._transpose_cast(g("fc1.weight").contiguous(), "w1pT", torch.bfloat16, 1, 64, 1024, 64 * 65536, 64, 65536,
                                            64, 4096, 1024 * 4096)
        w.defer("w1pT", make_w1pT)
        if not TUNING.weight_kernels:
            return self._prep_bwd_trunk_weights_torch(g)
        c2, c3 = g("conv2_1.weight").contiguous(), g("conv3_1.weight").contiguous()
        # conv3 with flipped taps and swapped channel roles, K order (chunk of 64 c_out, tap, c_out): the data gradient as a convolution
        wd3 = self.ws.get("wd3", 512 * 9216, torch.bfloat16)
        self._permute_cast(c3, wd3, 1, [512, 16, 9, 64], [9, 64 * 4608, -1, 4608], src_off=8)
        w["wd3"] = wd3.view(512, 9216)
        # [(tap, c_in)][c_out]: second operand of the column form of the data gradient over pair-specific windows
        w3col = self.ws.get("w3col", 4608 * 1024, torch.bfloat16)
        self._permute_cast(c3, w3col, 1, [9, 512, 1024], [1, 9, 4608])
        w["w3col"] = w3col.view(4608, 1024)
        # patch form of the same data gradient: per patch pixel pp = (py, px) the transposed tap matrices of its combinations, stacked along K
        opts = self._PATCH_OPTS
        d_off, d_ld, s_off, base = [], [], [], 0
        for py in range(4):
            for px in range(4):
                taps = [ky * 3 + kx for _, ky in opts(py) for _, kx in opts(px)]
                for j, t in enumerate(taps):
                    d_off.append(base + j * 1024); d_ld.append(len(taps) * 1024); s_off.append(t)
                base += 512 * len(taps) * 1024
        w3patch = self.ws.get("w3patch", base, torch.bfloat16)
        n = len(d_off)
        L = ctypes.c_long * n
        _lib.check(self.lib.sgc_segment_cast(_lib.ptr(c3), _lib.ptr(w3patch), 1, 512, 1024, _c_long(9), _c_long(4608), n, L(*d_off), L(*d_ld), L(*s_off),
                                             self._st()), "sgc_segment_cast")
        w["w3patch"] = w3patch
        # sparse form of the same data gradient (csrc/kernels_dgrad_sp.hip): per slot [512 c_in][(c_out, own pixel of the slot's set)]
        w3sp = self.ws.get("w3sp", 20 * 512 * 2048, torch.bfloat16)
        _lib.check(self.lib.sgc_windows_dgrad_sparse_weights(_lib.ptr(c3), _lib.ptr(w3sp), self._st()), "sgc_windows_dgrad_sparse_weights")
        w["w3sp"] = w3sp
        wd2 = self.ws.get("wd2", 2 * 128 * 4608, torch.bfloat16)
        for r in (0, 1):
            self._permute_cast(c2, wd2, 1, [128, 8, 9, 64], [9, 64 * 2304, -1, 2304], src_off=r * 128 * 9 + 8, dst_off=r * 128 * 4608)
        w["wd2"] = wd2.view(2, 128, 4608)

    def _prep_bwd_trunk_weights_torch(self, g):
        w = self.w
        w["wd3"] = conv_k_layout(g("conv3_1.weight").flip(2, 3).permute(1, 0, 2, 3)).to(torch.bfloat16).contiguous()
        # [(tap, c_in)][c_out]: second operand of the column form of the data gradient over pair-specific windows
        w["w3col"] = g("conv3_1.weight").permute(2, 3, 1, 0).reshape(9 * 512, 1024).to(torch.bfloat16).contiguous()
        # patch form of the same data gradient (sgc_windows_dgrad_patches): for every pixel pp = (py, px) of a window's 4 x 4 input patch
        # the tap matrices of the (own pixel q, tap k) combinations with q + k = pp, stacked along K: [512 c_in][combinations x 1024 c_out]
        w3 = g("conv3_1.weight")
        opts = self._PATCH_OPTS
        w["w3patch"] = torch.cat([torch.cat([w3[:, :, ky, kx].t() for _, ky in opts(py) for _, kx in opts(px)], dim=1).reshape(-1)
                                  for py in range(4) for px in range(4)]).to(torch.bfloat16).contiguous()
        c2 = g("conv2_1.weight")
        w["wd2"] = torch.stack([conv_k_layout(c2[:, r * 128:(r + 1) * 128].flip(2, 3).permute(1, 0, 2, 3))
                                for r in (0, 1)]).to(torch.bfloat16).contiguous()


    def _slab_sum(self, slabs, n, count):
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.sgc_slab_sum(_lib.ptr(slabs), _lib.ptr(out), _c_long(n), int(count), 0, self._st()), "sgc_slab_sum")
        return out


    def _colsum(self, X, rows, cols, elem=ELEM_BF16, blocks=None):
        if blocks is None:
            blocks = max(1, min(512, rows // 64))
        blocks = int(max(1, min(blocks, rows)))
        part = self.scratch.get("colsum_part", blocks * cols, torch.float32)
        _lib.check(self.lib.sgc_colsum(elem, _lib.ptr(X), _lib.ptr(part), _c_long(rows), cols, blocks, self._st()), "sgc_colsum")
        return self._slab_sum(part, cols, blocks)


    def _to_bf16(self, name, src, n):
        dst = self.scratch.get(name, n, torch.bfloat16)
        self._timed("convert", lambda: _lib.check(self.lib.sgc_convert_f16_bf16(_lib.ptr(src), _lib.ptr(dst), _c_long(n), self._st()),
                                                  "sgc_convert_f16_bf16"))
        return dst


    def train_forward(self, image_feature, image_depth, obj_img, bbox, cats, super_mh, sub_idx, obj_idx, seeds=(0, 0),
                      dropout=True, dense=None, role_inputs=None, cats_obj=None, super_mh_obj=None, shared_windows=None, pair_order=None) -> "TrainContext":
        """Forward that keeps what the backward needs (pool argmaxes, expansion routing mask).
        ``role_inputs=(h_sub, h_obj)``: the reference's per-step call on PRE-MASKED ``[b,257,32,32]`` inputs (``model.py:170``):
        row k of each is the subject / object crop of pair k, with labels ``cats`` / ``cats_obj``; every crop is its own
        "image" with one full-size box."""
        lib, ws, sc = self.lib, self.ws, self.scratch
        self.verify_checks()
        ctx = TrainContext()
        ctx.n_obj = int(obj_img.shape[0])
        ctx.P = P = int(sub_idx.shape[0])
        ctx.Ppad = Ppad = (P + 63) // 64 * 64
        ctx.obj_img, ctx.bbox, ctx.sub_idx, ctx.obj_idx = obj_img, bbox, sub_idx, obj_idx
        ctx.cats = (cats, cats if cats_obj is None else cats_obj)
        ctx.super_mh = (super_mh, super_mh if role_inputs is None else super_mh_obj)
        ctx.dropout, ctx.seeds = dropout, seeds
        if role_inputs is None:
            ctx.n_img = int(image_feature.shape[0])
            ctx.a_img = self.image_maps(image_feature, image_depth)
            ctx.x = (self._x, self._x)
        else:
            ctx.n_img = int(role_inputs[0].shape[0])
            a_s = self.image_maps(role_inputs[0], None, roles=(0,), tag="s")
            x_s = self._x
            a_o = self.image_maps(role_inputs[1], None, roles=(1,), tag="o")
            ctx.a_img, ctx.x = {0: a_s[0], 1: a_o[1]}, (x_s, self._x)
        share = role_inputs is None and shared_conv3_enabled(shared_windows, P)   # per-step calls: every crop is one full-size box, nothing is shared
        ctx.uv = self.object_halves(ctx.a_img, obj_img, bbox, with_bg=share,
                                    regions=shared_windows.get("conv2_windows") if (share and isinstance(shared_windows, dict)) else None)
        ctx.lsub, lobj_same = self.label_vectors(ctx.cats[0], ctx.super_mh[0])
        ctx.lobj = lobj_same if role_inputs is None else self.label_vectors(ctx.cats[1], ctx.super_mh[1])[1]
        # the per-pair backward (TUNING.shared_bwd off, A/B) reads every pixel of z / amz; the shared one only those next to X windows
        narrow = share and TUNING.shared_bwd
        wm_mode = narrow and shared_fc1_enabled()
        plan = self.shared_plan(bbox, sub_idx, obj_idx, P, shared_windows, keep=True, n_obj=ctx.n_obj, n_img=ctx.n_img,
                                objects=wm_mode and shared_objects_enabled(), obj_img=obj_img) if share else None
        Pt = P + (2 * ctx.n_obj + ctx.n_img if share else 0)         # pseudo-pairs and background maps live behind the real pairs
        z = sc.get("z_pad", Pt * 18 * 18 * 512, torch.float16)
        # bf16 copy of z for the weight gradients.  With the shared backward in its patch form the real pairs' copy is never read: the
        # patch gather converts the f16 rows it gathers (sgc_windows_im2patch_f16), so the expansion writes 1 KB less per pixel and
        # only the pseudo-pairs / background maps behind the real pairs get a bf16 map (``z_bf_base`` = pair index of the buffer's first map)
        ctx.z_bf_base = P if (narrow and TUNING.patch_wgrad and dense is not None and 0 < dense[2] <= 150) else 0
        z_bf = ws.get("z_pad_bf", (Pt - ctx.z_bf_base) * 18 * 18 * 512, torch.bfloat16)
        amz = ws.get("amz", Pt * 256 * 256, torch.uint8)             # two 4-bit routing codes per byte
        wm = self.window_major_rows(plan, P, 2 * ctx.n_obj) if wm_mode else None     # before the expansion: see forward_pairs
        self.expand(ctx.uv[0], ctx.uv[1], sub_idx, obj_idx, P, z, None if ctx.z_bf_base else z_bf, amz, dense=dense,
                    pixrect=plan["pixrect"] if narrow else None)
        ctx.z_bf = z_bf
        am = ws.get("argmax", Pt * 65536, torch.uint8)              # conv3 routing codes (shared path: only the rows of listed windows)
        h1 = ws.get("h1", Ppad * 4096, torch.float16)
        ctx.shared, ctx.y, ctx.y_bf = None, None, None
        if wm_mode:
            # conv3 and fc1 over shared windows: y and its bf16 copy exist only as the window-major rows fc1 multiplies
            ywm = sc.get("ywm", wm["rows"] * 1024, torch.float16)
            ywm_bf = ws.get("ywm_bf", wm["rows"] * 1024, torch.bfloat16)
            ctx.shared = self.conv3_shared(plan, z, ctx.uv[0], ctx.uv[1], bbox, obj_img, sub_idx, obj_idx, P, ywm, am, ywm_bf, keep=(z_bf, amz, ctx.z_bf_base), wm=wm)
            ctx.shared["ywm_bf"] = ywm_bf
            self.fc1_shared(wm, ywm, bbox, sub_idx, obj_idx, plan.get("incl_all", plan["incl"]), P, ctx.n_obj, h1, dropout, seeds[0], order=pair_order)
        else:
            y = sc.get("y", Ppad * 65536, torch.float16)
            y_bf = ws.get("y_bf", Ppad * 65536, torch.bfloat16)     # bf16 copy for the fc1 weight gradient, written by the same epilogue
            if Ppad > P:
                Workspace._zero(y_bf[P * 65536:])
            if share:
                ctx.shared = self.conv3_shared(plan, z, ctx.uv[0], ctx.uv[1], bbox, obj_img, sub_idx, obj_idx, P, y, am, y_bf, keep=(z_bf, amz, ctx.z_bf_base))
            else:
                self._timed("conv3_fwd", lambda: _lib.check(lib.sgc_conv3_relu_pool(_lib.ptr(z), _lib.ptr(self.w["w3r"]), _lib.ptr(self.w["b3"]),
                                                                                   _lib.ptr(y), _lib.ptr(am), _lib.ptr(y_bf), P, self._st()),
                                                            "sgc_conv3_relu_pool"))
            w1p = self.w["w1p"]                               # deferred copy (Weights): made here, outside the timed launch
            self._timed("fc1_fwd", lambda: _lib.check(lib.sgc_fc1_relu(_lib.ptr(y), _lib.ptr(w1p), _lib.ptr(self.w["bf1"]), _lib.ptr(h1), P, 65536,
                                                                       int(dropout), ctypes.c_uint(seeds[0]), self._st()), "sgc_fc1_relu"))
            ctx.y, ctx.y_bf = y, y_bf
        p = ws.get("p", Ppad * 512, torch.float32)
        self._timed("fc2_fwd", lambda: _lib.check(lib.sgc_fc2_labels_relu(_lib.ptr(h1), _lib.ptr(self.w["w2m"]), _lib.ptr(self.w["bf2"]), _lib.ptr(ctx.lsub),
                                           _lib.ptr(ctx.lobj), _lib.ptr(sub_idx), _lib.ptr(obj_idx), _lib.ptr(p), P, int(dropout),
                                           ctypes.c_uint(seeds[1]), self._st()), "sgc_fc2_labels_relu"))
        ctx.z, ctx.amz, ctx.am, ctx.h1, ctx.p = z, amz, am, h1, p
        ctx.out = self.head(p, P)
        return ctx


    def train_backward(self, ctx: "TrainContext", coefs, sub_csr, obj_csr, img_ptr, splits=8, grad_hook=None, dp_extra=None,
                       cs_coef=None, upstream=None):
        """Backward of the whole path.  Returns (loss scalar tensor, {reference parameter name: f32 gradient}).
        ``upstream=(g_rel, g_sup, g_conn, g_hidden)`` (``coefs`` None): no loss here - gradients of the outputs handed over by the
        caller's autograd (per-step ``forward()``); the returned loss is None."""
        lib, cfg, dev, w = self.lib, self.cfg, self.device, self.w
        ws = self.scratch              # everything allocated below is transient; what the forward kept lives in ``ctx`` / ``self.ws``
        P, Ppad, n_obj, n_img = ctx.P, ctx.Ppad, ctx.n_obj, ctx.n_img
        st = self._st
        hier = cfg.hierarchical
        R = cfg.num_relations
        scale = 2.0 if ctx.dropout else 1.0
        f = ctypes.c_float
        grads: Dict[str, torch.Tensor] = {}
        slabs_n = ctypes.c_int(0)

        # ---- head: loss, dlogits, d(fc2 pre-activation)
        dl = ws.get("dl", P * 64, torch.float32)
        loss_i = ws.get("loss_i", P, torch.float32)
        dpre = ws.get("dpre", Ppad * 512, torch.bfloat16)
        if Ppad > P:
            Workspace._zero(dpre[P * 512:])
        T = self.T
        if upstream is None:
            tgt, ca, cb, cc, cy = coefs
            _lib.check(lib.sgc_head_loss_bwd(_lib.ptr(ctx.out.relation), _lib.ptr(ctx.out.super_relation), _lib.ptr(ctx.out.connectivity),
                                             _lib.ptr(ctx.p), _lib.ptr(tgt), _lib.ptr(ca), _lib.ptr(cb), _lib.ptr(cc), _lib.ptr(cy),
                                             _lib.ptr(w["head_w"]), P, cfg.num_geometric if hier else R,
                                             cfg.num_possessive if hier else 0, cfg.num_semantic if hier else 0, int(hier),
                                             f(T[0]), f(T[1]), f(T[2]), f(scale), _lib.ptr(dl), _lib.ptr(loss_i), _lib.ptr(dpre),
                                             _lib.ptr(dp_extra), _lib.ptr(cs_coef), _lib.ptr(ctx.out.cand_conf),
                                             _lib.ptr(ctx.out.cand_pred), st()),
                       "sgc_head_loss_bwd")
            loss = loss_i.sum() if P > 0 else torch.zeros((), device=dev)      # P scalars: host-side glue
        else:
            g_rel, g_sup, g_conn, g_hid = (None if g is None else g.to(dev, torch.float32).contiguous() for g in upstream)
            _lib.check(lib.sgc_head_bwd_upstream(_lib.ptr(ctx.out.relation), _lib.ptr(ctx.out.super_relation), _lib.ptr(ctx.p),
                                                 _lib.ptr(g_rel), _lib.ptr(g_sup), _lib.ptr(g_conn), _lib.ptr(g_hid),
                                                 _lib.ptr(w["head_w"]), P, cfg.num_geometric if hier else R,
                                                 cfg.num_possessive if hier else 0, cfg.num_semantic if hier else 0, int(hier),
                                                 f(T[0]), f(T[1]), f(T[2]), f(scale), _lib.ptr(dl), _lib.ptr(dpre), st()),
                       "sgc_head_bwd_upstream")
            loss = None
        # The backward is two chains.  DATA gradients (fc2 -> fc1 -> un-pool -> conv3 -> pair contraction -> conv2 -> masks) run
        # on the caller's stream: each feeds the next.  WEIGHT gradients (one GEMM per layer + slab sums / transposes / bias
        # column sums) only have to be there when the optimizer runs, so they go to a side stream as soon as their two operands
        # exist: the HBM-bound kernels of the data chain (un-pool + pack, pair contraction, converts) and the tails of its GEMMs
        # then overlap with weight-gradient GEMM blocks instead of leaving the matrix cores idle.  ``side`` orders the side
        # stream after everything enqueued so far.  Measured -0.3 ... -2.2 ms per step in five alternated A/B pairs on three boxes
        # (small: every GEMM block owns its CU - 146 KiB LDS, all VGPRs - so kernels of two streams interleave block by block
        # instead of co-residing; profiles/README.md); results bit-identical either way (tests/test_configs_gpu.py).
        # ``SGC_BWD_STREAMS=0`` = one stream (used for per-kernel profiles: durations of overlapped launches mean little).
        side = self._side_chain()
        sl = ws.get("slabs", 32 * 1024 * 4608, torch.float32)      # split-K slabs of the weight-gradient chain (largest user: conv3)

        # ---- head weights
        with side():
            chunk = max(16, (P + 255) // 256)
            nb = (P + chunk - 1) // chunk
            part = ws.get("head_part", nb * 64 * 513, torch.float32)
            _lib.check(lib.sgc_head_wgrad(_lib.ptr(dl), _lib.ptr(ctx.p), _lib.ptr(part), P, chunk, st()), "sgc_head_wgrad")
            hw = self._slab_sum(part, 64 * 513, nb).view(64, 513)
            names = (["fc3_1", "fc3_2", "fc3_3", "fc5", "fc4"] if hier else ["fc3", "fc4"])
            sizes = ([cfg.num_geometric, cfg.num_possessive, cfg.num_semantic, 3, 1] if hier else [R, 1])
            r0 = 0
            for nm, sz in zip(names, sizes):
                grads[nm + ".weight"] = hw[r0:r0 + sz, :512].contiguous()
                grads[nm + ".bias"] = hw[r0:r0 + sz, 512].contiguous()
                r0 += sz

            # ---- fc2 weights: main block (split-K GEMM) + label columns (per-object row sums scattered by label)
            h1_bf = self._to_bf16("h1_bf", ctx.h1, Ppad * 4096)
            self._timed("fc2_wgrad", lambda: _lib.check(lib.sgc_fc2_wgrad(_lib.ptr(dpre), _lib.ptr(h1_bf), _lib.ptr(sl), Ppad, 32, ctypes.byref(slabs_n), st()),
                       "sgc_fc2_wgrad"))
            gfc2 = torch.empty_like(w["fc2_full"])                       # main block + label columns are written below
            ld2 = int(gfc2.shape[1])
            use_mh = ctx.super_mh[0] is not None and cfg.dataset == "vg"
            n_lab = 2 * cfg.num_classes + (2 * cfg.num_super_classes if use_mh else 0)
            if 4096 + n_lab < ld2:                                       # VG model without super-categories: nothing writes the
                gfc2[:, 4096 + n_lab:].zero_()                           # multi-hot columns - their gradient is zero, not garbage
            _lib.check(lib.sgc_slab_sum_ld(_lib.ptr(sl), _lib.ptr(gfc2), 512, 4096, _c_long(ld2), slabs_n.value, st()), "sgc_slab_sum_ld")
            dls = torch.empty(n_obj, 512, dtype=torch.float32, device=dev)
            dlo = torch.empty(n_obj, 512, dtype=torch.float32, device=dev)
            _lib.check(lib.sgc_segment_sum_rows(_lib.ptr(dpre), _lib.ptr(sub_csr[0]), _lib.ptr(sub_csr[1]), _lib.ptr(dls), n_obj, 512, st()),
                       "sgc_segment_sum_rows")
            _lib.check(lib.sgc_segment_sum_rows(_lib.ptr(dpre), _lib.ptr(obj_csr[0]), _lib.ptr(obj_csr[1]), _lib.ptr(dlo), n_obj, 512, st()),
                       "sgc_segment_sum_rows")
            mh_s, mh_o = (ctx.super_mh if use_mh else (None, None))
            cats_s, cats_o = (c if c.dtype == torch.int64 else c.long() for c in ctx.cats)
            _lib.check(lib.sgc_label_grads(_lib.ptr(dls), _lib.ptr(dlo), _lib.ptr(cats_s), _lib.ptr(cats_o), _lib.ptr(mh_s), _lib.ptr(mh_o),
                                           n_obj, cfg.num_classes, cfg.num_super_classes if use_mh else 0, _lib.ptr(gfc2), ld2, 4096, st()),
                       "sgc_label_grads")
            grads["fc2.weight"] = gfc2
            grads["fc2.bias"] = self._colsum(dpre, Ppad, 512)

        # ---- fc2 data gradient
        dh1 = ws.get("dh1", Ppad * 4096, torch.bfloat16)
        if Ppad > P:
            Workspace._zero(dh1[P * 4096:])
        self._timed("fc2_dgrad", lambda: _lib.check(lib.sgc_fc2_dgrad(_lib.ptr(dpre), _lib.ptr(w["w2mT"]), _lib.ptr(ctx.h1), _lib.ptr(dh1), P, f(scale), st()),
                   "sgc_fc2_dgrad"))

        if getattr(ctx, "generic", None) is not None:        # sizes other than 128 / 32: the f32 per-pair trunk (engine_generic.py)
            self._generic_trunk_backward(ctx, dh1, grads, side)
            side.join()
            return loss, grads

        # ---- fc1
        if getattr(ctx, "shared", None) is not None and ctx.shared.get("wm") is not None:
            dy = self._fc1_backward_rows(ctx, dh1, sub_csr, obj_csr, side, grads, grad_hook)
        else:
            dy = self._fc1_backward_pairs(ctx, dh1, side, grads, grad_hook)

        # ---- conv3
        nparts = ctypes.c_int(0)
        shared = ctx.shared if (getattr(ctx, "shared", None) is not None and TUNING.shared_bwd) else None
        n_objx = n_obj + (n_img if shared is not None else 0)      # the images' background objects take part in the conv2 backward
        if shared is not None:
            dz = self._conv3_backward_shared(ctx, shared, dy, sub_csr, obj_csr, img_ptr, side, sl, grads)
        else:
            dz = self._conv3_backward_pairs(ctx, dy, side, sl, grads)

        # ---- pair contraction + conv2 + masks + conv1 (per-role buffers: the side stream may still read role 0's while role 1 runs)
        gc2 = torch.empty(512, 256, 3, 3, dtype=torch.float32, device=dev)
        # second level of sharing on: an object's dU is zero outside the pixel rectangle of its pseudo-pair (every pair of the object
        # lives inside it), so the conv2 data gradient runs on those cells + one more ring (csrc/kernels_shared.hip: conv2_bwd_regions)
        c2_list = c2_n = None
        if (shared is not None and TUNING.conv2_bwd_regions and bool(shared.get("objects")) and shared.get("wm") is not None and n_obj > 0):
            c2_list = ws.get("c2b_list", n_objx * 256 + 64, torch.int32)
            c2_n = ws.get("c2b_n", 4, torch.int32)
            _lib.check(lib.sgc_conv2_bwd_regions(_lib.ptr(ctx.bbox), n_obj, n_objx, 1, _lib.ptr(c2_list), _lib.ptr(c2_n), st()), "sgc_conv2_bwd_regions")
        mapU, mapA = 34 * 34 * 512, 34 * 34 * 128
        for r, csr in ((0, sub_csr), (1, obj_csr)):
            dU = ws.get("dU_pad_%d" % r, n_objx * mapU, torch.bfloat16)
            if shared is not None:
                self._timed("contract", lambda: _lib.check(lib.sgc_pair_contract_windows(
                    _lib.ptr(dz), _lib.ptr(ctx.amz), _lib.ptr(csr[0]), _lib.ptr(csr[1]), _lib.ptr(shared["pixrect"]), _lib.ptr(img_ptr),
                    r, P, n_obj, n_img, int(bool(shared.get("objects")) and shared.get("wm") is not None), _lib.ptr(dU), st()),
                    "sgc_pair_contract_windows"))
            else:
                self._timed("contract", lambda: _lib.check(lib.sgc_pair_contract(_lib.ptr(dz), _lib.ptr(ctx.amz), _lib.ptr(csr[0]), _lib.ptr(csr[1]), _lib.ptr(dU), n_obj, st()),
                           "sgc_pair_contract"))
            with side():
                a_pad = self.ws.get("a_pad_%d" % r, n_objx * mapA, torch.float16)      # kept by the forward
                a_bf = self._to_bf16("a_pad_bf", a_pad, n_objx * mapA)
                self._timed("conv2_wgrad", lambda: _lib.check(lib.sgc_conv2_wgrad(_lib.ptr(dU), _lib.ptr(a_bf), _lib.ptr(sl), n_objx, 0, ctypes.byref(slabs_n), st()),
                           "sgc_conv2_wgrad"))
                dW2r = self._slab_sum(sl, 512 * 1152, slabs_n.value)
                gc2[:, r * 128:(r + 1) * 128] = dW2r.view(512, 3, 3, 128).permute(0, 3, 1, 2)
                if r == 1:
                    grads["conv2_1.bias"] = self._colsum(dU, n_objx * 34 * 34, 512)
            da = ws.get("da", n_objx * 1024 * 128, torch.bfloat16)
            if c2_list is not None:
                Workspace._zero(da)                  # unlisted cells: the gradient there is exactly zero
                self._timed("conv2_dgrad", lambda: _lib.check(lib.sgc_conv2_dgrad_regions(
                    _lib.ptr(dU), _lib.ptr(w["wd2"][r]), _lib.ptr(c2_list), _lib.ptr(c2_n), n_objx * 256, _lib.ptr(da), st()), "sgc_conv2_dgrad_regions"))
            else:
                self._timed("conv2_dgrad", lambda: _lib.check(lib.sgc_conv2_dgrad(_lib.ptr(dU), _lib.ptr(w["wd2"][r]), _lib.ptr(da), n_objx, st()), "sgc_conv2_dgrad"))
            dcst_bg = None
            if shared is not None:                   # the background objects are constant everywhere: all of their gradient goes to tanh(b1)
                dcst_bg = self.ws.get("dcst_bg_%d" % r, 128, torch.float32)      # on THIS stream: ``da`` is rewritten by the next role
                torch.sum(da[n_obj * 1024 * 128:].view(-1, 128).float(), 0, out=dcst_bg)
            dA = ws.get("dA", n_img * 1024 * 128, torch.float32)
            cpart = ws.get("dcst_part_%d" % r, n_img * 64 * 128, torch.float32)
            _lib.check(lib.sgc_object_masked_maps_bwd(_lib.ptr(da), _lib.ptr(img_ptr), _lib.ptr(ctx.bbox), _lib.ptr(dA), _lib.ptr(cpart),
                                                      ctypes.byref(nparts), n_img, 32, 128, st()), "sgc_object_masked_maps_bwd")
            n_cst = nparts.value
            dp1 = ws.get("dpre1_%d" % r, n_img * 1024 * 128, torch.bfloat16)
            _lib.check(lib.sgc_tanh_bwd(_lib.ptr(dA), _lib.ptr(ctx.a_img[r]), _lib.ptr(dp1), _c_long(n_img * 1024 * 128), st()),
                       "sgc_tanh_bwd")
            with side():
                x_bf = self._to_bf16("x_bf", ctx.x[r], n_img * 1024 * XC)
                _lib.check(lib.sgc_conv1_wgrad(_lib.ptr(dp1), _lib.ptr(x_bf), _lib.ptr(sl), n_img * 1024, XC, 16, ctypes.byref(slabs_n), st()),
                           "sgc_conv1_wgrad")
                dW1 = self._slab_sum(sl, 128 * XC, slabs_n.value).view(128, XC)
                nm = "conv1_%d" % (r + 1)
                grads[nm + ".weight"] = dW1[:, :257].reshape(128, 257, 1, 1).contiguous()
                tb = torch.tanh(w["b1"][r])
                dcst = self._slab_sum(cpart, 128, n_cst)
                if dcst_bg is not None:
                    dcst = dcst + dcst_bg
                grads[nm + ".bias"] = self._colsum(dp1, n_img * 1024, 128) + dcst * (1 - tb * tb)
        with side():
            grads["conv2_1.weight"] = gc2
        side.join()                              # the caller's stream continues only after every gradient is complete
        return loss, grads

    def _fc1_finish_wgrad(self, dW1p, dh1, Ppad, grads, grad_hook):
        """dW1p [4096][(window, channel)] -> the reference's column order (channel*64 + window), bias gradient, early all-reduce hook."""
        lib, dev, st = self.lib, self.device, self._st
        if self.fc1_grad_gemm_order:
            # handed over as the GEMM wrote it (a view of the engine's scratch: valid until the next backward); the optimizer's fused
            # update un-permutes it on the fly
            grads["fc1.weight"] = dW1p[:4096 * 65536].view(4096, 65536)
            if grad_hook is not None:
                grad_hook("fc1.weight", grads["fc1.weight"])
            grads["fc1.bias"] = self._colsum(dh1, Ppad, 4096)
            return
        gfc1 = torch.empty(4096, 65536, dtype=torch.float32, device=dev)
        _lib.check(lib.sgc_transpose_cast(_lib.ptr(dW1p), _lib.ptr(gfc1), 2, 4096, 16, _c_long(65536), _c_long(64), _c_long(1024),
                                          _c_long(65536), _c_long(4096), _c_long(64), st()), "sgc_transpose_cast")
        grads["fc1.weight"] = gfc1
        if grad_hook is not None:              # largest gradient (97 % of the bytes) is ready first: overlap its all-reduce
            grad_hook("fc1.weight", grads["fc1.weight"])
        grads["fc1.bias"] = self._colsum(dh1, Ppad, 4096)

    def _fc1_backward_pairs(self, ctx, dh1, side, grads, grad_hook):
        """fc1 backward as two [pairs, 65536] GEMMs; returns dy [Ppad*64, 1024] (pair-major pooled gradient)."""
        lib, w, ws, st, P, Ppad = self.lib, self.w, self.scratch, self._st, ctx.P, ctx.Ppad
        with side():
            dW1p = ws.get("dW1p", 4096 * 65536, torch.float32)
            self._timed("fc1_wgrad", lambda: _lib.check(lib.sgc_fc1_wgrad(_lib.ptr(dh1), _lib.ptr(ctx.y_bf), _lib.ptr(dW1p), Ppad, 65536, st()),
                                                        "sgc_fc1_wgrad"))
            self._fc1_finish_wgrad(dW1p, dh1, Ppad, grads, grad_hook)
        dy = ws.get("dy", Ppad * 65536, torch.bfloat16)
        w1pT = w["w1pT"]                               # deferred copy (Weights): made here, outside the timed launch
        self._timed("fc1_dgrad", lambda: _lib.check(lib.sgc_fc1_dgrad(_lib.ptr(dh1), _lib.ptr(w1pT), _lib.ptr(dy), P, 65536, st()), "sgc_fc1_dgrad"))
        return dy

    def _fc1_backward_rows(self, ctx, dh1, sub_csr, obj_csr, side, grads, grad_hook):
        """fc1 backward over the window-major rows (``csrc/kernels_shared.hip``): per-object sums of dh1 + one copy of dh1 per X entry,
        then the grouped weight- and data-gradient GEMMs; returns dywm [rows, 1024] (window-major pooled gradient)."""
        lib, w, ws, st, sh = self.lib, self.w, self.scratch, self._st, ctx.shared
        wm = sh["wm"]
        gwm = ws.get("gwm", wm["rows"] * 4096, torch.bfloat16)
        self._timed("fc1_bwd_rows", lambda: (
            _lib.check(lib.sgc_fc1_gsum(_lib.ptr(dh1), _lib.ptr(ctx.bbox), _lib.ptr(ctx.sub_idx), _lib.ptr(ctx.obj_idx), _lib.ptr(sub_csr[0]),
                                        _lib.ptr(sub_csr[1]), _lib.ptr(obj_csr[0]), _lib.ptr(obj_csr[1]), _lib.ptr(wm["goff"]), ctx.n_obj,
                                        _lib.ptr(gwm), st()), "sgc_fc1_gsum"),
            _lib.check(lib.sgc_fc1_xrows(_lib.ptr(dh1), _lib.ptr(sh.get("gather_all", sh["gather"])), _lib.ptr(wm["dest"]), wm["E"], _lib.ptr(wm["goff"]),
                                         _lib.ptr(wm["gend"]), _lib.ptr(gwm), _lib.ptr(sh["ywm_bf"]), st()), "sgc_fc1_xrows")))
        dy = ws.get("dywm", wm["rows"] * 1024, torch.bfloat16)

        def wgrad():
            with side():
                dW1p = ws.get("dW1p", 4096 * 65536, torch.float32)
                self._timed("fc1_wgrad", lambda: _lib.check(lib.sgc_fc1_windows_wgrad(_lib.ptr(gwm), _lib.ptr(sh["ywm_bf"]), _lib.ptr(wm["goff"]),
                                                                                      _lib.ptr(dW1p), wm["rows"], st()), "sgc_fc1_windows_wgrad"))
                self._fc1_finish_wgrad(dW1p, dh1, ctx.Ppad, grads, grad_hook)

        def dgrad():
            w1pT = w["w1pT"]                               # deferred copy (Weights): made here, outside the timed launch
            self._timed("fc1_dgrad", lambda: _lib.check(lib.sgc_fc1_windows_dgrad(_lib.ptr(gwm), _lib.ptr(w1pT), _lib.ptr(wm["tile_group"]),
                                                                                  _lib.ptr(dy), wm["rows"], st()), "sgc_fc1_windows_dgrad"))
        # GEMM beside GEMM buys nothing on this chip (two ping-pong GEMMs on two streams: 13.5 ms against 13.4 back to back) while an
        # HBM-bound kernel beside a GEMM hides ~40 % of its time (profiles/r03_overlap_microbench.txt).  ``TUNING.gemms_apart``: the
        # weight-gradient GEMM is enqueued AFTER the data-gradient GEMM - ``side()`` orders the side stream behind everything enqueued on
        # the caller's stream so far - and so runs beside the row sums / un-pool kernels that follow the data gradient instead of
        # beside the data gradient itself.  Off: round 2's order (both GEMMs at once).
        if TUNING.gemms_apart:
            dgrad()
            wgrad()
        else:
            wgrad()
            dgrad()
        return dy

    def _conv3_backward_pairs(self, ctx, dy, side, sl, grads):
        """conv3 backward over every window of every pair (no per-object sharing): bias + weight gradient, returns dz.
        Weight gradient on the sparse matrix cores: the pooled gradient + the arg-max byte ARE the 2:4-compressed operand
        (csrc/gemm_tn_sp.h; one pass over dy packs it and writes the bias partials).  Data gradient with the un-pool inside its
        operand staging (sgc_conv3_dgrad_pooled): the 21 GB un-pooled tensor is neither written nor read.  (The dense weight
        gradient / the two-pass un-pool, sgc_conv3_wgrad / sgc_unpool_relu_bwd / sgc_conv3_dgrad, remain in the C-ABI and are tested
        against these in tests/test_gemm_gpu.py; the step no longer switches to them.)"""
        lib, w, ws, st, P = self.lib, self.w, self.scratch, self._st, ctx.P
        slabs_n = ctypes.c_int(0)
        bpart = ws.get("b3_part", 2048 * 1024, torch.float32)
        nparts = ctypes.c_int(0)
        z_bf = ctx.z_bf
        if getattr(ctx, "z_bf_base", 0):
            raise RuntimeError("the per-pair conv3 backward needs the bf16 copy of z that this forward did not write "
                               "(TUNING changed between forward and backward)")
        pack_a = ws.get("w3_pack_a", P * 4 * 1024 * 64, torch.uint8)
        pack_i = ws.get("w3_pack_i", P * 4 * 1024 * 8, torch.uint8)
        self._timed("unpool", lambda: _lib.check(lib.sgc_unpool_relu_bwd_pack(
            _lib.ptr(dy), _lib.ptr(ctx.am), None, _lib.ptr(bpart), ctypes.byref(nparts), _lib.ptr(pack_a),
            _lib.ptr(pack_i), P, st()), "sgc_unpool_relu_bwd_pack"))
        n_b3 = nparts.value
        with side():
            grads["conv3_1.bias"] = self._slab_sum(bpart, 1024, n_b3)
            self._timed("conv3_wgrad", lambda: _lib.check(lib.sgc_conv3_wgrad_sparse(
                None, None, _lib.ptr(z_bf), _lib.ptr(pack_a), _lib.ptr(pack_i), _lib.ptr(sl), P, 0,
                ctypes.byref(slabs_n), st()), "sgc_conv3_wgrad_sparse"))
            dW3r = self._slab_sum(sl, 1024 * 4608, slabs_n.value)
            grads["conv3_1.weight"] = dW3r.view(1024, 3, 3, 512).permute(0, 3, 1, 2).contiguous()
        dz = ws.get("dz", P * 256 * 512, torch.bfloat16)
        self._timed("conv3_dgrad", lambda: _lib.check(lib.sgc_conv3_dgrad_pooled(_lib.ptr(dy), _lib.ptr(ctx.am), _lib.ptr(w["wd3"]), _lib.ptr(dz), P, st()),
                                                      "sgc_conv3_dgrad_pooled"))
        return dz

    def _conv3_backward_shared(self, ctx, sh, dy, sub_csr, obj_csr, img_ptr, side, sl, grads):
        """Backward of ``conv3_shared`` (autodiff of that graph).  Per-object part: the gradient rows of copied windows, summed per
        object (by the shared fc1's data gradient, or by ``sgc_shared_windows_assemble_bwd`` from a pair-major ``dy``), go through
        the ordinary conv3 backward of whole maps - the 2*n_obj pseudo-pairs, or with the second level only the n_img background
        maps, the pseudo-pairs' own windows being window-list entries like the X windows.  Listed windows: compact column form
        (un-pool -> [rows,1024]; weight gradient = rows^T x im2col(z); data gradient = rows x W^T -> col2im).
        Returns dz [(P + 2 n_obj + n_img) * 256, 512]: a pair's rows exist only inside its pixel rectangle (``plan['pixrect']``)."""
        lib, w, ws, st, P, n_obj, n_img = self.lib, self.w, self.scratch, self._st, ctx.P, ctx.n_obj, ctx.n_img
        n2, wm, objects = sh["n2"], sh.get("wm"), bool(sh.get("objects")) and sh.get("wm") is not None
        E = sh["entries"]
        Epad = (E + 15) // 16 * 16                                   # 4 rows per entry: the GEMMs want a multiple of 64 rows
        slabs_n, slabs_x, nparts, nparts_x = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        gather, gn = sh["gather"], sh["n_total"]
        z_bf = ctx.z_bf
        Pt = P + n2 + n_img
        dz = ws.get("dz", Pt * 256 * 512, torch.bfloat16)
        # ---- whole maps: the pseudo-pairs (first level) or the background maps (second level), gradient = sums of copied rows
        if objects:
            n_maps, map0 = n_img, P + n2
            dy_maps = ws.get("dy_bg", n_img * 65536, torch.bfloat16)
            _lib.check(lib.sgc_shared_objects_bg_grad(_lib.ptr(ctx.bbox), _lib.ptr(img_ptr), n_obj, n_img, _lib.ptr(wm["goff"]), _lib.ptr(dy),
                                                      _lib.ptr(dy_maps), st()), "sgc_shared_objects_bg_grad")
            am_maps = sh["am_bg"]
        else:
            n_maps, map0 = n2, P
            dy_maps = ws.get("dy_ps", n2 * 65536, torch.bfloat16)
            if wm is not None:
                # ``dy`` is the window-major gradient of the shared fc1: the per-object rows are already sums; bring them to pair-major order
                idx = (wm["goff"][:64].long()[None, :] + torch.arange(n2, device=self.device)[:, None]).reshape(-1)
                torch.index_select(dy.view(-1, 1024), 0, idx, out=dy_maps.view(-1, 1024))
            else:
                self._timed("conv3_bwd_assemble", lambda: _lib.check(lib.sgc_shared_windows_assemble_bwd(
                    _lib.ptr(ctx.bbox), _lib.ptr(ctx.sub_idx), _lib.ptr(ctx.obj_idx), _lib.ptr(sub_csr[0]), _lib.ptr(sub_csr[1]), _lib.ptr(obj_csr[0]),
                    _lib.ptr(obj_csr[1]), n_obj, _lib.ptr(dy), _lib.ptr(dy_maps), st()), "sgc_shared_windows_assemble_bwd"))
            am_maps = sh["am_ps"]
        dest = wm["dest_conv"] if wm is not None else None
        lin = sh.get("lin") if objects else None
        zb0 = getattr(ctx, "z_bf_base", 0)
        z_bf_maps = z_bf[(map0 - zb0) * 18 * 18 * 512:]
        dz_maps = dz[map0 * 256 * 512:]
        bpart = ws.get("b3_part", 2048 * 1024, torch.float32)
        bpart_x = ws.get("b3_part_x", 1024 * 1024, torch.float32)
        pack_a = ws.get("w3_pack_a", n_maps * 4 * 1024 * 64, torch.uint8)
        pack_i = ws.get("w3_pack_i", n_maps * 4 * 1024 * 8, torch.uint8)
        dy3_bg = bpart_l = None
        if lin is None:
            self._timed("unpool_objects", lambda: _lib.check(lib.sgc_unpool_relu_bwd_pack(
                _lib.ptr(dy_maps), _lib.ptr(am_maps), None, _lib.ptr(bpart), ctypes.byref(nparts), _lib.ptr(pack_a), _lib.ptr(pack_i), n_maps, st()),
                "sgc_unpool_relu_bwd_pack"))
        else:
            # the background maps also collect (minus) the gradient of the linear pairs' windows: their un-pooled gradient is dense,
            # so they take the two-pass un-pool and the dense conv3 backward (n_img maps)
            dy3_bg = ws.get("dy3_bg_pad", n_maps * 18 * 18 * 1024, torch.bfloat16)        # created zeroed: the halo stays zero
            self._timed("unpool_objects", lambda: _lib.check(lib.sgc_unpool_relu_bwd(
                _lib.ptr(dy_maps), _lib.ptr(am_maps), _lib.ptr(dy3_bg), _lib.ptr(bpart), ctypes.byref(nparts), n_maps, st()), "sgc_unpool_relu_bwd"))
        # ---- listed windows: compact un-pool.  The real pairs' windows in front of the list (one non-zero per window and channel in
        # their un-pooled gradient) go through the SPARSE forms of both backward GEMMs, which pack their operand from the pooled rows:
        # only the entries behind them (per-object entries: sums of several windows, + the boundary tile) are un-pooled
        e_real = sh.get("entries_real")
        sp_ok = e_real is not None and dest is not None and Epad and int(e_real) >= 4096
        e_spw = (int(e_real) // 16) * 16 if (sp_ok and TUNING.patch_wgrad and TUNING.sparse_wgrad) else 0
        e_spd = (int(e_real) // 256) * 256 if (sp_ok and TUNING.patch_dgrad and TUNING.sparse_dgrad and "w3sp" in w) else 0
        e_un0 = min(e_spw, e_spd)
        dy3x = ws.get("dy3x", max(Epad, 16) * 4 * 1024, torch.bfloat16)
        self._timed("unpool_windows", lambda: _lib.check(lib.sgc_windows_unpool_from(
            _lib.ptr(dy), _lib.ptr(ctx.am), _lib.ptr(gather), _lib.ptr(gn), _lib.ptr(dest), e_un0, Epad - e_un0, _lib.ptr(dy3x[e_un0 * 4096:]),
            _lib.ptr(bpart_x), ctypes.byref(nparts_x), st()), "sgc_windows_unpool_from"))
        nparts_s, bpart_s = ctypes.c_int(0), None
        if e_spd:
            spa = ws.get("w3d_pack_a", 4 * e_spd * 1024, torch.bfloat16)
            spi = ws.get("w3d_pack_i", 4 * e_spd * 64, torch.int32)
            if e_un0 == e_spd and e_un0 > 0:          # the un-pool pass skipped these windows: their bias partial sums come from the packer
                bpart_s = ws.get("b3_part_s", 1024 * 1024, torch.float32)
            self._timed("dgrad_pack_windows", lambda: _lib.check(lib.sgc_windows_dgrad_sparse_pack(
                _lib.ptr(dy), _lib.ptr(ctx.am), _lib.ptr(gather), _lib.ptr(dest), e_spd, _lib.ptr(spa), _lib.ptr(spi), _lib.ptr(bpart_s),
                ctypes.byref(nparts_s), st()), "sgc_windows_dgrad_sparse_pack"))
        elif e_un0 > 0:
            raise RuntimeError("internal: windows skipped by the un-pool pass without a sparse data gradient to count their bias")
        if lin is not None:
            # transpose of sgc_windows_linear_forward: + into the un-pooled rows of the two per-object entries, - into the background map
            e_real = sh["entries_real"]
            bpart_l = ws.get("b3_part_l", 64 * n_img * 1024, torch.float32)
            self._timed("linear_bwd", lambda: (
                _lib.check(lib.sgc_windows_linear_backward_objects(
                    _lib.ptr(ctx.bbox), _lib.ptr(ctx.sub_idx), _lib.ptr(ctx.obj_idx), _lib.ptr(sub_csr[0]), _lib.ptr(sub_csr[1]), _lib.ptr(obj_csr[0]),
                    _lib.ptr(obj_csr[1]), n_obj, P, _lib.ptr(gather), e_real, E - e_real, _lib.ptr(sh["incl_all"]), _lib.ptr(wm["dest"]),
                    _lib.ptr(dy), _lib.ptr(ctx.am), _lib.ptr(dy3x), st()), "sgc_windows_linear_backward_objects"),
                _lib.check(lib.sgc_windows_linear_backward_bg(
                    _lib.ptr(lin["gather"]), _lib.ptr(lin["drow"]), _lib.ptr(lin["order"]), _lib.ptr(lin["seg"]), n_img, _lib.ptr(dy),
                    _lib.ptr(ctx.am), _lib.ptr(dy3_bg), _lib.ptr(bpart_l), st()), "sgc_windows_linear_backward_bg")))
        with side():
            gb = self._slab_sum(bpart, 1024, nparts.value)
            if nparts_x.value:
                gb = gb + self._slab_sum(bpart_x, 1024, nparts_x.value)
            if bpart_s is not None and nparts_s.value:
                gb = gb + self._slab_sum(bpart_s, 1024, nparts_s.value)
            if bpart_l is not None:
                gb = gb + self._slab_sum(bpart_l, 1024, 64 * n_img)
            grads["conv3_1.bias"] = gb
            if lin is None:
                self._timed("conv3_wgrad_objects", lambda: _lib.check(lib.sgc_conv3_wgrad_sparse(
                    None, None, _lib.ptr(z_bf_maps), _lib.ptr(pack_a), _lib.ptr(pack_i), _lib.ptr(sl), n_maps, 0, ctypes.byref(slabs_n), st()),
                    "sgc_conv3_wgrad_sparse"))
            else:
                self._timed("conv3_wgrad_objects", lambda: _lib.check(lib.sgc_conv3_wgrad(
                    _lib.ptr(dy3_bg), _lib.ptr(z_bf_maps), _lib.ptr(sl), n_maps, 0, ctypes.byref(slabs_n), st()), "sgc_conv3_wgrad"))
            zcol = None
            if Epad:
                # im2col + plain ping-pong TN GEMM.  (Rows of z gathered by the window list inside the GEMM block - no column
                # buffer, sgc_windows_wgrad_gather - measured 11.0 ms against 8.3 + 2.3 ms: the nine shifted re-reads of the z rows
                # by different N tiles cost more than the im2col pass; kept in the C-ABI, not used by the step.)
                if TUNING.patch_wgrad:
                    # PATCH form: the 16 pixels of every listed window's input patch, read by the product at (own pixel + tap)
                    zcol = ws.get("zpatch", Epad * 16 * 512, torch.bfloat16)
                    if zb0:        # no bf16 copy of the real pairs' z: gather the f16 rows of the forward and convert
                        self._timed("im2col_windows", lambda: _lib.check(lib.sgc_windows_im2patch_f16(_lib.ptr(ctx.z), _lib.ptr(gather), _lib.ptr(gn), Epad,
                                                                                                     _lib.ptr(zcol), st()), "sgc_windows_im2patch_f16"))
                    else:
                        self._timed("im2col_windows", lambda: _lib.check(lib.sgc_windows_im2patch(_lib.ptr(z_bf), _lib.ptr(gather), _lib.ptr(gn), Epad,
                                                                                                 _lib.ptr(zcol), st()), "sgc_windows_im2patch"))
                else:
                    if zb0:
                        raise RuntimeError("the im2col form needs the bf16 copy of z that this forward did not write (TUNING changed between forward and backward)")
                    zcol = ws.get("zcol", Epad * 4 * 9 * 512, torch.bfloat16)
                    self._timed("im2col_windows", lambda: _lib.check(lib.sgc_windows_im2col(_lib.ptr(z_bf), _lib.ptr(gather), _lib.ptr(gn), Epad,
                                                                                           _lib.ptr(zcol), st()), "sgc_windows_im2col"))

        def auto_splits(k_rows, tiles=72):
            # host mirror of csrc/gemm_tn.h:tn_auto_splits (72 tiles of the [1024][4608] gradient): what a launch with splits = 0 writes
            nk, best = max(int(k_rows) >> 6, 1), 1
            for s_ in range(1, 65):
                if s_ > 1 and nk // s_ < 8:
                    break
                blocks = tiles * s_
                best = s_
                if blocks >= 256 and blocks * 100 >= ((blocks + 255) // 256) * 256 * 95:
                    break
            best = min(best, nk)
            per = (nk + best - 1) // best
            return (nk + per - 1) // per

        def wgrad_windows():
            # the second big GEMM of the window backward.  With ``TUNING.gemms_apart`` it is enqueued after the data-gradient GEMM
            # (the side stream then waits for it) and runs beside col2im / the pair contraction; the im2col above runs beside the
            # data-gradient GEMM.  Round 2 let the two GEMMs run side by side: 18.8 ms for the pair against 7.7 + 7.6 alone.
            with side():
                n_slabs = slabs_n.value
                e_sp = e_spw
                # slab capacity is checked BEFORE anything is launched into the 32-slab buffer (the launches' own counts are the mirror's)
                need = n_slabs + ((auto_splits(e_sp * 4) + (auto_splits((Epad - e_sp) * 4) if Epad > e_sp else 0)) if (Epad and e_sp >= 4096)
                                  else (auto_splits(Epad * 4) if Epad else 0))
                if need > 32:
                    raise RuntimeError("split-K slabs of the conv3 weight gradient (%d) exceed the 32-slab buffer" % need)
                if Epad and e_sp >= 4096:
                    # the real pairs' windows: their un-pooled gradient has ONE non-zero per window and channel (4 consecutive K indices)
                    # - the 2:4 pattern of the sparse matrix cores; packed straight from the pooled rows.  The per-object entries behind
                    # them (sums of several windows: dense) and the boundary tile stay on the dense block.
                    slabs_t = ctypes.c_int(0)
                    pack_a = ws.get("w3x_pack_a", (e_sp // 16) * 1024 * 64, torch.uint8)
                    pack_i = ws.get("w3x_pack_i", (e_sp // 16) * 1024 * 8, torch.uint8)
                    slx = sl[n_slabs * 1024 * 4608:]
                    self._timed("conv3_wgrad_windows", lambda: _lib.check(lib.sgc_windows_wgrad_patch_sparse(
                        _lib.ptr(dy), _lib.ptr(ctx.am), _lib.ptr(gather), _lib.ptr(dest), e_sp, _lib.ptr(zcol), _lib.ptr(pack_a), _lib.ptr(pack_i),
                        _lib.ptr(slx), 0, ctypes.byref(slabs_x), st()), "sgc_windows_wgrad_patch_sparse"))
                    n_slabs += slabs_x.value
                    if Epad > e_sp:
                        slt = sl[n_slabs * 1024 * 4608:]
                        self._timed("conv3_wgrad_windows_tail", lambda: _lib.check(lib.sgc_windows_wgrad_patch(
                            _lib.ptr(dy3x[e_sp * 4 * 1024:]), _lib.ptr(zcol[e_sp * 16 * 512:]), _lib.ptr(slt), (Epad - e_sp) * 4, 0,
                            ctypes.byref(slabs_t), st()), "sgc_windows_wgrad_patch"))
                        n_slabs += slabs_t.value
                elif Epad:
                    slx = sl[n_slabs * 1024 * 4608:]
                    self._timed("conv3_wgrad_windows", lambda: _lib.check((lib.sgc_windows_wgrad_patch if TUNING.patch_wgrad else lib.sgc_windows_wgrad)(
                        _lib.ptr(dy3x), _lib.ptr(zcol), _lib.ptr(slx), Epad * 4, 0, ctypes.byref(slabs_x), st()), "sgc_windows_wgrad"))
                    n_slabs += slabs_x.value
                assert n_slabs <= need, "split-K counts of the launches differ from their host mirror"
                dW3r = self._slab_sum(sl, 1024 * 4608, n_slabs)
                grads["conv3_1.weight"] = dW3r.view(1024, 3, 3, 512).permute(0, 3, 1, 2).contiguous()

        if not TUNING.gemms_apart:
            wgrad_windows()
        # ---- data gradients
        if lin is None:
            self._timed("conv3_dgrad_objects", lambda: _lib.check(lib.sgc_conv3_dgrad_pooled(
                _lib.ptr(dy_maps), _lib.ptr(am_maps), _lib.ptr(w["wd3"]), _lib.ptr(dz_maps), n_maps, st()), "sgc_conv3_dgrad_pooled"))
        else:
            self._timed("conv3_dgrad_objects", lambda: _lib.check(lib.sgc_conv3_dgrad(
                _lib.ptr(dy3_bg), _lib.ptr(w["wd3"]), _lib.ptr(dz_maps), n_maps, st()), "sgc_conv3_dgrad"))
        if Epad and TUNING.patch_dgrad:
            # PATCH form: the 16 pixels of every listed window's input patch leave the GEMM already summed over the taps (K = 1024 x
            # 1 / 2 per output element instead of 1024: 4.7 instead of 8.4 GB of stores per launch at the benchmark's size, and the
            # sum over a pair's windows reads 20 instead of 36 rows per window)
            slots = int(lib.sgc_windows_patch_slots())
            patch = ws.get("xpatch", Epad * slots * 512, torch.bfloat16)
            if e_spd:
                self._timed("conv3_dgrad_windows", lambda: _lib.check(lib.sgc_windows_dgrad_patches_sparse(
                    _lib.ptr(spa), _lib.ptr(spi), e_spd, _lib.ptr(w["w3sp"]), _lib.ptr(patch), st()), "sgc_windows_dgrad_patches_sparse"))
                if Epad > e_spd:          # the dense form's 20 rows per entry behind the sparse form's 16 rows per entry
                    self._timed("conv3_dgrad_windows_tail", lambda: _lib.check(lib.sgc_windows_dgrad_patches(
                        _lib.ptr(dy3x[e_spd * 4096:]), _lib.ptr(w["w3patch"]), _lib.ptr(patch[e_spd * 16 * 512:]), Epad - e_spd, st()),
                        "sgc_windows_dgrad_patches"))
            else:
                self._timed("conv3_dgrad_windows", lambda: _lib.check(lib.sgc_windows_dgrad_patches(_lib.ptr(dy3x), _lib.ptr(w["w3patch"]), _lib.ptr(patch), Epad, st()),
                                                                      "sgc_windows_dgrad_patches"))
            if TUNING.gemms_apart:
                wgrad_windows()                      # side stream: after the data-gradient GEMM, beside the patch sums / the contraction
            self._timed("col2im_windows", lambda: (
                _lib.check(lib.sgc_windows_patch_sum2(_lib.ptr(patch), e_spd, _lib.ptr(ctx.bbox), _lib.ptr(ctx.sub_idx), _lib.ptr(ctx.obj_idx),
                                                      _lib.ptr(sh["incl"]), P, _lib.ptr(dz), st()), "sgc_windows_patch_sum2"),
                _lib.check(lib.sgc_windows_patch_sum_objects2(_lib.ptr(patch), e_spd, _lib.ptr(ctx.bbox), n_obj, P, _lib.ptr(sh["incl"]), _lib.ptr(dz), st()),
                           "sgc_windows_patch_sum_objects2") if objects else None))
        elif Epad:
            col = ws.get("xcol", Epad * 4 * 9 * 512, torch.bfloat16)
            self._timed("conv3_dgrad_windows", lambda: _lib.check(lib.sgc_windows_dgrad_cols(_lib.ptr(dy3x), _lib.ptr(w["w3col"]), _lib.ptr(col), Epad * 4, st()),
                                                                  "sgc_windows_dgrad_cols"))
            if TUNING.gemms_apart:
                wgrad_windows()                      # side stream: after the data-gradient GEMM, beside col2im / the contraction
            self._timed("col2im_windows", lambda: (
                _lib.check(lib.sgc_windows_col2im(_lib.ptr(col), _lib.ptr(ctx.bbox), _lib.ptr(ctx.sub_idx), _lib.ptr(ctx.obj_idx), _lib.ptr(sh["incl"]),
                                                  P, _lib.ptr(dz), st()), "sgc_windows_col2im"),
                _lib.check(lib.sgc_windows_col2im_objects(_lib.ptr(col), _lib.ptr(ctx.bbox), n_obj, P, _lib.ptr(sh["incl"]), _lib.ptr(dz), st()),
                           "sgc_windows_col2im_objects") if objects else None))
        elif TUNING.gemms_apart:
            wgrad_windows()
        return dz

    # ------------------------------------------------------------------ two-stream backward
    def _side_chain(self):
        """Callable context manager that runs its body on this device's side stream, ordered after everything enqueued on the
        caller's stream so far; ``join()`` orders the caller's stream after the side stream.  Both are no-ops with
        ``TUNING.bwd_streams`` off."""
        import contextlib
        eng = self
        enabled = TUNING.bwd_streams

        class Chain:
            def __init__(self):
                self.main = torch.cuda.current_stream(eng.device)
                if enabled:
                    if getattr(eng, "_side_stream", None) is None:       # one per engine: image groups on concurrent lanes keep apart
                        eng._side_stream = torch.cuda.Stream(device=eng.device)
                    self.side = eng._side_stream
                    ev = torch.cuda.Event()
                    ev.record(self.main)
                    self.side.wait_event(ev)         # the side stream's previous work may not overtake buffers reused by this step

            @contextlib.contextmanager
            def __call__(self):
                if not enabled:
                    yield
                    return
                ev = torch.cuda.Event()
                ev.record(self.main)
                self.side.wait_event(ev)
                with torch.cuda.stream(self.side):
                    yield

            def join(self):
                if enabled:
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                    self.main.wait_event(ev)
        return Chain()


    def commonsense_coefficients(self, cand_pred: torch.Tensor, bitmaps, step: torch.Tensor, n_steps: int, scat: torch.Tensor,
                                 ocat: torch.Tensor, lambda_commonsense=1.0, lambda_weak=0.1, lambda_strong=10.0):
        """Per-candidate coefficients of the train_cs penalty (train_utils.py:36-62 + the running-sum step weights of
        train_test.py:219-233): kappa = (T - t) * lambda_cs * (lambda_weak * weak / #weak_t + lambda_strong * strong / #strong_t).
        ``cand_pred`` [P, n_cand] int32: the forward's per-super-category predicates of ALL pairs of the minibatch (the per-step
        counts couple its images; image groups hand over the gathered rows)."""
        P = int(cand_pred.shape[0])
        nc = int(cand_pred.shape[1])
        cand_pred = cand_pred.contiguous()
        weak = torch.empty(P, nc, dtype=torch.float32, device=self.device)
        strong = torch.empty(P, nc, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.sgc_commonsense_flags(_lib.ptr(scat), _lib.ptr(ocat), _lib.ptr(cand_pred), P, nc,
                                                  _lib.ptr(bitmaps.aligned), _lib.ptr(bitmaps.violated), bitmaps.C, bitmaps.R,
                                                  _lib.ptr(weak), _lib.ptr(strong), self._st()), "sgc_commonsense_flags")
        cw = torch.zeros(n_steps, device=self.device).index_add_(0, step, weak.sum(1))        # tiny host-side glue
        cs = torch.zeros(n_steps, device=self.device).index_add_(0, step, strong.sum(1))
        wt = (n_steps - step).float()[:, None] * lambda_commonsense
        kap = wt * (lambda_weak * weak / cw.clamp(min=1)[step][:, None] + lambda_strong * strong / cs.clamp(min=1)[step][:, None])
        return kap.contiguous()

    def supcon_loss(self, feats: torch.Tensor, labels: torch.Tensor, grad_scale: float = 1.0, temperature: float = 0.07):
        """SupConLossHierar on feats [2M,512] f32 (view 0 rows then view 1 rows), labels [M] int32.
        Returns (loss scalar tensor, dF [2M,512] = grad_scale * dloss/dfeats)."""
        M = int(labels.shape[0])
        n = 2 * M
        G = self.ws.get("supcon_G", n * n, torch.float32)
        rows = torch.empty(n, dtype=torch.float32, device=self.device)
        dF = torch.empty(n, 512, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.sgc_supcon_hierar(_lib.ptr(feats), _lib.ptr(labels), M, ctypes.c_float(temperature), ctypes.c_float(grad_scale),
                                              _lib.ptr(G), _lib.ptr(rows), _lib.ptr(dF), self._st()), "sgc_supcon_hierar")
        return rows.sum() / n, dF
