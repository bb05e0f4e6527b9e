"""Host orchestration of the fused pairwise relation path on one MI355X.

All compute is in the HIP kernels of ``csrc/`` reached through the C-ABI of ``include/sgc_relhead.h``;
torch supplies device memory, the current HIP stream and (elsewhere) torch.distributed.  The restructuring
identities are SURVEY §7.1:

  conv1 is 1x1 and the bbox mask is per pixel   -> conv1 runs once per image and role (``sgc_conv1_tanh``)
  conv2 is linear over a channel concat          -> U_i + V_j from per-object convs (``sgc_conv2_object``)
  everything after that ReLU is per pair         -> expansion, conv3, fc1, fc2, head over all pairs at once
  one-hot label concat                           -> per-object 512-vectors gathered in fc2's epilogue

Layout of the engine (round 6; one 1 800-line module before): ``engine_core.py`` (containers, ``Tuning``), ``engine_weights.py`` (16-bit weight
layouts), ``engine_plan.py`` (shared-window planning), ``engine_fwd.py`` (forward stages), ``engine_bwd.py`` (backward) - mixins of the ONE class
below; this module keeps the class itself (construction, deferred checks, timers, loss-side kernels) and re-exports every shared name.
"""
from __future__ import annotations

from .engine_core import *          # noqa: F401,F403
from .engine_core import __all__ as _core_all
from .engine_bwd import BackwardMixin
from .engine_fwd import ForwardMixin
from .engine_plan import PlanMixin
from .engine_weights import WeightsMixin


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


class RelHeadEngine(WeightsMixin, PlanMixin, ForwardMixin, BackwardMixin):
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
