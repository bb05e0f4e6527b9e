"""Drop-in modules for the reference's ``model.py`` backed by the gfx950 kernels.

Same class names, constructor arguments, parameter names/shapes (reference checkpoints load with
``load_state_dict``, including the DDP ``module.`` prefix via ``strip_ddp_prefix``) and ``forward`` signatures
and return tuples as the reference (``model.py:9-34, 37-102, 105-186``).  Two entry points:

* ``forward(h_sub, h_obj, c1, c2, s1, s2, rank, h_sub_aug=None, h_obj_aug=None)`` - the reference's per-step
  call on pre-masked ``[b,257,32,32]`` inputs, so the reference's own pair loops run unchanged;
* ``forward_pairs(...)`` / ``training_step(...)`` - the fused path: one call per minibatch over all ordered
  pairs, in the reference's candidate order.

There is no CPU or eager-PyTorch fallback: the HIP extension must load and the module must live on a GPU.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from .engine import PairOutputs, RelHeadEngine, make_engine
from .pairs import DeviceScene, pair_targets_fast, super_multihot
from .synthetic import HeadConfig, predicate_counts


def _shared_hint(scene):
    """What the host already knows about the scene's pair-specific windows (count, per-window counts) - saves a device read-back."""
    n = getattr(scene, "shared_windows", None)
    return None if n is None else dict(windows=n, per_window=getattr(scene, "window_entries", None),
                                       object_windows=getattr(scene, "object_windows", None),
                                       linear_windows=getattr(scene, "linear_windows", None),
                                       conv2_windows=getattr(scene, "conv2_windows", None))


def _dense(scene):
    """(img_ptr, pid, max_n) when the scene carries the all-pairs lookup table (flatten_scene builds it)."""
    if getattr(scene, "pid", None) is None or scene.max_n <= 0:
        return None
    return (scene.img_ptr, scene.pid, scene.max_n)


def _csr_by_device(index: torch.Tensor, n: int):
    """Device form of ``engine.csr_by``: (ptr [n+1], list) int32 of the positions grouped by object id, stable (sums run in list order)."""
    idx = index.long()
    order = torch.argsort(idx, stable=True)
    ptr = torch.searchsorted(idx[order].contiguous(), torch.arange(n + 1, device=index.device))
    return ptr.to(torch.int32).contiguous(), order.to(torch.int32).contiguous()


def strip_ddp_prefix(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Reference checkpoints are saved from the DDP wrapper (``train_test.py:322``; ``utils.py:207-214``)."""
    return {k.replace("module.", ""): v for k, v in state_dict.items()}


class _PairStepFunction(torch.autograd.Function):
    """The reference's per-step classifier call (``train_utils.py:26-27``: b pre-masked subject / object crops) as ONE autograd
    node: forward = the HIP trunk + head on a child engine that keeps its context alive, backward = the HIP backward driven by
    the gradients of the outputs (``sgc_head_bwd_upstream``), so the reference's own ``training()`` - hundreds of calls, then
    one ``losses.backward()`` (``train_test.py:189-276``) - trains the f32 master parameters.  The inputs get no gradient (the
    reference computes them under ``no_grad``, ``train_test.py:154-156``)."""

    @staticmethod
    def forward(ctx, module, h_sub, h_obj, c1, c2, s1, s2, seeds, *params):
        cfg = module.head_config()
        eng = module.refresh_weights(backward=True).child()
        dev = eng.device
        b = int(h_sub.shape[0])
        hs = h_sub.detach().to(dev, torch.float32).contiguous()
        ho = h_obj.detach().to(dev, torch.float32).contiguous()
        full = torch.tensor([[0, cfg.feature_size, 0, cfg.feature_size]], dtype=torch.int32, device=dev).repeat(b, 1).contiguous()
        ids = torch.arange(b, dtype=torch.int32, device=dev)
        mh1 = mh2 = None
        if s1 is not None and cfg.dataset == "vg":
            mh1 = torch.from_numpy(super_multihot([list(s1)], cfg.num_super_classes)).to(dev)
            mh2 = torch.from_numpy(super_multihot([list(s2)], cfg.num_super_classes)).to(dev)
        tctx = eng.train_forward(None, None, ids, full, c1.to(dev).long().contiguous(), mh1, ids, ids, seeds=seeds, dropout=module.training,
                                 role_inputs=(hs, ho), cats_obj=c2.to(dev).long().contiguous(), super_mh_obj=mh2)
        ctx.eng, ctx.tctx, ctx.module = eng, tctx, module
        ctx.names = [n for n, _ in module.named_parameters()]
        ctx.csr = (torch.arange(b + 1, dtype=torch.int32, device=dev), ids)
        ctx.img_ptr = torch.arange(b + 1, dtype=torch.int32, device=dev)
        ctx.weights_version = module._weights_version
        out = tctx.out
        sup = out.super_relation if out.super_relation is not None else torch.zeros(b, 0, device=dev)
        return out.relation, sup, out.connectivity.view(-1, 1), out.hidden.clone()

    @staticmethod
    def backward(ctx, g_rel, g_sup, g_conn, g_hidden):
        module = ctx.module
        if module._weights_version is not ctx.weights_version and module._weights_version != ctx.weights_version:
            raise RuntimeError("parameters changed between the per-step forward and its backward (optimizer.step() before "
                               "losses.backward()?): the 16-bit weight copies of the forward are gone")
        hier = module.hierarchical
        _, grads = ctx.eng.train_backward(ctx.tctx, None, ctx.csr, ctx.csr, ctx.img_ptr,
                                          upstream=(g_rel, g_sup if hier else None, g_conn.reshape(-1), g_hidden))
        ctx.eng = ctx.tctx = None                                  # free the step's buffers
        named = dict(module.named_parameters())
        return (None,) * 8 + tuple(grads[n].view_as(named[n]) for n in ctx.names)


class _RelationBase(nn.Module):
    hierarchical = True

    def _build_trunk(self, args, input_dim, feature_size, num_classes, num_super_classes):
        self.input_dim = input_dim
        self.feature_size = feature_size
        self.num_classes = num_classes
        self.num_super_classes = num_super_classes
        self.dataset = args["dataset"]["dataset"]
        self.conv1_1 = nn.Conv2d(2 * input_dim + 1, input_dim, kernel_size=1, stride=1, padding=0)
        self.conv1_2 = nn.Conv2d(2 * input_dim + 1, input_dim, kernel_size=1, stride=1, padding=0)
        self.conv2_1 = nn.Conv2d(2 * input_dim, 4 * input_dim, kernel_size=3, stride=1, padding=1)
        self.conv3_1 = nn.Conv2d(4 * input_dim, 8 * input_dim, kernel_size=3, stride=1, padding=1)
        self.dropout1 = nn.Dropout(p=0.5)     # kept for state/API parity; dropout itself runs in the kernels
        self.dropout2 = nn.Dropout(p=0.5)
        self.maxpool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.fc1 = nn.Linear(8 * input_dim * (feature_size // 4) ** 2, 4096)
        if self.dataset == "vg":
            self.fc2 = nn.Linear(4096 + 2 * (num_classes + num_super_classes), 512)
        else:
            self.fc2 = nn.Linear(4096 + 2 * num_classes, 512)
        self._engine: Optional[RelHeadEngine] = None
        self._weights_version = None
        self._step = 0
        self.dropout_seed = 0x5EED

    # ------------------------------------------------------------------ engine plumbing
    def head_config(self) -> HeadConfig:
        raise NotImplementedError

    def engine(self) -> RelHeadEngine:
        dev = self.fc1.weight.device
        if dev.type != "cuda":
            raise RuntimeError("the relation head runs only on a GPU through its HIP kernels (no CPU fallback); "
                               "move the module to a cuda device")
        if self._engine is None or self._engine.device != dev:
            self._engine = make_engine(self.head_config(), dev)
            self._engine.T = (float(getattr(self, "T1", 1)), float(getattr(self, "T2", 1)), float(getattr(self, "T3", 1)))
            self._weights_version = None
        return self._engine

    def _version(self):
        return tuple(p._version for p in self.parameters()) + tuple(p.data_ptr() for p in self.parameters())

    def refresh_weights(self, backward: bool = False):
        """Re-derive the 16-bit compute copies when the f32 master parameters changed."""
        eng = self.engine()
        v = (self._version(), backward)
        if self._weights_version is None or self._weights_version[0] != v[0] or (backward and not self._weights_version[1]):
            sd = {k: p for k, p in self.named_parameters()}
            sync = self.__dict__.get("weight_sync")      # distributed.ShardedSGD.attach: fc1.weight's all-gather may still be in flight
            with torch.no_grad():
                eng.load_weights(sd, fc1_sync=sync)
                if backward:
                    eng.prep_bwd_weights(sd, fc1_sync=sync)
            self._weights_version = v
        return eng

    def lane_engine(self, k: int) -> RelHeadEngine:
        """Engine ``k`` of this module: 0 is the module's own, the others share its weights (the 16-bit copies are re-derived once,
        by ``refresh_weights``) and own their workspace, so that image groups can be in flight on different streams."""
        eng = self.refresh_weights(backward=True)
        if k == 0:
            return eng
        for key in ("w1p", "w1pT"):                    # deferred copies: made on the caller's stream, before the lanes' streams read them
            if key in eng.w or key in eng.w.deferred:  # (the generic trunk keeps fc1's copies under other names, made eagerly)
                eng.w[key]
        lanes = self.__dict__.setdefault("_lane_engines", {})
        if k not in lanes or lanes[k].device != eng.device:
            lanes[k] = make_engine(self.head_config(), eng.device)
        lane = lanes[k]
        lane.w, lane.T, lane.head_rows = eng.w, eng.T, eng.head_rows
        return lane

    # ------------------------------------------------------------------ fused path
    def forward_pairs(self, scene: DeviceScene, iou_mask: Optional[torch.Tensor] = None, select: Optional[torch.Tensor] = None) -> PairOutputs:
        """All ordered pairs of a minibatch in one pass (eval numerics unless ``self.training``).  ``select`` [P]: compute only
        these pairs (the others come back with confidence -inf), see ``RelHeadEngine.forward_pairs``."""
        eng = self.refresh_weights()
        seeds = self._next_seeds() if self.training else (0, 0)
        if scene.n_pairs == 0:
            cfg, dev = self.head_config(), eng.device
            nc = 3 if cfg.hierarchical else 1
            z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
            return PairOutputs(z(0, cfg.num_relations), z(0, 3) if cfg.hierarchical else None, z(0), z(0, 512), z(0, nc),
                               z(0, nc, dt=torch.int32))
        with torch.no_grad():
            return eng.forward_pairs(scene.image_feature, scene.image_depth, scene.obj_img, scene.bbox, scene.cats,
                                     scene.super_mh, scene.sub_idx, scene.obj_idx, train=self.training, seeds=seeds,
                                     iou_mask=iou_mask, dense=_dense(scene), select=select,
                                     shared_windows=_shared_hint(scene), pair_order=getattr(scene, "sub_list", None))

    def _next_seeds(self):
        self._step += 1
        return ((self.dropout_seed * 2654435761 + 2 * self._step) & 0xFFFFFFFF,
                (self.dropout_seed * 2654435761 + 2 * self._step + 1) & 0xFFFFFFFF)

    def _class_weight_device(self, class_weight, dev):
        """[R] f32 device tensor of the class weights of the relation NLL (default 1 - count / total, ``train_test.py:105-117``)."""
        if class_weight is None:
            counts = predicate_counts(self.head_config()).numpy()
            class_weight = 1 - counts / counts.sum()
        cw = np.ascontiguousarray(class_weight, dtype=np.float32)
        if getattr(self, "_cw_cache", None) is None or self._cw_cache[0].shape != cw.shape or not np.array_equal(self._cw_cache[0], cw) \
                or self._cw_cache[1].device != dev:
            self._cw_cache = (cw.copy(), torch.from_numpy(cw.copy()).to(dev))
        return self._cw_cache[1]

    def minibatch_loss_coefficients(self, scene: DeviceScene, class_weight=None, lambda_connectivity: float = 0.1,
                                    lambda_not_connected: float = 1.0):
        """(tgt, a, b, c, y) per ordered pair of ``scene`` (``engine.loss_coefficients``) on the device."""
        eng = self.refresh_weights(backward=True)
        if scene.directed is None:
            raise ValueError("the scene carries no relation targets")
        return eng.loss_coefficients_device(scene.step_ptr, scene.n_steps, scene.directed, self._class_weight_device(class_weight, eng.device),
                                            lambda_connectivity, lambda_not_connected)

    def zero_gradient_step(self, reducer=None):
        """The step of a rank that has NOTHING to score (no image with two objects, or a minibatch whose samples the loader dropped):
        zero loss, zero gradients - but with ``reducer`` over more than one rank it still takes part in this step's collectives, in
        the same order, with zeros: the other ranks reduce their gradients now, and a rank that skipped them would pair its NEXT
        step's all-reduce with the peers' current one (or leave them waiting for ever)."""
        dev = next(self.parameters()).device
        if reducer is not None and reducer.world > 1:
            zeros = {n: torch.zeros_like(p) for n, p in self.named_parameters()}
            if "fc1.weight" in zeros:
                reducer.hook("fc1.weight", zeros["fc1.weight"])
            reducer.finish_grads(zeros)           # "zeros" now holds the mean of the OTHER ranks' gradients: apply it like they do
            if not getattr(reducer, "owns_grads", False):
                for n, p in self.named_parameters():
                    if p.grad is None:
                        p.grad = zeros[n]
                    else:
                        p.grad.add_(zeros[n])
        if not getattr(reducer, "owns_grads", False):
            for p in self.parameters():
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
        self.last_outputs = None
        self.last_connectivity_stats = None
        return torch.zeros((), device=dev)

    def training_step(self, scene: DeviceScene, relationships=None, subj_or_obj=None, directed: Optional[np.ndarray] = None,
                      lambda_connectivity: float = 0.1, lambda_not_connected: float = 1.0, class_weight=None,
                      grad_hook=None, reducer=None, image_feature_aug: Optional[torch.Tensor] = None, lambda_contrast: float = 1.0,
                      commonsense=None, lambda_commonsense: float = 1.0, lambda_cs_weak: float = 0.1,
                      lambda_cs_strong: float = 10.0, loss_coefs=None, grads_out: Optional[Dict[str, torch.Tensor]] = None,
                      engine: Optional[RelHeadEngine] = None, coupled: Optional[dict] = None):
        """Forward + loss + backward over all ordered pairs; gradients land in ``param.grad`` (accumulating like
        autograd).  Loss follows ``train_test.py:189-258`` / ``train_utils.py:64-157`` (hierarchical NLL, BCE on
        connectivity, running-sum step weights).  With ``image_feature_aug`` (DETR features of the colour-jittered view,
        ``train_test.py:154``) the supervised-contrastive term of ``train_test.py:260-273`` is added: the augmented trunk
        is run ONLY for the connected pairs (the only ones the loss reads; the reference runs it for every pair).
        ``commonsense=(aligned_keys, violated_keys)`` adds the train_cs penalty of ``train_utils.py:36-62``.
        ``reducer`` (``distributed.GradReducer``): the step's gradients are mean-reduced across ranks before they are
        accumulated into ``param.grad`` (the fc1 weight gradient is handed to RCCL as soon as it is enqueued).
        ``loss_coefs`` / ``grads_out`` serve the image-group chunking of ``pair_loop.train_minibatch``: the per-pair loss
        coefficients of THIS scene's pairs taken from the whole minibatch's (per-step means and running-sum weights couple the
        images of a minibatch, the gradients are additive over images once the coefficients are fixed), and a dict the step's
        gradients are summed into instead of ``param.grad`` (no reduction here: the caller reduces the sum once).  ``engine``: run on
        this engine (``lane_engine``: same weights, own workspace) instead of the module's - image groups on concurrent streams.
        ``coupled`` (image groups with the contrastive / commonsense terms, which couple ALL pairs of a minibatch through the
        forward - ``pair_loop._train_image_groups`` runs every group's forward first, ``coupled_forward``, then the minibatch-level
        pieces, then this step per group): dict(seeds=(main, aug) - the dropout seeds of the group's first forward, so that this
        pass reproduces it bit for bit -, cs_coef [P, n_cand] or None - this group's rows of the minibatch's commonsense
        coefficients -, contrast=(dF_main [M, 512], dF_aug [M, 512]) or None - this group's rows of the SupCon feature gradient,
        M = its connected pairs in pair order).  The loss returned then lacks the contrastive term (the caller adds it once)."""
        # a reducer that is also the optimizer (distributed.ShardedSGD after ``attach``) may want fc1.weight's gradient in GEMM order on
        # EVERY path that feeds it - also these direct calls (gradient accumulation: zero_grad, several training_step, step)
        eng = engine if engine is not None else self.refresh_weights(backward=True)
        if bool(getattr(reducer, "fc1_gemm_order", False)) and not eng.fc1_grad_gemm_order:
            eng.fc1_grad_gemm_order = True
            try:
                return self.training_step(scene, relationships, subj_or_obj, directed, lambda_connectivity, lambda_not_connected, class_weight,
                                          grad_hook, reducer, image_feature_aug, lambda_contrast, commonsense, lambda_commonsense,
                                          lambda_cs_weak, lambda_cs_strong, loss_coefs, grads_out, engine, coupled)
            finally:
                eng.fc1_grad_gemm_order = False
        if reducer is not None:
            grad_hook = reducer.hook
        cfg = self.head_config()
        dev = eng.device
        P = scene.n_pairs
        if P == 0 and grads_out is not None:       # an image group without pairs adds nothing to the minibatch's gradient sum
            self.last_outputs = None
            self.last_connectivity_stats = None
            return torch.zeros((), device=dev)
        if P == 0:                                 # no image with two objects: nothing to score, zero loss and gradients
            return self.zero_gradient_step(reducer)
        # directed target per pair on the device: an explicit array wins, else what flatten_scene derived from the batch's
        # relationships / subj_or_obj lists (sgc_scene_tables), else derive it from the lists given here
        if torch.is_tensor(directed):
            directed_d = directed.to(dev, torch.int32).contiguous()
        elif directed is not None:
            directed_d = torch.from_numpy(np.ascontiguousarray(directed, dtype=np.int32)).to(dev)
        elif scene.directed is not None and (relationships is None or relationships is scene._rel_src):
            directed_d = scene.directed
        elif relationships is not None:            # lists other than the ones the scene was flattened from
            directed_d = torch.from_numpy(pair_targets_fast(relationships, subj_or_obj, scene.pidx).astype(np.int32)).to(dev)
        else:
            raise ValueError("training_step needs relation targets: build the scene from a batch with relationships / subj_or_obj, "
                             "or pass them (or a directed array) here")
        cw_d = self._class_weight_device(class_weight, dev)
        if loss_coefs is not None:
            if (image_feature_aug is not None or commonsense is not None) and coupled is None:
                raise ValueError("externally supplied loss coefficients (image-group chunking) with the contrastive / commonsense terms "
                                 "need the minibatch-level pieces too: pass coupled=... (pair_loop._train_image_groups does)")
            coefs_d = tuple(c.contiguous() for c in loss_coefs)
        else:
            coefs_d = eng.loss_coefficients_device(scene.step_ptr, scene.n_steps, directed_d, cw_d, lambda_connectivity,
                                                   lambda_not_connected)
        sub_csr, obj_csr, img_ptr = scene.sub_csr, scene.obj_csr, scene.img_ptr
        with torch.no_grad():
            seeds_main = coupled["seeds"][0] if coupled is not None else (self._next_seeds() if self.training else (0, 0))
            ctx = eng.train_forward(scene.image_feature, scene.image_depth, scene.obj_img, scene.bbox, scene.cats,
                                    scene.super_mh, scene.sub_idx, scene.obj_idx,
                                    seeds=seeds_main, dropout=self.training,
                                    dense=_dense(scene), shared_windows=_shared_hint(scene), pair_order=getattr(scene, "sub_list", None))
            cs_coef = None
            if coupled is not None:
                cs_coef = None if coupled.get("cs_coef") is None else coupled["cs_coef"].contiguous()
            elif commonsense is not None:
                cs_coef = eng.commonsense_coefficients(ctx.out.cand_pred, self._commonsense_bitmaps(commonsense, dev), scene.step.long(),
                                                       scene.n_steps, scene.cats[scene.sub_idx.long()], scene.cats[scene.obj_idx.long()],
                                                       lambda_commonsense, lambda_cs_weak, lambda_cs_strong)
            dp_main = None
            extra = None
            loss_c = None
            if image_feature_aug is not None:
                # connected pairs selected on the device; the only host synchronisation is their count (the augmented trunk's sizes)
                conn_idx = torch.nonzero(directed_d >= 0).flatten()
            if image_feature_aug is not None and int(conn_idx.numel()) > 0:
                extra = self._contrast_trunk(eng, scene, image_feature_aug, conn_idx,
                                             coupled["seeds"][1] if coupled is not None else (self._next_seeds() if self.training else (0, 0)))
                M = extra["M"]
                if coupled is not None:                    # the minibatch's SupCon gradient rows of this group's connected pairs
                    dF_main, extra["dp_aug"] = (t.contiguous() for t in coupled["contrast"])
                    assert int(dF_main.shape[0]) == M and int(extra["dp_aug"].shape[0]) == M
                else:
                    feats = torch.cat((ctx.p[:P * 512].view(P, 512)[conn_idx], extra["ctx"].p[:M * 512].view(M, 512)), dim=0).contiguous()
                    lam2 = float(lambda_contrast) * float(lambda_contrast)
                    loss_c, dF = eng.supcon_loss(feats, directed_d[conn_idx].to(torch.int32).contiguous(), grad_scale=lam2)
                    dF_main, extra["dp_aug"] = dF[:M], dF[M:].contiguous()
                dp_main = torch.zeros(P, 512, dtype=torch.float32, device=dev)
                dp_main[conn_idx] = dF_main
            loss, grads = eng.train_backward(ctx, coefs_d, sub_csr, obj_csr, img_ptr,
                                             grad_hook=grad_hook if extra is None else None, dp_extra=dp_main,
                                             cs_coef=cs_coef)
            if extra is not None:
                eng_a = extra["engine"]
                _, grads_a = eng_a.train_backward(extra["ctx"], extra["coefs"], extra["sub_csr"], extra["obj_csr"], img_ptr,
                                                  dp_extra=extra["dp_aug"])
                for k in grads:
                    grads[k] = grads[k] + grads_a[k].view_as(grads[k])
                if grad_hook is not None:
                    grad_hook("fc1.weight", grads["fc1.weight"])
                if loss_c is not None:
                    if not bool(torch.isnan(loss_c)):
                        loss = loss + lambda_contrast * lambda_contrast * loss_c      # lambda applied twice (train_test.py:270-273)
                    self.last_contrast_loss = loss_c
            if grads_out is not None:
                for name, p in self.named_parameters():
                    g = grads[name].view_as(p)
                    if name in grads_out:
                        grads_out[name].add_(g)
                    else:
                        grads_out[name] = g.clone()          # the engine reuses its gradient buffers in the next group's pass
            else:
                if reducer is not None:
                    reducer.finish_grads(grads)
                if not getattr(reducer, "owns_grads", False):      # a sharded reducer keeps the mean gradient shards itself
                    self.accumulate_grads(grads, eng)
            # connectivity statistics of train_one_direction (train_utils.py:66-87) summed over the minibatch: a [5] device
            # tensor (not connected, connected, predicted connected, precision numerator, recall numerator), no host sync
            raw_d = scene.raw_target if (directed is None and scene.raw_target is not None) else directed_d
            self.last_connectivity_stats = eng.connectivity_stats(ctx.out.connectivity, directed_d, raw_d)
        self.last_outputs = ctx.out
        return loss

    def accumulate_grads(self, grads, eng):
        """``param.grad`` += the step's gradients (like autograd).  fc1.weight's carries its column order: the reference's, or the GEMM's
        (only inside ``pair_loop.train_minibatch`` with an optimizer that consumes it); never mixed within one accumulation."""
        gemm_order = bool(getattr(eng, "fc1_grad_gemm_order", False))
        for name, p in self.named_parameters():
            g = grads[name].view_as(p)
            if name == "fc1.weight":
                if p.grad is not None and bool(getattr(p, "_sgc_grad_gemm_order", False)) != gemm_order:
                    raise RuntimeError("fc1.weight.grad is being accumulated in two different column orders")
                p._sgc_grad_gemm_order = gemm_order
            if p.grad is None:
                p.grad = g if g.is_contiguous() else g.contiguous()
            else:
                p.grad.add_(g)

    def _commonsense_bitmaps(self, commonsense, dev):
        from .commonsense import TripletBitmaps
        cfg = self.head_config()
        if getattr(self, "_cs_bitmaps", None) is None or self._cs_bitmaps[0] is not commonsense:
            self._cs_bitmaps = (commonsense, TripletBitmaps(commonsense[0], commonsense[1], cfg.num_classes, cfg.num_relations, dev))
        return self._cs_bitmaps[1]

    def _contrast_trunk(self, eng, scene, image_feature_aug, conn_idx, seeds):
        """Augmented-view trunk (``train_test.py:154,196,204``) for the connected pairs only: its context, pair lists and the zero
        loss coefficients of its backward pass (the SupCon gradient enters through ``dp_extra``)."""
        dev = eng.device
        if getattr(self, "_engine_aug", None) is None or self._engine_aug.device != dev:
            self._engine_aug = make_engine(self.head_config(), dev)
        eng_a = self._engine_aug
        eng_a.w, eng_a.T, eng_a.head_rows = eng.w, eng.T, eng.head_rows          # shared weights, own workspace
        M = int(conn_idx.numel())
        sub_a = scene.sub_idx[conn_idx].contiguous()
        obj_a = scene.obj_idx[conn_idx].contiguous()
        ctx_a = eng_a.train_forward(image_feature_aug.to(dev, torch.float32).contiguous(), scene.image_depth, scene.obj_img,
                                    scene.bbox, scene.cats, scene.super_mh, sub_a, obj_a, seeds=seeds, dropout=self.training)
        n_obj = int(scene.obj_img.shape[0])
        zeros = torch.zeros(M, dtype=torch.float32, device=dev)
        coefs = (torch.full((M,), -1, dtype=torch.int32, device=dev), zeros, zeros, zeros, zeros)
        return dict(engine=eng_a, ctx=ctx_a, M=M, coefs=coefs, sub_csr=_csr_by_device(sub_a, n_obj), obj_csr=_csr_by_device(obj_a, n_obj))

    def coupled_forward(self, scene: DeviceScene, image_feature_aug: Optional[torch.Tensor] = None, want_candidates: bool = False,
                        engine: Optional[RelHeadEngine] = None):
        """First pass of an image group whose minibatch carries the contrastive / commonsense terms (``pair_loop._train_image_groups``):
        the training-mode forward only.  Returns dict(seeds=(main, aug) - to be handed back to ``training_step(coupled=...)`` so that
        the second pass draws the same dropout masks -, cand_pred [P, n_cand] (with ``want_candidates``: what the commonsense flags
        read), conn_idx [M] connected pairs in pair order, hidden / hidden_aug [M, 512]: the rows SupConLossHierar reads)."""
        eng = engine if engine is not None else self.refresh_weights(backward=True)
        if scene.directed is None:
            raise ValueError("the scene carries no relation targets")
        P = scene.n_pairs
        seeds = (self._next_seeds() if self.training else (0, 0), self._next_seeds() if (self.training and image_feature_aug is not None) else (0, 0))
        out = dict(seeds=seeds, cand_pred=None, conn_idx=None, hidden=None, hidden_aug=None)
        if P == 0:
            return out
        with torch.no_grad():
            ctx = eng.train_forward(scene.image_feature, scene.image_depth, scene.obj_img, scene.bbox, scene.cats, scene.super_mh,
                                    scene.sub_idx, scene.obj_idx, seeds=seeds[0], dropout=self.training, dense=_dense(scene),
                                    shared_windows=_shared_hint(scene), pair_order=getattr(scene, "sub_list", None))
            if want_candidates:
                out["cand_pred"] = ctx.out.cand_pred.clone()
            if image_feature_aug is not None:
                conn_idx = torch.nonzero(scene.directed >= 0).flatten()
                M = int(conn_idx.numel())
                out["conn_idx"] = conn_idx
                if M > 0:
                    out["hidden"] = ctx.p[:P * 512].view(P, 512)[conn_idx].clone()
                    ctx_a = self._contrast_trunk(eng, scene, image_feature_aug, conn_idx, seeds[1])["ctx"]
                    out["hidden_aug"] = ctx_a.p[:M * 512].view(M, 512).clone()
        return out

    # ------------------------------------------------------------------ reference per-step call
    def _step_call(self, h_sub, h_obj, c1, c2, s1, s2):
        """(relation [b,R], super [b,3] | None, connectivity [b,1], hidden [b,512]) of one per-step call.  With autograd on
        and trainable parameters the call is an autograd node (``_PairStepFunction``); otherwise the inference trunk."""
        params = [p for _, p in self.named_parameters()]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            seeds = self._next_seeds() if self.training else (0, 0)
            rel, sup, conn, pred = _PairStepFunction.apply(self, h_sub, h_obj, c1, c2, s1, s2, seeds, *params)
            return rel, (sup if self.hierarchical else None), conn, pred
        out = self._compat_forward(h_sub, h_obj, c1, c2, s1, s2)
        # ``hidden`` is a view of the engine workspace: own it before the trunk runs again (augmented view, next step)
        return out.relation, out.super_relation, out.connectivity.view(-1, 1), out.hidden.clone()

    def _compat_forward(self, h_sub, h_obj, c1, c2, s1, s2):
        """Reference semantics for arbitrary pre-masked inputs: each of the b rows is one (subject, object) pair."""
        cfg = self.head_config()
        eng = self.refresh_weights()
        dev = eng.device
        b = int(h_sub.shape[0])
        with torch.no_grad():
            hs = h_sub.to(dev, torch.float32).contiguous()
            ho = h_obj.to(dev, torch.float32).contiguous()
            mh1 = mh2 = None
            if s1 is not None:
                mh1 = torch.from_numpy(super_multihot([list(s1)], cfg.num_super_classes)).to(dev)
                mh2 = torch.from_numpy(super_multihot([list(s2)], cfg.num_super_classes)).to(dev)
            seeds = self._next_seeds() if self.training else (0, 0)
            return eng.compat_forward(hs, ho, c1.to(dev).long(), c2.to(dev).long(), mh1, mh2, train=self.training, seeds=seeds)


class BayesianRelationClassifier(_RelationBase):
    """Hierarchical (Bayesian) local predictor - drop-in for reference ``model.py:105-186``."""
    hierarchical = True

    def __init__(self, args, input_dim=128, feature_size=32, num_classes=150, num_super_classes=17, num_geometric=15,
                 num_possessive=11, num_semantic=24, T1=1, T2=1, T3=1):
        super().__init__()
        self._build_trunk(args, input_dim, feature_size, num_classes, num_super_classes)
        self.num_geometric, self.num_possessive, self.num_semantic = num_geometric, num_possessive, num_semantic
        self.fc3_1 = nn.Linear(512, num_geometric)
        self.fc3_2 = nn.Linear(512, num_possessive)
        self.fc3_3 = nn.Linear(512, num_semantic)
        self.fc4 = nn.Linear(512, 1)
        self.fc5 = nn.Linear(512, 3)
        self.T1, self.T2, self.T3 = T1, T2, T3

    def head_config(self) -> HeadConfig:
        return HeadConfig(dataset=self.dataset, hidden_dim=self.input_dim, feature_size=self.feature_size,
                          num_classes=self.num_classes, num_super_classes=self.num_super_classes,
                          num_geometric=self.num_geometric, num_possessive=self.num_possessive,
                          num_semantic=self.num_semantic, hierarchical=True)

    def forward(self, h_sub, h_obj, c1, c2, s1, s2, rank, h_sub_aug=None, h_obj_aug=None):
        ng, npos = self.num_geometric, self.num_possessive
        rel, sup, conn, pred = self._step_call(h_sub, h_obj, c1, c2, s1, s2)
        pred_aug = None
        if h_sub_aug is not None:      # second (augmented) view: same trunk, hidden only (model.py:172)
            pred_aug = self._step_call(h_sub_aug, h_obj_aug, c1, c2, s1, s2)[3]
        return (rel[:, :ng], rel[:, ng:ng + npos], rel[:, ng + npos:], sup, conn, pred, pred_aug)


class FlatRelationClassifier(_RelationBase):
    """Flat-classification ablation - drop-in for reference ``model.py:37-102``."""
    hierarchical = False

    def __init__(self, args, input_dim=128, output_dim=50, feature_size=32, num_classes=150, num_super_classes=17):
        super().__init__()
        self._build_trunk(args, input_dim, feature_size, num_classes, num_super_classes)
        self.output_dim = output_dim
        self.fc3 = nn.Linear(512, output_dim)
        self.fc4 = nn.Linear(512, 1)

    def head_config(self) -> HeadConfig:
        return HeadConfig(dataset=self.dataset, hidden_dim=self.input_dim, feature_size=self.feature_size,
                          num_classes=self.num_classes, num_super_classes=self.num_super_classes,
                          num_geometric=self.output_dim, num_possessive=0, num_semantic=0, hierarchical=False)

    def forward(self, h_sub, h_obj, c1, c2, s1, s2, rank, h_sub_aug=None, h_obj_aug=None, one_hot=True):
        rel, _, conn, pred = self._step_call(h_sub, h_obj, c1, c2, s1, s2)
        pred_aug = None
        if h_sub_aug is not None:
            pred_aug = self._step_call(h_sub_aug, h_obj_aug, c1, c2, s1, s2)[3]
        return rel, conn, pred, pred_aug


class BayesianHead(nn.Module):
    """Plug-and-play hierarchical head on ``[M,512]`` features - drop-in for reference ``model.py:9-34``."""

    def __init__(self, input_dim=512, num_geometric=15, num_possessive=11, num_semantic=24, T1=1, T2=1, T3=1):
        super().__init__()
        if input_dim != 512:
            raise NotImplementedError("the head kernel is specialised to 512 input features")
        self.fc3_1 = nn.Linear(input_dim, num_geometric)
        self.fc3_2 = nn.Linear(input_dim, num_possessive)
        self.fc3_3 = nn.Linear(input_dim, num_semantic)
        self.fc5 = nn.Linear(input_dim, 3)
        self.T1, self.T2, self.T3 = T1, T2, T3
        self.ng, self.np_, self.ns = num_geometric, num_possessive, num_semantic
        self._eng = None

    def forward(self, h):
        dev = h.device
        if dev.type != "cuda":
            raise RuntimeError("BayesianHead runs only on a GPU through its HIP kernel (no CPU fallback)")
        cfg = HeadConfig(num_geometric=self.ng, num_possessive=self.np_, num_semantic=self.ns)
        if self._eng is None or self._eng.device != dev:
            self._eng = RelHeadEngine(cfg, dev)
        eng = self._eng
        eng.T = (float(self.T1), float(self.T2), float(self.T3))
        with torch.no_grad():
            rows = torch.cat([self.fc3_1.weight, self.fc3_2.weight, self.fc3_3.weight, self.fc5.weight])
            Wc = torch.zeros(64, 512, device=dev)
            Wc[:rows.shape[0]] = rows
            bc = torch.zeros(64, device=dev)
            bc[:rows.shape[0]] = torch.cat([self.fc3_1.bias, self.fc3_2.bias, self.fc3_3.bias, self.fc5.bias])
            eng.w["head_wt"], eng.w["head_b"] = Wc.t().contiguous(), bc
            out = eng.head(h.float().contiguous().view(-1), int(h.shape[0]))
        rel = out.relation
        return rel[:, :self.ng], rel[:, self.ng:self.ng + self.np_], rel[:, self.ng + self.np_:], out.super_relation
