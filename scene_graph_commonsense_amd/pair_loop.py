"""Fused replacements for the reference's pair loops (the bodies of ``testing()`` / ``eval_pc`` and ``training()``).

``evaluate_minibatch`` is what ``train_test.py:373-437`` + ``train_utils.py:160-196`` do for one minibatch: score
every ordered pair, apply the overlap filter (a step in which no image's two boxes overlap is skipped entirely -
no candidates and no targets), and feed the Recall@K evaluators in the reference's candidate order.
``train_minibatch`` is ``train_test.py:174-277`` (pass ``image_feature_aug`` for the contrastive term, ``commonsense`` for train_cs).
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from . import _lib
from .pairs import DeviceScene, flatten_scene, match_target_sgd, pair_targets_fast


# ------------------------------------------------------------------------------------------------ image-group chunking
# A minibatch is scored in ONE fused pass whose workspace grows with its ordered pairs (~2.3 MB per pair in training at
# 11 % pair-specific windows).  The reference's per-step loop handles any object count (``evaluate.py:375-444``: up to 2 x 100
# predicted objects per image, 16 images), so a minibatch that does not fit is cut into consecutive IMAGE GROUPS that run back to
# back: images are independent in the forward, the evaluator is fed in the reference's candidate order across groups from the
# scattered outputs, and in training the per-pair loss coefficients are computed for the WHOLE minibatch first (its per-step means
# and running-sum weights couple the images) so that the groups' gradients simply add (tests/test_configs_gpu.py: additivity).
# Bytes per pair / per pair-specific window / per object, measured with tools/mem_report.py (+15 % margin in ``plan_image_groups``):
_COST = {
    # (per pair, per X window, per pair on the per-pair kernels, per object)
    True: (0.95e6, 0.09e6, 2.7e6, 8.0e6),           # training: z f16 + routing codes + dz (no bf16 copy of the pairs' z: -0.33e6), patch forms of
                                                    # the backward (16 + 20 rows of 1 KB per convolved window; the column forms of rounds 1-2: 0.125e6)
    False: (0.40e6, 0.03e6, 0.60e6, 3.0e6),         # evaluation: z f16, window-major rows, f32 fc1 products
}


def freeze_setup_objects() -> int:
    """Call once after the model, datasets and evaluators are built, before the minibatch loop: collects garbage and moves everything
    alive (``torch``'s and ``numpy``'s module objects, the annotations: 2-3 x 10^5 tracked containers) to the permanent generation.
    Without it CPython's full (generation-2) collection walks all of them every few dozen minibatches - 77 ms measured on the MI355X
    box, a step and a half of the benchmark's 43-46 ms, landing in the middle of an asynchronous launch sequence
    (profiles/r05_host_gc.txt).  The per-minibatch garbage (pair tables, ragged lists) stays collectable.  Returns the number frozen."""
    import gc
    gc.collect()
    gc.freeze()
    return gc.get_freeze_count()


def slice_batch(batch, a: int, b: int):
    """Images [a, b) of a ``SceneBatch`` (views; the per-image lists are shared)."""
    from .synthetic import SceneBatch
    pick = lambda x: None if x is None else list(x[a:b])
    return SceneBatch(batch.image_feature[a:b], batch.image_depth[a:b], pick(batch.bbox), pick(batch.categories),
                      pick(batch.super_categories), pick(getattr(batch, "relationships", None)), pick(getattr(batch, "subj_or_obj", None)),
                      [int(x.shape[0]) for x in batch.bbox[a:b]])


def default_workspace_budget(model) -> int:
    """70 % of what this process could allocate right now: free HBM + the allocator's cached blocks + the engine's own workspace
    (it is reused)."""
    dev = next(model.parameters()).device
    free, _ = torch.cuda.mem_get_info(dev)
    cached = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
    eng = getattr(model, "_engine", None)
    own = eng.ws.nbytes() if eng is not None else 0
    return int(0.7 * (free + cached + own))


def image_workspace_bytes(cfg, bbox, train: bool) -> float:
    """Estimated workspace of one image's pairs in a fused pass (``bbox``: its [n,4] boxes as given)."""
    from .engine import shared_conv3_enabled
    from .pairs import count_shared_windows, normalise_boxes
    n = int(bbox.shape[0])
    pairs = n * (n - 1)
    per_pair, per_win, per_pair_full, per_obj = _COST[bool(train)]
    if pairs == 0:
        return per_obj * n
    xw = None
    if cfg.feature_size == 32:
        bb = normalise_boxes(torch.as_tensor(bbox).detach().cpu(), cfg.feature_size)
        xw = count_shared_windows(bb, [0, n])
    if xw is not None and shared_conv3_enabled(xw, pairs):
        return per_pair * pairs + per_win * xw + per_obj * n
    return per_pair_full * pairs + per_obj * n


def plan_image_groups(cfg, batch, train: bool, budget_bytes: float):
    """Consecutive image ranges [(a, b), ...] whose estimated workspace (+15 %) stays within ``budget_bytes``; an image that
    exceeds the budget on its own is a group of one (its pairs cannot be split: loss and ranking are per image)."""
    n = [int(b.shape[0]) for b in batch.bbox]
    if 1.15 * sum(_COST[bool(train)][2] * k * (k - 1) + _COST[bool(train)][3] * k for k in n) <= budget_bytes:
        return [(0, len(n))]                        # fits even on the per-pair kernels: no need to count windows (the usual case)
    cost = [1.15 * image_workspace_bytes(cfg, b, train) for b in batch.bbox]
    groups, a, acc = [], 0, 0.0
    for k, c in enumerate(cost):
        if k > a and acc + c > budget_bytes:
            groups.append((a, k))
            a, acc = k, 0.0
        acc += c
    groups.append((a, len(cost)))
    return groups


def split_balanced(num_objects, k: int):
    """At most ``k`` consecutive image ranges with about equal numbers of ordered pairs (lanes of the two-stream step)."""
    pairs = np.asarray([n * (n - 1) for n in num_objects], dtype=np.float64)
    total, B = float(pairs.sum()), len(pairs)
    if k <= 1 or B < 2 or total == 0:
        return [(0, B)]
    cum = np.cumsum(pairs)
    cuts = [0]
    for j in range(1, k):
        c = int(np.searchsorted(cum, total * j / k, side="left")) + 1
        c = min(max(c, cuts[-1] + 1), B - (k - j))
        cuts.append(c)
    cuts.append(B)
    return [(a, b) for a, b in zip(cuts, cuts[1:]) if b > a]


def _budget(model, workspace_budget):
    if workspace_budget is not None:
        return float(workspace_budget)
    b = getattr(model, "workspace_budget_bytes", None)
    if b is not None:
        return float(b)
    # measured when first needed and again every 64 calls (the planner runs every step; what ELSE lives in HBM - the feature
    # extractor, an evaluator's state, another model - changes over an epoch)
    n = getattr(model, "_auto_budget_calls", 0)
    if getattr(model, "_auto_budget", None) is None or n >= 64:
        model._auto_budget = float(default_workspace_budget(model))
        n = 0
    model._auto_budget_calls = n + 1
    return model._auto_budget


def _group_rows(scene: DeviceScene, a: int, b: int) -> torch.Tensor:
    """Pairs of the images [a, b) in the whole minibatch's pair order = the pair order of the group's own scene (restricting
    ``keep_in_batch`` to a subset of images keeps their relative order)."""
    img = scene.image
    return torch.nonzero((img >= a) & (img < b)).flatten()


def forward_pairs_chunked(model, cfg, batch, scene: DeviceScene, groups, iou: Optional[torch.Tensor], select: Optional[torch.Tensor]):
    """``model.forward_pairs`` over image groups; the groups' rows are scattered into full-size outputs in the minibatch's pair order."""
    from .engine import PairOutputs
    dev = scene.bbox.device
    P = scene.n_pairs
    nc = 3 if cfg.hierarchical else 1
    full = PairOutputs(torch.zeros(P, cfg.num_relations, device=dev), torch.zeros(P, 3, device=dev) if cfg.hierarchical else None,
                       torch.zeros(P, device=dev), torch.zeros(P, 512, device=dev), torch.full((P, nc), -float("inf"), device=dev),
                       torch.zeros(P, nc, dtype=torch.int32, device=dev))
    for a, b in groups:
        rows = _group_rows(scene, a, b)
        if rows.numel() == 0:
            continue
        sub = flatten_scene(cfg, slice_batch(batch, a, b), dev)
        assert sub.n_pairs == int(rows.numel())
        out = model.forward_pairs(sub, iou_mask=None if iou is None else iou[rows].contiguous(),
                                  select=None if select is None else select[rows].contiguous())
        full.relation[rows] = out.relation
        if full.super_relation is not None:
            full.super_relation[rows] = out.super_relation
        full.connectivity[rows] = out.connectivity
        full.hidden[rows] = out.hidden
        full.cand_conf[rows] = out.cand_conf
        full.cand_pred[rows] = out.cand_pred
    return full


def overlap_mask(scene: DeviceScene) -> torch.Tensor:
    """[P] uint8: the pair's two boxes share a grid cell (``train_test.py:403-408``)."""
    lib = _lib.load()
    P = scene.n_pairs
    out = torch.empty(P, dtype=torch.uint8, device=scene.bbox.device)
    _lib.check(lib.sgc_overlap_filter(_lib.ptr(scene.bbox), _lib.ptr(scene.sub_idx), _lib.ptr(scene.obj_idx), _lib.ptr(out), P,
                                      _lib.stream_ptr()), "sgc_overlap_filter")
    return out


def _step_filter(scene: DeviceScene, iou: Optional[torch.Tensor]):
    """(included [P] bool, any_overlap [T] bool) on the device: a direction-step is kept when at least one image's two boxes
    overlap (``train_test.py:403-410``; the filter is symmetric and tested once per (g, e), so both directions share the decision)."""
    dev = scene.bbox.device
    if iou is None:
        return torch.ones(scene.n_pairs, dtype=torch.bool, device=dev), torch.ones(scene.n_steps, dtype=torch.bool, device=dev)
    step = scene.step.long()
    per_step = torch.zeros(scene.n_steps, dtype=torch.int32, device=dev).index_add_(0, step, iou.int())
    any_overlap = per_step > 0
    return any_overlap[step], any_overlap


def _unfiltered_selection(scene: DeviceScene, iou: torch.Tensor, evaluators) -> torch.Tensor:
    """[P] bool (device): the pairs whose trunk must be computed when the overlap filter is on.  A filtered candidate can only
    matter when its image has fewer unfiltered pairs than the evaluator ranks (then -inf entries enter the top K and may still
    match a ground-truth triple): such images are computed completely."""
    k_max = max([100] + [int(e.top_k[-1]) for e in evaluators if e is not None])
    image = scene.image.long()
    per_image = torch.zeros(int(scene.image_feature.shape[0]), dtype=torch.int32, device=iou.device).index_add_(0, image, iou.int())
    return iou.bool() | (per_image < k_max)[image]


def feed_evaluators(model, scene: DeviceScene, out, evaluator=None, evaluator_top3=None, overlap=None, directed=None):
    """Append one minibatch to the Recall@K evaluators in the reference's candidate order (``evaluator.py:231-246``).
    ``overlap`` = [P] uint8 device mask of the overlap filter (``train_test.py:403-410``): direction-steps in which no image's
    boxes overlap are skipped entirely (no candidates, no targets); ``None`` = no filter (``training()``, ``:209``).
    Everything stays on the device (the pair tables of the scene are device tensors); the only host synchronisation is the size
    of the kept set.  Returns ``included`` [P] bool device tensor (pairs of the kept steps)."""
    cfg = model.head_config()
    dev = scene.bbox.device
    included, any_overlap = _step_filter(scene, overlap)
    if evaluator is None and evaluator_top3 is None:
        return included
    if directed is None:
        if scene.directed is None:
            raise ValueError("the evaluator feed needs relation targets: flatten the scene from a batch with relationships / subj_or_obj")
        directed = scene.directed
    iou = overlap if overlap is not None else torch.ones(scene.n_pairs, dtype=torch.uint8, device=dev)
    sel = torch.nonzero(included).flatten()
    sizes = (scene.step_ptr[1:] - scene.step_ptr[:-1])[any_overlap]            # pairs per kept direction-step, in order
    which = scene.image.long()[sel]
    tgt = torch.as_tensor(directed).to(dev)[sel].long()
    sub, obj = scene.sub_idx.long()[sel], scene.obj_idx.long()[sel]
    scat, ocat = scene.cats[sub], scene.cats[obj]
    raw = getattr(scene, "_bbox_raw_d", None)
    if raw is None:
        raw = scene._bbox_raw_d = torch.from_numpy(scene.bbox_raw).to(dev)
    sbox, obox = raw[sub], raw[obj]
    logsig = torch.log(torch.sigmoid(out.connectivity[sel]))
    iou_sel = iou[sel].bool()
    if evaluator is not None:
        evaluator.accumulate_candidates(which, out.cand_conf[sel], out.cand_pred[sel], tgt, logsig, scat, ocat, sbox, obox,
                                        iou_mask=iou_sel, call_sizes=sizes)
    if evaluator_top3 is not None and cfg.hierarchical:
        conf3 = out.cand_conf[sel].max(dim=1)[0]
        evaluator_top3.accumulate_candidates(which, conf3, out.cand_pred[sel], tgt, logsig, scat, ocat, sbox, obox, iou_mask=iou_sel)
    return included


class MinibatchLookahead:
    """Iterate the minibatches of a loader ONE AHEAD: ``prepare(data)`` (the drivers' ``_to_batch`` + ``flatten_scene``: loader,
    feature extractor launches, ragged lists -> pinned staging -> device) of minibatch k+1 runs when the consumer calls
    ``fetch_next()`` - which ``evaluate_minibatch(while_running=...)`` does right after it has enqueued the forward of minibatch k,
    i.e. while the device is busy, instead of after the feed's host synchronisation with the device idle (4-5 ms of a 20 ms evaluated
    minibatch at 8 x 64).  ``prepare(index, data)`` returns None for a minibatch to skip and ``MinibatchLookahead.STOP`` to end the
    iteration.  Yields (index, prepared)."""
    STOP = object()

    def __init__(self, loader, prepare):
        self._it, self._prepare, self._next, self._fetched = enumerate(loader), prepare, None, False
        self.fetch_next()

    def fetch_next(self):
        if self._fetched:
            return
        self._fetched, self._next = True, None
        for i, data in self._it:
            item = self._prepare(i, data)
            if item is MinibatchLookahead.STOP:
                self._it = iter(())
                return
            if item is not None:
                self._next = (i, item)
                return

    def __iter__(self):
        return self

    def __next__(self):
        self.fetch_next()                        # the consumer did not look ahead (an exception path, a caller without the hook)
        if self._next is None:
            raise StopIteration
        cur, self._fetched = self._next, False
        return cur


def evaluate_minibatch(model, batch, evaluator=None, evaluator_top3=None, overlap_filtering: bool = True,
                       scene: Optional[DeviceScene] = None, skip_filtered: bool = False, workspace_budget: Optional[float] = None,
                       while_running=None):
    """Returns (scene, outputs, included[P] bool numpy, directed targets numpy).
    ``skip_filtered=True`` runs the per-pair trunk only for the pairs that pass the overlap filter (about 40 % of the ordered pairs
    on the synthetic boxes) in every image that has at least top-K such pairs: Recall@K is unchanged (a filtered pair's
    confidence is -inf either way, ``evaluator.py:131-134``, and cannot reach the top K there); ``outputs`` of the skipped pairs
    are zeros.  The reference cannot skip them: its batched per-step call always scores the whole batch.
    ``model.last_connectivity_stats`` holds the counters of ``evaluate_one_direction`` (``train_utils.py:176-184``) summed over the
    kept steps ([5] int64 device tensor; None with ``skip_filtered``).
    ``workspace_budget`` (bytes; default ``model.workspace_budget_bytes`` or 70 % of the free HBM): a minibatch whose fused pass
    would need more is scored in consecutive image groups (``plan_image_groups``) - same outputs, same evaluator feed.
    ``while_running``: called once, with no arguments, after the forward has been enqueued and before the evaluator feed's first
    host synchronisation - the place for host work that does not depend on this minibatch (``MinibatchLookahead.fetch_next``)."""
    cfg = model.head_config()
    dev = next(model.parameters()).device
    if scene is None:
        scene = flatten_scene(cfg, batch, dev)
    iou = overlap_mask(scene) if overlap_filtering else None
    select = None
    if skip_filtered and overlap_filtering:
        select = _unfiltered_selection(scene, iou, (evaluator, evaluator_top3))
    groups = plan_image_groups(cfg, batch, False, _budget(model, workspace_budget))
    if len(groups) > 1:
        out = forward_pairs_chunked(model, cfg, batch, scene, groups, iou, select)
    else:
        out = model.forward_pairs(scene, iou_mask=iou, select=select)
    model.last_image_groups = groups
    if while_running is not None:
        while_running()
    directed_d = scene.directed
    if directed_d is None:
        directed_d = torch.from_numpy(pair_targets_fast(batch.relationships, batch.subj_or_obj, scene.pidx).astype(np.int32)).to(dev)
    included = feed_evaluators(model, scene, out, evaluator, evaluator_top3, overlap=iou, directed=directed_d)
    model.last_connectivity_stats = None
    if select is None and scene.raw_target is not None:
        model.last_connectivity_stats = model.engine().connectivity_stats(out.connectivity, directed_d, scene.raw_target,
                                                                          included.to(torch.uint8))
    return scene, out, included.cpu().numpy(), directed_d.cpu().numpy().astype(np.int64)


def evaluate_sgdet_minibatch(model, image_feature, image_depth, categories_pred, cat_pred_confidence, bbox_pred, evaluator,
                             sub2super=None, targets=None, overlap_filtering: bool = True, skip_filtered: bool = False,
                             workspace_budget: Optional[float] = None):
    """SGDET evaluation of one minibatch (``evaluate.py:375-444``): every ordered pair of the PREDICTED objects of each image
    (per-image lists as returned by ``object_frontend.DetrFrontEnd.sgdet``: categories, category confidences, boxes
    (x0,x1,y0,y1) on the grid, one entry per image of ``image_feature``), overlap filter, evaluator fed in the reference's
    candidate order with ``predcls=False`` semantics (category confidences added).  ``targets`` =
    (relationships, subj_or_obj, categories_target, bbox_target) sets the ground truth through ``match_target_sgd`` +
    ``Evaluator.accumulate_target``; call ``evaluator.compute(per_class, predcls=False)`` afterwards.
    ``sub2super``: the ``sub2super_cat_dict`` (VG hierarchical label vectors)."""
    from .synthetic import SceneBatch
    cfg = model.head_config()
    dev = next(model.parameters()).device
    if len(categories_pred) != int(image_feature.shape[0]):
        raise ValueError("one predicted-object list per image is required (the reference drops the whole minibatch otherwise)")
    sp = None
    if cfg.dataset == "vg":
        if sub2super is None:
            raise ValueError("sub2super_cat_dict is required for Visual Genome label vectors")
        sp = [[torch.as_tensor(sub2super[int(c)]) for c in cats.tolist()] for cats in categories_pred]
    batch = SceneBatch(image_feature, image_depth, [b.detach().float().cpu() for b in bbox_pred],
                       [c.detach().cpu().long() for c in categories_pred], sp, None, None)
    scene = flatten_scene(cfg, batch, dev)
    iou = overlap_mask(scene) if overlap_filtering else None
    select = _unfiltered_selection(scene, iou, (evaluator,)) if (skip_filtered and overlap_filtering) else None
    groups = plan_image_groups(cfg, batch, False, _budget(model, workspace_budget))       # 16 images x 100+ detections: image groups
    if len(groups) > 1:
        out = forward_pairs_chunked(model, cfg, batch, scene, groups, iou, select)
    else:
        out = model.forward_pairs(scene, iou_mask=iou, select=select)
    model.last_image_groups = groups
    included, any_overlap = _step_filter(scene, iou)
    sel = torch.nonzero(included).flatten()
    sub, obj = scene.sub_idx.long()[sel], scene.obj_idx.long()[sel]
    conf_obj = torch.cat([c.reshape(-1).to(dev, torch.float32) for c in cat_pred_confidence])
    raw = torch.from_numpy(scene.bbox_raw).to(dev)
    iou_sel = iou[sel].bool() if iou is not None else torch.ones(int(sel.numel()), dtype=torch.bool, device=dev)
    evaluator.accumulate_candidates(scene.image.long()[sel], out.cand_conf[sel], out.cand_pred[sel], None,
                                    torch.log(torch.sigmoid(out.connectivity[sel])), scene.cats[sub], scene.cats[obj], raw[sub], raw[obj],
                                    iou_mask=iou_sel, call_sizes=(scene.step_ptr[1:] - scene.step_ptr[:-1])[any_overlap],
                                    cat_confidence=conf_obj[sub] + conf_obj[obj])
    if targets is not None:
        cs, co, bs, bo, rt = match_target_sgd(*targets)
        evaluator.accumulate_target(rt, cs, co, bs, bo)
    return scene, out, included.cpu().numpy()


def train_minibatch(model, batch, optimizer=None, reducer=None, scene: Optional[DeviceScene] = None,
                    workspace_budget: Optional[float] = None, streams: Optional[int] = None, **loss_kw):
    """One optimisation step over all ordered pairs of the minibatch; returns the loss tensor.
    A minibatch whose fused pass would exceed ``workspace_budget`` (bytes; default ``model.workspace_budget_bytes`` or 70 % of
    the free HBM) is run in consecutive image groups: whole-minibatch loss coefficients first, then forward + backward per group
    with its rows of them, the gradients summed and reduced across ranks once.  Equal to the one-pass step up to f32 summation
    order (dropout masks differ: the keep bit is indexed by a pair's position within its pass).
    ``streams`` (default ``model.pipeline_streams`` or 1): image groups run on that many concurrent HIP streams, each with its own
    engine workspace - the HBM-bound kernels of one group (expansion, contraction, im2col / col2im, fc1 assembly, ...) then share
    the chip with the MFMA-bound GEMMs of the other instead of leaving the matrix cores idle; a minibatch that fits in one pass
    is cut into ``streams`` balanced groups for that purpose.  Same arithmetic as the sequential groups.
    After a step taken HERE (``optimizer`` given) with ``TUNING.fused_sgd`` (default) ``model.fc1.weight.grad`` is None - its gradient
    lived in the engine's scratch in GEMM order and the fused update consumed it - while every other parameter keeps its ``.grad``
    like the reference's optimizer leaves it; callers that want to look at fc1.weight's gradient (norm logging, clipping) pass
    ``optimizer=None`` and step themselves, or run under ``engine.tuning(fused_sgd=False)``."""
    cfg = model.head_config()
    dev = next(model.parameters()).device
    if optimizer is not None:
        optimizer.zero_grad(set_to_none=True)
    lanes = int(streams if streams is not None else getattr(model, "pipeline_streams", 1) or 1)
    chunk_free = loss_kw.get("image_feature_aug") is None and loss_kw.get("commonsense") is None
    budget = _budget(model, workspace_budget)
    if loss_kw.get("image_feature_aug") is not None:
        budget *= 0.85          # the augmented view's trunk for the connected pairs lives in a second engine's workspace
    groups = plan_image_groups(cfg, batch, True, budget)
    if len(groups) == 1 and lanes > 1 and chunk_free:
        groups = split_balanced([int(b.shape[0]) for b in batch.bbox], lanes)
    model.last_image_groups = groups
    if scene is None:
        scene = flatten_scene(cfg, batch, dev)
    # fc1.weight's gradient in GEMM order for an optimizer that consumes it (optim.FusedSGD / distributed.ShardedSGD: ``fuse_fc1``):
    # the backward skips the transposition to the reference's column order, the optimizer's fused update un-permutes on the fly.  Not
    # with an all-reduce reducer over several ranks (a rank without pairs would apply its peers' mean gradient untagged); a ShardedSGD
    # that has taken one such gradient takes every later one that way (its accumulators and its peers' must agree).
    fuse_eng = None
    from .engine import TUNING
    # the object that ends up holding the step's gradients: a reducer that owns them (ShardedSGD - also when the caller steps it later,
    # ``optimizer=None``: gradient accumulation), else the optimizer
    owner = reducer if (reducer is not None and getattr(reducer, "owns_grads", False)) else optimizer
    sharded = owner is reducer and reducer is not None
    if sharded and optimizer is not None and optimizer is not reducer:
        raise ValueError("a reducer that owns the gradients (ShardedSGD) is its own optimizer: pass it as both, or optimizer=None")
    allreduce = reducer is not None and not sharded and getattr(reducer, "active", False)
    # decided by configuration only, never by this rank's data: every rank of a step must hand over the same column order (a rank
    # without pairs contributes zeros - order-free - but its optimizer must read its peers' mean gradient the way they wrote it)
    # (a ShardedSGD switches in ``attach`` only, where the ranks agree on it; a local FusedSGD whenever the tuning flag is on)
    if owner is not None and not allreduce and (getattr(owner, "fc1_gemm_order", False) or (not sharded and TUNING.fused_sgd)):
        fuse = getattr(owner, "fuse_fc1", None)
        fuse_eng = fuse(model) if fuse is not None else None
        if fuse_eng is None and getattr(owner, "fc1_gemm_order", False):
            raise RuntimeError("the gradient owner expects fc1.weight's gradient in GEMM order but this module cannot produce it")
    try:
        if fuse_eng is not None:
            fuse_eng.fc1_grad_gemm_order = True
        if len(groups) == 1:
            loss = model.training_step(scene, batch.relationships, batch.subj_or_obj, reducer=reducer, **loss_kw)
        else:
            loss = _train_image_groups(model, cfg, batch, scene, groups, reducer, loss_kw, lanes=lanes)
    except BaseException:
        # a GEMM-order gradient that may already hang on fc1.weight is a view of the engine's scratch in an order nobody else reads
        fc1 = getattr(model, "fc1", None)
        if fuse_eng is not None and fc1 is not None and getattr(fc1.weight, "_sgc_grad_gemm_order", False):
            fc1.weight.grad = None
            fc1.weight._sgc_grad_gemm_order = False
        raise
    finally:
        if fuse_eng is not None:
            fuse_eng.fc1_grad_gemm_order = False
    model.last_scene = scene
    if optimizer is not None:
        try:
            optimizer.step()
        except BaseException:
            fc1 = getattr(model, "fc1", None)
            if fc1 is not None and getattr(fc1.weight, "_sgc_grad_gemm_order", False):
                fc1.weight.grad = None
                fc1.weight._sgc_grad_gemm_order = False
            raise
    return loss


_LANE_STREAMS = {}


def _lane_stream(dev, k):
    key = (torch.device(dev), k)
    if key not in _LANE_STREAMS:
        _LANE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _LANE_STREAMS[key]


def _coupled_terms(model, cfg, batch, scene: DeviceScene, groups, subs, aug, commonsense, lam):
    """The minibatch-level pieces of the contrastive (``train_test.py:260-273``, ``sup_contrast/losses.py:85-181``) and commonsense
    (``train_utils.py:36-62``) terms for a minibatch that runs in image groups.  Both couple all pairs through the FORWARD: the
    SupCon loss reads the hidden rows of every connected pair of the minibatch (two views), the commonsense penalty's per-step
    means count the flagged candidates of every image in the step.  So every group's training-mode forward runs first
    (``model.coupled_forward``; nothing but [M, 512] rows and [P, n_cand] predicates is kept), the loss and its feature gradient /
    the per-candidate coefficients are computed ONCE for the minibatch, and the groups' second pass (same dropout seeds: the same
    forward bit for bit) gets its rows of them.  Costs one extra forward per group - the price of not holding every group's
    workspace at once.  Returns (per-group ``coupled`` dicts, contrastive loss tensor or None)."""
    dev = scene.bbox.device
    eng = model.refresh_weights(backward=True)
    P = scene.n_pairs
    nc = 3 if cfg.hierarchical else 1
    cand_full = torch.zeros(P, nc, dtype=torch.int32, device=dev) if commonsense is not None else None
    conn_full = torch.nonzero(scene.directed >= 0).flatten() if aug is not None else None
    M = int(conn_full.numel()) if conn_full is not None else 0
    feats = torch.zeros(2 * M, 512, dtype=torch.float32, device=dev) if M > 0 else None
    per_group = []
    for (a, b), (rows, sub) in zip(groups, subs):
        if sub is None:
            per_group.append(None)
            continue
        st = model.coupled_forward(sub, None if aug is None else aug[a:b], want_candidates=commonsense is not None)
        pos = None
        if cand_full is not None:
            cand_full[rows] = st["cand_pred"]
        if aug is not None and st["conn_idx"] is not None and int(st["conn_idx"].numel()) > 0:
            pos = torch.searchsorted(conn_full, rows[st["conn_idx"]])       # this group's connected pairs in the minibatch's list
            feats[pos] = st["hidden"]
            feats[M + pos] = st["hidden_aug"]
        per_group.append(dict(seeds=st["seeds"], pos=pos))
    cs_full = None
    if commonsense is not None:
        cs_full = eng.commonsense_coefficients(cand_full, model._commonsense_bitmaps(commonsense, dev), scene.step.long(), scene.n_steps,
                                               scene.cats[scene.sub_idx.long()], scene.cats[scene.obj_idx.long()],
                                               lam["lambda_commonsense"], lam["lambda_cs_weak"], lam["lambda_cs_strong"])
    loss_c, dF = None, None
    if M > 0:
        lam2 = float(lam["lambda_contrast"]) ** 2
        loss_c, dF = eng.supcon_loss(feats, scene.directed[conn_full].to(torch.int32).contiguous(), grad_scale=lam2)
    out = []
    for (rows, sub), g in zip(subs, per_group):
        if g is None:
            out.append(None)
            continue
        out.append(dict(seeds=g["seeds"], cs_coef=None if cs_full is None else cs_full[rows],
                        contrast=None if (dF is None or g["pos"] is None) else (dF[g["pos"]], dF[M + g["pos"]])))
    return out, loss_c


def _train_image_groups(model, cfg, batch, scene: DeviceScene, groups, reducer, loss_kw, lanes: int = 1):
    from .engine import PairOutputs
    dev = scene.bbox.device
    kw = dict(loss_kw)
    aug, commonsense = kw.pop("image_feature_aug", None), kw.pop("commonsense", None)
    lam = dict(lambda_contrast=kw.pop("lambda_contrast", 1.0), lambda_commonsense=kw.pop("lambda_commonsense", 1.0),
               lambda_cs_weak=kw.pop("lambda_cs_weak", 0.1), lambda_cs_strong=kw.pop("lambda_cs_strong", 10.0))
    coupled_terms = aug is not None or commonsense is not None
    if coupled_terms:
        lanes = 1                                   # two passes per group on the module's own engine
        if scene.directed is None:
            raise ValueError("the contrastive / commonsense terms need relation targets in the scene")
    coefs = model.minibatch_loss_coefficients(scene, kw.pop("class_weight", None), kw.pop("lambda_connectivity", 0.1),
                                              kw.pop("lambda_not_connected", 1.0))
    subs = []
    for a, b in groups:
        rows = _group_rows(scene, a, b)
        if rows.numel() == 0:
            subs.append((rows, None))
            continue
        sub_b = slice_batch(batch, a, b)
        sub = flatten_scene(cfg, sub_b, dev)
        assert sub.n_pairs == int(rows.numel())
        sub._batch = sub_b
        subs.append((rows, sub))
    coupled, loss_c = (None, None)
    if coupled_terms:
        coupled, loss_c = _coupled_terms(model, cfg, batch, scene, groups, subs, aug, commonsense, lam)
    P = scene.n_pairs
    nc = 3 if cfg.hierarchical else 1
    full = PairOutputs(torch.zeros(P, cfg.num_relations, device=dev), torch.zeros(P, 3, device=dev) if cfg.hierarchical else None,
                       torch.zeros(P, device=dev), torch.zeros(P, 512, device=dev), torch.zeros(P, nc, device=dev),
                       torch.zeros(P, nc, dtype=torch.int32, device=dev))
    lanes = max(1, min(int(lanes), len(groups)))
    main = torch.cuda.current_stream(dev)
    lane = [dict(stream=main if lanes == 1 else _lane_stream(dev, k), engine=model.lane_engine(k), acc={},
                 stats=torch.zeros(5, dtype=torch.int64, device=dev), loss=torch.zeros((), device=dev)) for k in range(lanes)]
    if lanes > 1:
        for ln in lane:
            ln["stream"].wait_stream(main)          # coefficients, weights and the full-size outputs are ready; last step's reads are done
    for gi, (a, b) in enumerate(groups):
        ln = lane[gi % lanes]
        with torch.cuda.stream(ln["stream"]):
            rows, sub = subs[gi]
            if sub is None:
                continue
            sub_b = sub._batch
            extra = {}
            if coupled is not None:
                extra = dict(coupled=coupled[gi], image_feature_aug=None if aug is None else aug[a:b])
            ln["loss"] = ln["loss"] + model.training_step(sub, sub_b.relationships, sub_b.subj_or_obj,
                                                          loss_coefs=tuple(c[rows] for c in coefs), grads_out=ln["acc"],
                                                          engine=ln["engine"] if coupled is None else None, **extra, **kw)
            out = model.last_outputs
            full.relation[rows] = out.relation
            if full.super_relation is not None:
                full.super_relation[rows] = out.super_relation
            full.connectivity[rows] = out.connectivity
            full.hidden[rows] = out.hidden
            full.cand_conf[rows] = out.cand_conf
            full.cand_pred[rows] = out.cand_pred
            if model.last_connectivity_stats is not None:
                ln["stats"] += model.last_connectivity_stats
    if lanes > 1:
        for ln in lane:
            main.wait_stream(ln["stream"])
    acc, stats, loss = lane[0]["acc"], lane[0]["stats"], lane[0]["loss"]
    for ln in lane[1:]:                               # lane sums -> the minibatch's sums (fixed order: deterministic)
        for name, g in ln["acc"].items():
            if name in acc:
                acc[name].add_(g)
            else:
                acc[name] = g
        stats, loss = stats + ln["stats"], loss + ln["loss"]
    if loss_c is not None:
        if not bool(torch.isnan(loss_c)):
            loss = loss + float(lam["lambda_contrast"]) ** 2 * loss_c          # lambda applied twice (train_test.py:270-273)
        model.last_contrast_loss = loss_c
    for name, p in model.named_parameters():          # no group had a pair: a zero gradient, not a missing one
        if name not in acc:
            acc[name] = torch.zeros_like(p)
    # the minibatch's gradient = the sum over its image groups: mean-reduce it across ranks once, then accumulate like autograd
    if reducer is not None:
        if getattr(reducer, "owns_grads", False) and bool(getattr(reducer, "fc1_gemm_order", False)) != bool(lane[0]["engine"].fc1_grad_gemm_order):
            raise RuntimeError("fc1.weight's gradient order differs from what the sharded optimizer accumulates")
        if "fc1.weight" in acc:
            reducer.hook("fc1.weight", acc["fc1.weight"])
        reducer.finish_grads(acc)
    if not getattr(reducer, "owns_grads", False):
        model.accumulate_grads(acc, lane[0]["engine"])
    model.last_outputs, model.last_connectivity_stats = full, stats
    return loss
