"""Fused replacements for the reference's pair loops (the bodies of ``testing()`` / ``eval_pc`` and ``training()``).

``evaluate_minibatch`` is what ``train_test.py:373-437`` + ``train_utils.py:160-196`` do for one minibatch: score
every ordered pair, apply the overlap filter (a step in which no image's two boxes overlap is skipped entirely -
no candidates and no targets), and feed the Recall@K evaluators in the reference's candidate order.
``train_minibatch`` is ``train_test.py:174-277`` (pass ``image_feature_aug`` for the contrastive term; no commonsense term).
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from . import _lib
from .pairs import DeviceScene, flatten_scene, match_target_sgd, pair_targets_fast


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


def evaluate_minibatch(model, batch, evaluator=None, evaluator_top3=None, overlap_filtering: bool = True,
                       scene: Optional[DeviceScene] = None, skip_filtered: bool = False):
    """Returns (scene, outputs, included[P] bool numpy, directed targets numpy).
    ``skip_filtered=True`` runs the per-pair trunk only for the pairs that pass the overlap filter (about 40 % of the ordered pairs
    on the synthetic boxes) in every image that has at least top-K such pairs: Recall@K is unchanged (a filtered pair's
    confidence is -inf either way, ``evaluator.py:131-134``, and cannot reach the top K there); ``outputs`` of the skipped pairs
    are zeros.  The reference cannot skip them: its batched per-step call always scores the whole batch.
    ``model.last_connectivity_stats`` holds the counters of ``evaluate_one_direction`` (``train_utils.py:176-184``) summed over the
    kept steps ([5] int64 device tensor; None with ``skip_filtered``)."""
    cfg = model.head_config()
    dev = next(model.parameters()).device
    if scene is None:
        scene = flatten_scene(cfg, batch, dev)
    iou = overlap_mask(scene) if overlap_filtering else None
    select = None
    if skip_filtered and overlap_filtering:
        select = _unfiltered_selection(scene, iou, (evaluator, evaluator_top3))
    out = model.forward_pairs(scene, iou_mask=iou, select=select)
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
                             sub2super=None, targets=None, overlap_filtering: bool = True, skip_filtered: bool = False):
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
    out = model.forward_pairs(scene, iou_mask=iou, select=select)
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


def train_minibatch(model, batch, optimizer=None, reducer=None, scene: Optional[DeviceScene] = None, **loss_kw):
    """One optimisation step over all ordered pairs of the minibatch; returns the loss tensor."""
    cfg = model.head_config()
    dev = next(model.parameters()).device
    if scene is None:
        scene = flatten_scene(cfg, batch, dev)
    if optimizer is not None:
        optimizer.zero_grad(set_to_none=True)
    loss = model.training_step(scene, batch.relationships, batch.subj_or_obj, reducer=reducer, **loss_kw)
    model.last_scene = scene
    if optimizer is not None:
        optimizer.step()
    return loss
