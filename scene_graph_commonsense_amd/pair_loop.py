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
from .pairs import DeviceScene, flatten_scene, pair_targets_fast


def overlap_mask(scene: DeviceScene) -> torch.Tensor:
    """[P] uint8: the pair's two boxes share a grid cell (``train_test.py:403-408``)."""
    lib = _lib.load()
    P = scene.pidx.n_pairs
    out = torch.empty(P, dtype=torch.uint8, device=scene.bbox.device)
    _lib.check(lib.sgc_overlap_filter(_lib.ptr(scene.bbox), _lib.ptr(scene.sub_idx), _lib.ptr(scene.obj_idx), _lib.ptr(out), P,
                                      _lib.stream_ptr()), "sgc_overlap_filter")
    return out


def evaluate_minibatch(model, batch, evaluator=None, evaluator_top3=None, overlap_filtering: bool = True,
                       scene: Optional[DeviceScene] = None):
    """Returns (scene, outputs, included[P] bool numpy, directed targets numpy)."""
    cfg = model.head_config()
    dev = next(model.parameters()).device
    if scene is None:
        scene = flatten_scene(cfg, batch, dev)
    pidx = scene.pidx
    P = pidx.n_pairs
    iou = overlap_mask(scene) if overlap_filtering else torch.ones(P, dtype=torch.uint8, device=dev)
    out = model.forward_pairs(scene, iou_mask=iou)
    directed = pair_targets_fast(batch.relationships, batch.subj_or_obj, pidx)
    iou_h = iou.cpu().numpy().astype(bool)
    n_steps = len(pidx.call_sizes)
    any_overlap = np.bincount(pidx.step[iou_h], minlength=n_steps) > 0
    # the filter is symmetric, and the reference tests it once per (g,e): both directions share the decision
    included = any_overlap[pidx.step]
    if evaluator is not None or evaluator_top3 is not None:
        sel = torch.from_numpy(np.nonzero(included)[0]).to(dev)
        sizes = pidx.call_sizes[any_overlap]
        which = torch.from_numpy(pidx.image[included]).to(dev)
        tgt = torch.from_numpy(directed[included]).to(dev)
        scat, ocat = scene.cats[scene.sub_idx.long()][sel], scene.cats[scene.obj_idx.long()][sel]
        raw = torch.from_numpy(scene.bbox_raw).to(dev)
        sbox, obox = raw[scene.sub_idx.long()][sel], raw[scene.obj_idx.long()][sel]
        logsig = torch.log(torch.sigmoid(out.connectivity[sel]))
        iou_sel = iou[sel].bool()
        if evaluator is not None:
            # iou_mask was already folded into cand_conf by the head kernel (-inf), re-applied here for clarity
            evaluator.accumulate_candidates(which, out.cand_conf[sel], out.cand_pred[sel], tgt, logsig, scat, ocat, sbox, obox,
                                            iou_mask=iou_sel, call_sizes=sizes)
        if evaluator_top3 is not None and cfg.hierarchical:
            conf3 = out.cand_conf[sel].max(dim=1)[0]
            evaluator_top3.accumulate_candidates(which, conf3, out.cand_pred[sel], tgt, logsig, scat, ocat, sbox, obox,
                                                 iou_mask=iou_sel)
    return scene, out, included, directed


def train_minibatch(model, batch, optimizer=None, reducer=None, scene: Optional[DeviceScene] = None, **loss_kw):
    """One optimisation step over all ordered pairs of the minibatch; returns the loss tensor."""
    cfg = model.head_config()
    dev = next(model.parameters()).device
    if scene is None:
        scene = flatten_scene(cfg, batch, dev)
    if optimizer is not None:
        optimizer.zero_grad(set_to_none=True)
    loss = model.training_step(scene, batch.relationships, batch.subj_or_obj,
                               grad_hook=None if reducer is None else reducer.hook, **loss_kw)
    if reducer is not None:
        reducer.finish(list(model.named_parameters()))
    if optimizer is not None:
        optimizer.step()
    return loss
