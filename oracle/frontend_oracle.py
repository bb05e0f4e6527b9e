"""CPU oracle for the SGDET / SGCLS object front-end.  TEST INFRASTRUCTURE ONLY.

Literal float32 restatement (PyTorch CPU ops + plain Python loops) of what the reference does between the DETR
decoder outputs and the pair loop: only ``tests/`` may import it; nothing under ``scene_graph_commonsense_amd/``
does and the product path never falls back to it.

Reference map (paths relative to the reference repo):
  detr_candidates          evaluate.py:311-335 (= :545-566): softmax, has-object test, top-k categories and their
                           probabilities, DETR (alphabetical) -> dataset (frequency) class index, cxcywh -> (x0,x1,y0,y1)
                           on the feature grid, clamp, repeat per category, drop mapped category == num_classes
  nms                      torchvision.ops.nms, torchvision==0.15.2 (requirements.txt:160) - NOT present in the build image:
                           restated from its CPU kernel's published semantics (stable descending score sort, greedy,
                           suppress when inter / (area_i + area_j - inter) > threshold, areas without +1)
  per_class_nms            evaluate.py:347-366 (= :573-589): classes in torch.unique order, kept indices concatenated
  iou                      utils.py:58-74 (rasterised 32x32 masks, int() truncation)
  match_object_categories  utils.py:377-425

Pinning: ``tests/golden/make_frontend_golden.py`` runs the reference's own ``utils.match_object_categories`` /
``utils.iou`` and ``dataset_utils.object_class_alp2fre`` (importable with stubs) and stores their outputs; the inline
lines of evaluate.py cannot be imported (tensorboard, process group) and are restated in that script around
plain torch ops, with THIS file's ``nms`` standing in for the absent torchvision: the NMS step is therefore
"parity unpinned" (checked against hand-computed cases in tests/test_frontend_cpu.py only).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def detr_candidates(pred_logits: Tensor, pred_boxes: Tensor, alp2fre: Sequence[int], num_classes: int, topk_cat: int,
                    feature_size: int):
    """evaluate.py:311-345.  Returns per-image lists (images without any object are dropped, as in the reference):
    categories [n_i] int64, confidence [n_i] f32, boxes [n_i,4] f32 as (x0,x1,y0,y1) on the grid, kept image indices."""
    prob = F.softmax(pred_logits, dim=2)
    logits_pred = torch.argmax(prob, dim=2)
    has_object = logits_pred < num_classes
    Q = pred_logits.shape[1]
    top_idx = torch.topk(prob, dim=2, k=topk_cat)[1].view(-1, Q, topk_cat)
    top_val = torch.topk(prob, dim=2, k=topk_cat)[0].view(-1, Q, topk_cat)
    cats, confs, boxes, kept_images = [], [], [], []
    for i in range(pred_logits.shape[0]):
        if torch.sum(has_object[i]) == 0:
            continue
        conf = top_val[i, has_object[i], :].flatten()
        cat = top_idx[i, has_object[i], :].flatten().clone()
        for j in range(len(cat)):
            cat[j] = alp2fre[int(cat[j])]
        cat_mask = cat != num_classes
        b = pred_boxes[i, has_object[i]].clone()
        c = b.clone()
        b[:, [0, 2]] = c[:, [0, 1]] - c[:, [2, 3]] / 2
        b[:, [1, 3]] = c[:, [0, 1]] + c[:, [2, 3]] / 2
        b = torch.clamp(b, 0, 1)
        b = (b * feature_size).repeat_interleave(topk_cat, dim=0)
        cats.append(cat[cat_mask]); confs.append(conf[cat_mask]); boxes.append(b[cat_mask]); kept_images.append(i)
    return cats, confs, boxes, kept_images


def nms(boxes: Tensor, scores: Tensor, iou_threshold: float) -> Tensor:
    """torchvision.ops.nms (0.15.2 CPU kernel): boxes [n,4] (x1,y1,x2,y2); returns kept indices by decreasing score."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = torch.sort(scores, stable=True, descending=True)[1]
    suppressed = [False] * n
    keep = []
    zero = torch.zeros((), dtype=boxes.dtype)
    for _i in range(n):
        i = int(order[_i])
        if suppressed[i]:
            continue
        keep.append(i)
        for _j in range(_i + 1, n):
            j = int(order[_j])
            if suppressed[j]:
                continue
            w = torch.maximum(zero, torch.minimum(x2[i], x2[j]) - torch.maximum(x1[i], x1[j]))
            h = torch.maximum(zero, torch.minimum(y2[i], y2[j]) - torch.maximum(y1[i], y1[j]))
            inter = w * h
            ovr = inter / (areas[i] + areas[j] - inter)
            if float(ovr) > float(iou_threshold):     # the f32 quotient against the DOUBLE threshold (C++ promotion in the kernel)
                suppressed[j] = True
    return torch.tensor(keep, dtype=torch.int64)


def per_class_nms(cat: Tensor, conf: Tensor, box: Tensor, iou_threshold: float, nms_fn=nms):
    """evaluate.py:347-366 for one image: box (x0,x1,y0,y1).  Returns (cat, conf, box, kept candidate indices)."""
    b = box[:, [0, 2, 1, 3]]
    keep_idx: Optional[Tensor] = None
    for cls in torch.unique(cat):
        cur = cat == cls
        k = nms_fn(b[cur], conf[cur], iou_threshold)
        idx = torch.nonzero(cur).flatten()[k]
        keep_idx = idx if keep_idx is None else torch.hstack((keep_idx, idx))
    if keep_idx is None:
        keep_idx = torch.zeros(0, dtype=torch.int64)
    return cat[keep_idx], conf[keep_idx], box[keep_idx], keep_idx


def iou(bbox_target, bbox_pred, feature_size: int = 32) -> float:
    """utils.py:58-74."""
    mp = torch.zeros(feature_size, feature_size)
    mp[int(bbox_pred[0]):int(bbox_pred[1]), int(bbox_pred[2]):int(bbox_pred[3])] = 1
    mt = torch.zeros(feature_size, feature_size)
    mt[int(bbox_target[0]):int(bbox_target[1]), int(bbox_target[2]):int(bbox_target[3])] = 1
    inter = torch.sum(torch.logical_and(mt, mp))
    union = torch.sum(torch.logical_or(mt, mp))
    if union == 0:
        return 0
    return float(inter) / float(union)


def match_object_categories(categories_pred: List[Tensor], cat_pred_confidence: List[Tensor], bbox_pred: List[Tensor],
                            bbox_target: List[Tensor], stable_ties: bool = True):
    """utils.py:377-425.  ``stable_ties`` resolves equal IoUs by the lower prediction index (torch.topk leaves the order of
    equal values unspecified; the product path is stable)."""
    matched, matched_conf = [], []
    target_matched = list(bbox_target)
    if len(bbox_target) != len(bbox_pred):
        return None, None, None
    for i in range(len(bbox_target)):
        repeat = 0
        cur, cur_conf = [], []
        for k, bt in enumerate(bbox_target[i]):
            all_ious = [iou(bt, bp) for bp in bbox_pred[i]]
            if len(all_ious) < 2:
                return None, None, None
            t = torch.tensor(all_ious)
            if stable_ties:
                order = torch.sort(t, stable=True, descending=True)[1][:2]
                top = (t[order], order)
            else:
                top = torch.topk(t, 2)
            if top[0][0] == top[0][1]:
                cur.append(categories_pred[i][top[1][0]]); cur.append(categories_pred[i][top[1][1]])
                cur_conf.append(cat_pred_confidence[i][top[1][0]] * top[0][0])
                cur_conf.append(cat_pred_confidence[i][top[1][1]] * top[0][1])
                tm = target_matched[i]
                target_matched[i] = torch.cat([tm[:k + repeat], tm[k + repeat].view(1, 4), tm[k + repeat:]])
                repeat += 1
            else:
                cur.append(categories_pred[i][top[1][0]])
                cur_conf.append(cat_pred_confidence[i][top[1][0]] * top[0][0])
        matched.append(cur); matched_conf.append(cur_conf)
    return matched, matched_conf, target_matched


def frontend_sgdet(pred_logits: Tensor, pred_boxes: Tensor, alp2fre, num_classes: int = 150, topk_cat: int = 2,
                   feature_size: int = 32, nms_threshold: float = 0.5):
    """evaluate.py:309-366: candidates + per-class NMS.  Returns (cats, confs, boxes, kept image indices)."""
    cats, confs, boxes, imgs = detr_candidates(pred_logits, pred_boxes, alp2fre, num_classes, topk_cat, feature_size)
    for i in range(len(cats)):
        cats[i], confs[i], boxes[i], _ = per_class_nms(cats[i], confs[i], boxes[i], nms_threshold)
    return cats, confs, boxes, imgs
