"""SGDET / SGCLS object front-end on the GPU.

Mirrors what the reference does between the DETR decoder outputs and the pair loop (``evaluate.py:309-366`` for SGDET,
``:543-589`` + ``utils.match_object_categories`` for SGCLS) with three HIP kernels behind the C-ABI
(``csrc/kernels_frontend.hip``): soft-max / top-k / class re-indexing / box conversion per query, per-class NMS per image,
and the ground-truth-box <-> prediction matching.  The outputs keep the reference's shapes and ORDER (per-image lists;
classes ascending then scores descending after NMS) so they can feed ``pair_loop.evaluate_minibatch`` or the reference's
own loops unchanged.

Behavioural notes: equal scores / IoUs resolve to the lower index (``torch.topk`` and ``torchvision.ops.nms`` leave ties to
the sort implementation); images in which no query has an object are dropped, as the reference's list comprehensions do
(``kept_images`` says which survived).
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib


def _alp2fre_tensor(alp2fre, device) -> torch.Tensor:
    if isinstance(alp2fre, dict):
        alp2fre = [alp2fre[i] for i in range(len(alp2fre))]
    return torch.as_tensor(list(alp2fre), dtype=torch.int32, device=device)


class DetrFrontEnd:
    """``DetrFrontEnd(object_class_alp2fre(), num_classes=150, topk_cat=2, feature_size=32, nms=0.5)`` - the constants of
    ``config.yaml:38-41`` / ``dataset_utils.py:606-614``."""

    def __init__(self, alp2fre, num_classes: int = 150, topk_cat: int = 2, feature_size: int = 32, nms: float = 0.5,
                 device="cuda:0"):
        self.device = torch.device(device)
        self.lib = _lib.load()
        self.alp2fre = _alp2fre_tensor(alp2fre, self.device)
        self.num_classes, self.topk, self.F, self.nms = int(num_classes), int(topk_cat), int(feature_size), float(nms)

    # ------------------------------------------------------------------ kernels
    def candidates(self, pred_logits: torch.Tensor, pred_boxes: torch.Tensor):
        """evaluate.py:311-345 without the ragged lists: cand_cat [B,Q,k] int32 (-1 = dropped), cand_conf [B,Q,k], cand_box [B,Q,4]."""
        B, Q, C1 = pred_logits.shape
        if C1 != self.alp2fre.numel():
            raise ValueError("pred_logits has %d classes, the class map %d" % (C1, self.alp2fre.numel()))
        lg = pred_logits.to(self.device, torch.float32).contiguous()
        bx = pred_boxes.to(self.device, torch.float32).contiguous()
        cat = torch.empty(B, Q, self.topk, dtype=torch.int32, device=self.device)
        conf = torch.empty(B, Q, self.topk, dtype=torch.float32, device=self.device)
        box = torch.empty(B, Q, 4, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.sgc_detr_candidates(_lib.ptr(lg), _lib.ptr(bx), _lib.ptr(self.alp2fre), B, Q, C1, self.num_classes,
                                                self.topk, ctypes.c_float(self.F), _lib.ptr(cat), _lib.ptr(conf), _lib.ptr(box),
                                                _lib.stream_ptr()), "sgc_detr_candidates")
        return cat, conf, box

    def nms_slots(self, cat, conf, box):
        B, Q, k = cat.shape
        slot = torch.empty(B, Q * k, dtype=torch.int32, device=self.device)
        count = torch.empty(B, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.sgc_nms_per_class(_lib.ptr(cat), _lib.ptr(conf), _lib.ptr(box), B, Q, k, ctypes.c_double(self.nms),
                                              _lib.ptr(slot), _lib.ptr(count), _lib.stream_ptr()), "sgc_nms_per_class")
        return slot, count

    # ------------------------------------------------------------------ reference-shaped results
    def sgdet(self, pred_logits: torch.Tensor, pred_boxes: torch.Tensor):
        """Returns (categories_pred, cat_pred_confidence, bbox_pred, kept_images) as per-image lists after per-class NMS:
        categories int64 [n_i], confidences f32 [n_i], boxes f32 [n_i,4] = (x0,x1,y0,y1) on the feature grid."""
        cat, conf, box = self.candidates(pred_logits, pred_boxes)
        slot, count = self.nms_slots(cat, conf, box)
        B, Q, k = cat.shape
        counts = count.tolist()                      # the only host synchronisation: the ragged list lengths
        cats, confs, boxes, kept = [], [], [], []
        cat2, conf2 = cat.view(B, Q * k), conf.view(B, Q * k)
        for b in range(B):
            if counts[b] == 0:
                continue
            s = slot[b, :counts[b]].long()
            cats.append(cat2[b, s].long()); confs.append(conf2[b, s]); boxes.append(box[b, s // k]); kept.append(b)
        return cats, confs, boxes, kept

    def match_object_categories(self, categories_pred: List[torch.Tensor], cat_pred_confidence: List[torch.Tensor],
                                bbox_pred: List[torch.Tensor], bbox_target: List[torch.Tensor]):
        """Drop-in for ``utils.match_object_categories`` (utils.py:377-425).  The per-image results are 1-D tensors (iterating
        them yields the 0-d tensors the reference's Python lists hold)."""
        if len(bbox_target) != len(bbox_pred):
            return None, None, None
        B = len(bbox_target)
        n_pred = [int(b.shape[0]) for b in bbox_pred]
        n_tgt = [int(b.shape[0]) for b in bbox_target]
        if any(nt > 0 and npd < 2 for nt, npd in zip(n_tgt, n_pred)):
            return None, None, None
        dev = self.device
        pp = torch.tensor([0] + list(torch.tensor(n_pred).cumsum(0).tolist()), dtype=torch.int32, device=dev)
        tp = torch.tensor([0] + list(torch.tensor(n_tgt).cumsum(0).tolist()), dtype=torch.int32, device=dev)
        pbox = torch.cat([b.to(dev, torch.float32).reshape(-1, 4) for b in bbox_pred]).contiguous() if sum(n_pred) else torch.zeros(0, 4, device=dev)
        tbox = torch.cat([b.to(dev, torch.float32).reshape(-1, 4) for b in bbox_target]).contiguous() if sum(n_tgt) else torch.zeros(0, 4, device=dev)
        NT = int(tbox.shape[0])
        idx = torch.empty(NT, 2, dtype=torch.int32, device=dev)
        val = torch.empty(NT, 2, dtype=torch.float32, device=dev)
        if NT:
            _lib.check(self.lib.sgc_match_boxes_top2(_lib.ptr(pbox), _lib.ptr(pp), _lib.ptr(tbox), _lib.ptr(tp), B, max(n_tgt), self.F,
                                                     _lib.ptr(idx), _lib.ptr(val), _lib.stream_ptr()), "sgc_match_boxes_top2")
        tie = val[:, 0] == val[:, 1]                                   # "top two come from the same repeated bounding box"
        take = torch.stack([torch.ones_like(tie), tie], dim=1)         # [NT,2]: first always, second on a tie
        rep = 1 + tie.long()
        # image-local prediction index -> index into the concatenated predictions; one masked selection for the whole batch
        img_of_tgt = torch.repeat_interleave(torch.arange(B, device=dev), torch.tensor(n_tgt, device=dev))
        gsel = (idx.long() + pp[:-1].long()[img_of_tgt][:, None])[take]          # row-major: (k,0) then (k,1)
        iou = val[take]
        allcat = torch.cat([c.to(dev).reshape(-1) for c in categories_pred]) if sum(n_pred) else torch.zeros(0, dtype=torch.int64, device=dev)
        allconf = torch.cat([c.to(dev, torch.float32).reshape(-1) for c in cat_pred_confidence]) if sum(n_pred) else torch.zeros(0, device=dev)
        m_all, c_all = allcat[gsel], allconf[gsel] * iou
        t_all = torch.repeat_interleave(tbox if all(t.dtype == torch.float32 for t in bbox_target) else
                                        torch.cat([b.to(dev).reshape(-1, 4) for b in bbox_target]), rep, dim=0)
        per_img = torch.zeros(B, dtype=torch.int64, device=dev).index_add_(0, img_of_tgt, rep).tolist()   # the one host sync
        matched, matched_conf, target_matched = list(m_all.split(per_img)), list(c_all.split(per_img)), list(t_all.split(per_img))
        return matched, matched_conf, target_matched

    def sgcls(self, pred_logits, pred_boxes, bbox_target: List[torch.Tensor]):
        """evaluate.py:543-600: candidates + NMS + matching against the ground-truth boxes of the kept images."""
        cats, confs, boxes, kept = self.sgdet(pred_logits, pred_boxes)
        tgt = [bbox_target[b] for b in kept]
        m, mc, tm = self.match_object_categories(cats, confs, boxes, tgt)
        return m, mc, tm, kept
