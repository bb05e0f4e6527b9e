"""Drop-in ``Evaluator`` / ``Evaluator_Top3`` (reference ``evaluator.py:15-586, 589-790``).

Same constructor, ``accumulate`` / ``accumulate_target`` / ``compute`` / ``compute_precision`` / ``clear_data``
signatures and return tuples.  What changes underneath:

* candidates are appended to lists and concatenated once (the reference re-allocates with ``hstack`` per step);
* the per-image ranking (``argsort`` + top-100, ``evaluator.py:292-316``) is the HIP kernel ``sgc_topk_per_image``
  (radix select + bitonic sort in LDS), *stable* on ties - the reference's ``torch.argsort`` is not, so tie order
  is defined here as append order;
* the top-100 x ground-truth matching (label equality, grid IoU >= 0.5 on both boxes, predicate equality; scan
  continues past a label/box match whose predicate differs) is vectorised on the host instead of a triple Python
  loop of device scalar compares.  Grid IoU of two rasterised rectangles is their closed-form integer overlap.

Reference quirks that are kept: ``confidence += connectivity`` mutates state inside ``compute``; hit/target
counters persist across ``clear_data``; ``-inf`` candidates are not removed; ``Evaluator_Top3`` counts a hit for
``j < max(k, num_target)``.  There is no CPU ranking fallback: tensors must live on the GPU.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from .pairs import slice_norm

# object-category equivalence classes used when labels are predicted (reference utils.py:355-373; data)
_EQUIV = [[1, 5, 11, 23, 38, 44, 121, 124, 148, 149], [0, 50], [92, 137]]
_UNSYMM = {123: [14, 63, 95, 87, 123], 108: [89, 102, 67, 72, 71, 81, 96, 105, 90, 111, 108],
           60: [145, 106, 142, 144, 77, 60]}


def _equiv_matrix(n: int = 160) -> np.ndarray:
    m = np.eye(n, dtype=bool)
    for grp in _EQUIV:
        for a in grp:
            for b in grp:
                m[a, b] = True
    for key, members in _UNSYMM.items():
        for t in members:
            m[key, t] = True
            m[t, key] = True
    return m


def _norm_boxes(b: np.ndarray, F: int) -> np.ndarray:
    out = np.empty(b.shape, dtype=np.int64)
    flat_in, flat_out = b.reshape(-1), out.reshape(-1)
    for i in range(flat_in.shape[0]):
        flat_out[i] = slice_norm(int(flat_in[i]), F)
    return out


def grid_iou_matrix(bt: np.ndarray, bp: np.ndarray, F: int) -> np.ndarray:
    """IoU of rasterised rectangles (x0,x1,y0,y1) [T,4] x [K,4] -> [T,K]; union == 0 -> 0 (evaluator.py:84-94)."""
    bt, bp = _norm_boxes(bt, F), _norm_boxes(bp, F)
    wt = np.maximum(bt[:, 1] - bt[:, 0], 0); ht = np.maximum(bt[:, 3] - bt[:, 2], 0)
    wp = np.maximum(bp[:, 1] - bp[:, 0], 0); hp = np.maximum(bp[:, 3] - bp[:, 2], 0)
    at, ap = (wt * ht)[:, None], (wp * hp)[None, :]
    iw = np.minimum(bt[:, None, 1], bp[None, :, 1]) - np.maximum(bt[:, None, 0], bp[None, :, 0])
    ih = np.minimum(bt[:, None, 3], bp[None, :, 3]) - np.maximum(bt[:, None, 2], bp[None, :, 2])
    inter = np.maximum(iw, 0) * np.maximum(ih, 0)
    inter = np.where((at > 0) & (ap > 0), inter, 0)
    union = at + ap - inter
    return np.where(union > 0, inter / np.maximum(union, 1), 0.0)


def rank_topk(conf: torch.Tensor, which: torch.Tensor, K: int):
    """Group candidates per image and rank each group with the HIP top-K kernel.
    Returns (images [n_img], per-image candidate positions (indices into the flat arrays) list, top [n_img,K] local idx)."""
    if not conf.is_cuda:
        raise RuntimeError("Evaluator ranking runs on the GPU (sgc_topk_per_image); move the evaluator inputs to cuda")
    lib = _lib.load()
    images, counts = torch.unique(which, return_counts=True)
    order = torch.sort(which, stable=True)[1]                       # grouped by image, append order kept
    conf_g = conf[order].contiguous().float()
    seg = torch.zeros(len(images) + 1, dtype=torch.int32, device=conf.device)
    seg[1:] = torch.cumsum(counts, 0).int()
    out_idx = torch.empty(len(images), K, dtype=torch.int32, device=conf.device)
    out_cnt = torch.empty(len(images), dtype=torch.int32, device=conf.device)
    _lib.check(lib.sgc_topk_per_image(_lib.ptr(conf_g), _lib.ptr(seg), len(images), K, _lib.ptr(out_idx), _lib.ptr(out_cnt),
                                      _lib.stream_ptr()), "sgc_topk_per_image")
    return images.cpu().numpy(), order.cpu().numpy(), seg.cpu().numpy(), out_idx.cpu().numpy(), out_cnt.cpu().numpy()


def first_hit(label_ok: np.ndarray, iou_ok: np.ndarray, pred_ok: np.ndarray) -> np.ndarray:
    """[T,K] conditions -> rank of the first candidate satisfying all of them, or K if none."""
    ok = label_ok & iou_ok & pred_ok
    K = ok.shape[1]
    return np.where(ok.any(1), ok.argmax(1), K)


def rank_topk_device(conf: torch.Tensor, which: torch.Tensor, K: int):
    """``rank_topk`` without the host copies: (images [n_img] sorted, keep_pos [n_img,K] int32 flat candidate positions in ranked
    order (-1 padded), keep_cnt [n_img] int32, top [n_img,K] image-local indices, seg [n_img+1], order)."""
    if not conf.is_cuda:
        raise RuntimeError("Evaluator ranking runs on the GPU (sgc_topk_per_image); move the evaluator inputs to cuda")
    lib = _lib.load()
    images, counts = torch.unique(which, return_counts=True)
    order = torch.sort(which, stable=True)[1]                       # grouped by image, append order kept
    conf_g = conf[order].contiguous().float()
    seg = torch.zeros(len(images) + 1, dtype=torch.int32, device=conf.device)
    seg[1:] = torch.cumsum(counts, 0).int()
    top = torch.empty(len(images), K, dtype=torch.int32, device=conf.device)
    cnt = torch.empty(len(images), dtype=torch.int32, device=conf.device)
    _lib.check(lib.sgc_topk_per_image(_lib.ptr(conf_g), _lib.ptr(seg), len(images), K, _lib.ptr(top), _lib.ptr(cnt),
                                      _lib.stream_ptr()), "sgc_topk_per_image")
    flat = (seg[:-1, None].long() + top.clamp(min=0).long()).clamp(max=max(int(order.numel()) - 1, 0))
    keep_pos = torch.where(top >= 0, order[flat], torch.full_like(flat, -1)).int().contiguous()
    return images, keep_pos, cnt, top, seg, order


def recall_hits(cand, keep_pos, keep_cnt, targets, K, feature_size, iou_thresh, equiv=None):
    """Device hit test (``sgc_recall_hits``).  ``cand`` = dict(scat, ocat int64 [n]; pred int64 [n] or [n,3]; sbox, obox [n,4]),
    ``targets`` = dict(row int32, rel / scat / ocat int64, sbox / obox [t,4]) of the connected ground-truth triples.  Returns the
    hit ranks as a host int array [t] (K = no hit)."""
    lib = _lib.load()
    dev = keep_pos.device
    n_t = int(targets["rel"].shape[0])
    if n_t == 0:
        return np.zeros(0, dtype=np.int32)
    i64 = lambda x: x.to(dev, torch.int64).contiguous()
    f32 = lambda x: x.to(dev, torch.float32).contiguous().view(-1, 4)
    pred = i64(cand["pred"])
    n_pred = 1 if pred.dim() == 1 else int(pred.shape[1])
    hit = torch.empty(n_t, dtype=torch.int32, device=dev)
    c_s, c_o, c_sb, c_ob = i64(cand["scat"]), i64(cand["ocat"]), f32(cand["sbox"]), f32(cand["obox"])
    t_row, t_rel, t_s, t_o = targets["row"].to(dev, torch.int32).contiguous(), i64(targets["rel"]), i64(targets["scat"]), i64(targets["ocat"])
    t_sb, t_ob = f32(targets["sbox"]), f32(targets["obox"])
    eq = None if equiv is None else equiv.to(dev, torch.uint8).contiguous()
    _lib.check(lib.sgc_recall_hits(_lib.ptr(c_s), _lib.ptr(c_o), _lib.ptr(pred), n_pred, _lib.ptr(c_sb), _lib.ptr(c_ob), _lib.ptr(keep_pos),
                                   _lib.ptr(keep_cnt), int(K), _lib.ptr(t_row), _lib.ptr(t_rel), _lib.ptr(t_s), _lib.ptr(t_o), _lib.ptr(t_sb),
                                   _lib.ptr(t_ob), n_t, _lib.ptr(eq), 0 if eq is None else int(eq.shape[0]), int(feature_size),
                                   ctypes.c_double(float(iou_thresh)), _lib.ptr(hit), _lib.stream_ptr()), "sgc_recall_hits")
    return hit.cpu().numpy()


class Evaluator:
    def __init__(self, args, num_classes, iou_thresh, top_k, max_cache_size=10000):
        self.args = args
        self.hierar = args["models"]["hierarchical_pred"]
        self.top_k = list(top_k)
        self.num_classes = num_classes
        self.iou_thresh = iou_thresh
        self.num_connected_target = 0.0
        self.motif_total = 0.0
        self.motif_correct = 0.0
        self.result_dict = {k: 0.0 for k in self.top_k}
        self.result_per_class = {k: torch.zeros(self.num_classes) for k in self.top_k}
        self.num_conn_target_per_class = torch.zeros(self.num_classes)
        self.feature_size = args["models"]["feature_size"]
        self.run_mode = args["training"]["run_mode"]
        self.dataset = args["dataset"]["dataset"]
        self.zero_shot_triplets = None
        self.train_triplets = None
        if self.dataset == "vg":
            def _load(key):
                p = args["dataset"].get(key)
                return torch.load(p) if p and os.path.exists(p) else None
            self.train_triplets = _load("train_triplets")
            self.test_triplets = _load("test_triplets")
            zs = _load("zero_shot_triplets")
            self.zero_shot_triplets = set(zs) if zs is not None else set()
            self.result_dict_zs = {k: 0.0 for k in self.top_k}
            self.result_per_class_zs = {k: torch.zeros(self.num_classes) for k in self.top_k}
            self.num_connected_target_zs = 0.0
            self.num_conn_target_per_class_zs = torch.zeros(self.num_classes)
        elif self.dataset == "oiv6":
            self.result_per_class_ap = torch.zeros(self.num_classes)
            self.result_per_class_ap_union = torch.zeros(self.num_classes)
            self.num_conn_target_per_class_ap = torch.zeros(self.num_classes)
        # commonsense filter (run modes train_cs / eval_cs; reference evaluator.py:76-81 loads the two triplet dicts)
        self._cs = None
        self._cs_keys = None
        if self.run_mode in ("train_cs", "eval_cs"):
            suffix = "_gpt4v" if args["models"].get("llm_model") == "gpt4v" else ""
            d = args["dataset"]
            pa = d.get("commonsense_aligned_triplets", "triplets/commonsense_aligned_triplets%s.pt" % suffix)
            pv = d.get("commonsense_violated_triplets", "triplets/commonsense_violated_triplets%s.pt" % suffix)
            self._cs_keys = (list(torch.load(pa).keys()), list(torch.load(pv).keys()))
        self.annotation_paths = None
        self._equiv = _equiv_matrix()
        self.last_topk = {}
        self.clear_data()

    # ------------------------------------------------------------------ state
    def clear_data(self):
        self._l = {k: [] for k in ("which", "conf", "conn", "pred", "scat", "ocat", "sbox", "obox",
                                   "which_t", "rel_t", "scat_t", "ocat_t", "sbox_t", "obox_t")}
        self._conn_added = 0
        self._targets_by_image = None

    def load_annotation_paths(self, annot_path):
        self.annotation_paths = annot_path

    def _cat(self, key):
        lst = self._l[key]
        if len(lst) == 0:
            return None
        if len(lst) > 1:
            lst[:] = [torch.vstack(lst) if lst[0].dim() == 2 else torch.hstack(lst)]
        return lst[0]

    # attribute views with the reference's names
    which_in_batch = property(lambda s: s._cat("which"))
    confidence = property(lambda s: s._cat("conf"))
    connectivity = property(lambda s: s._cat("conn"))
    relation_pred = property(lambda s: s._cat("pred"))
    subject_cat_pred = property(lambda s: s._cat("scat"))
    object_cat_pred = property(lambda s: s._cat("ocat"))
    subject_bbox_pred = property(lambda s: s._cat("sbox"))
    object_bbox_pred = property(lambda s: s._cat("obox"))
    relation_target = property(lambda s: s._cat("rel_t") if s._targets_by_image is None else s._targets_by_image[0])
    which_in_batch_target = property(lambda s: s._cat("which_t"))

    # ------------------------------------------------------------------ accumulate
    def accumulate(self, which_in_batch, relation_pred, relation_target, super_relation_pred, connectivity,
                   subject_cat_pred, object_cat_pred, subject_cat_target, object_cat_target,
                   subject_bbox_pred, object_bbox_pred, subject_bbox_target, object_bbox_target, iou_mask,
                   predcls=True, cat_subject_confidence=None, cat_object_confidence=None, height=None, width=None):
        """Reference per-step feed (``evaluator.py:118-269``): ``relation_pred`` is [b,R] log-probs / logits."""
        a = self.args["models"]
        if self.hierar:
            ng, npos = a["num_geometric"], a["num_possessive"]
            segs = [(0, ng), (ng, ng + npos), (ng + npos, relation_pred.shape[1])]
            mx = [torch.max(relation_pred[:, lo:hi], dim=1) for lo, hi in segs]
            conf = torch.hstack([m[0] for m in mx])
            pred = torch.hstack([m[1] + lo for m, (lo, _) in zip(mx, segs)])
            rep = 3
        else:
            m = torch.max(relation_pred, dim=1)
            conf, pred, rep = m[0].clone(), m[1], 1
        self._append(which_in_batch, conf, pred, connectivity, subject_cat_pred, object_cat_pred, subject_bbox_pred,
                     object_bbox_pred, iou_mask, rep, predcls, relation_target, subject_cat_target, object_cat_target,
                     subject_bbox_target, object_bbox_target, cat_subject_confidence, cat_object_confidence)

    def accumulate_candidates(self, which_in_batch, cand_conf, cand_pred, relation_target, connectivity, subject_cat,
                              object_cat, subject_bbox, object_bbox, iou_mask=None, call_sizes=None, cat_confidence=None):
        """Fused feed: per-pair candidates straight from ``sgc_bayes_head`` ([P,3] / [P,1]) for a whole minibatch in
        reference order.  ``call_sizes`` (pairs per direction-step) reproduces the reference's blocked append order
        [geo block | poss block | sem block] per step; without it the order is per-pair interleaved.
        SGDET / SGCLS (``predcls=False``, evaluator.py:129-130,164-165): pass ``cat_confidence`` [P] = subject + object category
        confidence (added to every candidate of the pair before the overlap mask) and ``relation_target=None`` (targets come
        through ``accumulate_target``)."""
        rep = cand_conf.shape[1]
        P = cand_conf.shape[0]
        if call_sizes is not None and rep > 1:
            # position of candidate (pair p, super s) in the reference's append order
            sizes = (call_sizes if torch.is_tensor(call_sizes) else torch.as_tensor(np.asarray(call_sizes))).to(cand_conf.device).long()
            starts = torch.cumsum(sizes, 0) - sizes
            step_of = torch.repeat_interleave(torch.arange(len(sizes), device=sizes.device), sizes)
            within = torch.arange(P, device=sizes.device) - starts[step_of]
            pos = (rep * starts[step_of])[:, None] + torch.arange(rep, device=sizes.device)[None, :] * sizes[step_of][:, None] \
                + within[:, None]
            perm = torch.empty(P * rep, dtype=torch.long, device=sizes.device)
            perm[pos.reshape(-1)] = torch.arange(P * rep, device=sizes.device)
            pair_of = perm // rep
            conf = cand_conf.reshape(-1)[perm]
            pred = cand_pred.reshape(-1)[perm].long()
        else:
            pair_of = torch.arange(P, device=cand_conf.device).repeat_interleave(rep)
            conf, pred = cand_conf.reshape(-1), cand_pred.reshape(-1).long()
        conf = conf.clone()
        if cat_confidence is not None:
            conf = conf + cat_confidence[pair_of]
        if iou_mask is not None:
            conf[~iou_mask.bool()[pair_of]] = -math.inf
        conf = self._commonsense(subject_cat[pair_of], pred, object_cat[pair_of], conf)
        L = self._l
        L["which"].append(which_in_batch[pair_of]); L["conf"].append(conf); L["pred"].append(pred)
        L["conn"].append(connectivity[pair_of]); L["scat"].append(subject_cat[pair_of]); L["ocat"].append(object_cat[pair_of])
        L["sbox"].append(subject_bbox[pair_of]); L["obox"].append(object_bbox[pair_of])
        if relation_target is not None:
            L["which_t"].append(which_in_batch); L["rel_t"].append(relation_target)
            L["scat_t"].append(subject_cat); L["ocat_t"].append(object_cat)
            L["sbox_t"].append(subject_bbox); L["obox_t"].append(object_bbox)

    def _commonsense(self, scat, pred, ocat, conf):
        if self._cs_keys is None:
            return conf
        if self._cs is None or self._cs.device != conf.device:
            from .commonsense import TripletBitmaps
            self._cs = TripletBitmaps(self._cs_keys[0], self._cs_keys[1], self.args["models"]["num_classes"], self.num_classes,
                                      conf.device)
        return self._cs.filter_(scat, pred, ocat, conf.contiguous().float())

    def _append(self, which, conf, pred, conn, scat, ocat, sbox, obox, iou_mask, rep, predcls, rel_t, scat_t, ocat_t,
                sbox_t, obox_t, csub, cobj):
        if not predcls:
            conf = conf + (csub + cobj).repeat(rep)
        conf = conf.clone()
        conf[~iou_mask.repeat(rep)] = -math.inf
        conf = self._commonsense(scat.repeat(rep), pred, ocat.repeat(rep), conf)
        L = self._l
        L["which"].append(which.repeat(rep)); L["conf"].append(conf); L["pred"].append(pred)
        L["conn"].append(conn.repeat(rep)); L["scat"].append(scat.repeat(rep)); L["ocat"].append(ocat.repeat(rep))
        L["sbox"].append(sbox.repeat(rep, 1)); L["obox"].append(obox.repeat(rep, 1))
        if predcls:
            L["which_t"].append(which); L["rel_t"].append(rel_t)
            L["scat_t"].append(scat_t); L["ocat_t"].append(ocat_t)
            L["sbox_t"].append(sbox_t); L["obox_t"].append(obox_t)

    def accumulate_target(self, relation_target, subject_cat_target, object_cat_target, subject_bbox_target,
                          object_bbox_target):
        """SGCLS/SGDET: per-image lists (entries may be None), ``evaluator.py:272-277``."""
        self._targets_by_image = (relation_target, subject_cat_target, object_cat_target, subject_bbox_target,
                                  object_bbox_target)

    # ------------------------------------------------------------------ compute
    def compute(self, per_class=False, predcls=True):
        """Recall@K / mR@K / zero-shot recall (``evaluator.py:280-367``).  Ranking (``sgc_topk_per_image``) and the hit test
        (``sgc_recall_hits``: one wavefront per ground-truth triple) run on the device; only the connected ground-truth triples
        and their hit ranks come back to the host for the counters and the zero-shot key lookup."""
        if self._cat("conf") is not None:
            conf = self._cat("conf")
            conf += self._cat("conn")                         # reference mutates its state the same way
            which = self._cat("which")
            K = self.top_k[-1]
            images, keep_pos, keep_cnt, top, _, _ = rank_topk_device(conf, which, K)
            dev = conf.device
            top_h, cnt_h, images_h = top.cpu().numpy(), keep_cnt.cpu().numpy(), images.cpu().numpy()
            for r, image in enumerate(images_h):
                self.last_topk[int(image)] = top_h[r, :cnt_h[r]].copy()
            if self._targets_by_image is None:
                rel_all = self._cat("rel_t")
                sel = torch.nonzero(rel_all != -1).flatten()
                wt = self._cat("which_t")[sel]
                row = torch.searchsorted(images, wt)
                known = images[row.clamp(max=len(images) - 1)] == wt              # targets of images without any candidate are not scanned
                sel, row = sel[known], row[known]
                tg = dict(row=row.int(), rel=rel_all[sel], scat=self._cat("scat_t")[sel], ocat=self._cat("ocat_t")[sel],
                          sbox=self._cat("sbox_t")[sel], obox=self._cat("obox_t")[sel])
            else:                                              # per-image lists (SGDET / SGCLS, accumulate_target)
                tb = self._targets_by_image
                rows, parts = [], [[] for _ in range(5)]
                for r, image in enumerate(images_h):
                    if tb[0][int(image)] is None:
                        continue
                    vals = [torch.as_tensor(x[int(image)]).to(dev) for x in tb]
                    ok = torch.nonzero(vals[0] != -1).flatten()
                    rows.append(torch.full((len(ok),), r, dtype=torch.int32, device=dev))
                    for k_ in range(5):
                        parts[k_].append(vals[k_][ok].reshape(len(ok), -1) if k_ >= 3 else vals[k_][ok].reshape(-1))
                if rows:
                    tg = dict(row=torch.cat(rows), rel=torch.cat(parts[0]), scat=torch.cat(parts[1]), ocat=torch.cat(parts[2]),
                              sbox=torch.cat(parts[3]).float(), obox=torch.cat(parts[4]).float())
                else:
                    tg = dict(row=torch.zeros(0, dtype=torch.int32, device=dev), rel=torch.zeros(0, dtype=torch.int64, device=dev))
            if int(tg["rel"].shape[0]) > 0:
                cand = dict(scat=self._cat("scat"), ocat=self._cat("ocat"), pred=self._cat("pred"), sbox=self._cat("sbox"), obox=self._cat("obox"))
                hit = recall_hits(cand, keep_pos, keep_cnt, tg, K, self.feature_size, self.iou_thresh,
                                  None if predcls else torch.from_numpy(self._equiv.astype(np.uint8)))
                rel_h = tg["rel"].cpu().numpy().astype(np.int64)
                sc_h, oc_h = tg["scat"].cpu().numpy().astype(np.int64), tg["ocat"].cpu().numpy().astype(np.int64)
                n_keep = cnt_h[tg["row"].cpu().numpy()]
                for i in range(len(rel_h)):
                    ti = int(rel_h[i])
                    zs = False
                    if self.dataset == "vg":
                        zs = ("%d_%d_%d" % (int(sc_h[i]), ti, int(oc_h[i]))) in self.zero_shot_triplets
                    for k in self.top_k:
                        if hit[i] < k and hit[i] < n_keep[i]:
                            self.result_dict[k] += 1.0
                            if per_class:
                                self.result_per_class[k][ti] += 1.0
                            if zs:
                                self.result_dict_zs[k] += 1.0
                                if per_class:
                                    self.result_per_class_zs[k][ti] += 1.0
                    self.num_connected_target += 1.0
                    self.num_conn_target_per_class[ti] += 1.0
                    if zs:
                        self.num_connected_target_zs += 1.0
                        self.num_conn_target_per_class_zs[ti] += 1.0
        recall_k = [self.result_dict[k] / max(self.num_connected_target, 1e-3) for k in self.top_k]
        recall_k_per_class = [self.result_per_class[k] / self.num_conn_target_per_class for k in self.top_k]
        mean_recall_k = [torch.nanmean(r) for r in recall_k_per_class]
        recall_k_zs = recall_k_per_class_zs = mean_recall_k_zs = None
        if self.dataset == "vg":
            recall_k_zs = [self.result_dict_zs[k] / max(self.num_connected_target_zs, 1e-3) for k in self.top_k]
            recall_k_per_class_zs = [self.result_per_class_zs[k] / self.num_conn_target_per_class_zs for k in self.top_k]
            mean_recall_k_zs = [torch.nanmean(r) for r in recall_k_per_class_zs]
        return recall_k, recall_k_per_class, mean_recall_k, recall_k_zs, recall_k_per_class_zs, mean_recall_k_zs

    def compute_precision(self):
        """OpenImages weighted mean AP over the top-20 predictions per image (``evaluator.py:522-566``)."""
        conf, which = self._cat("conf"), self._cat("which")
        images, order, seg, top, cnt = rank_topk(conf, which, 20)
        h = {k: self._cat(k).cpu().numpy() for k in ("pred", "scat", "ocat", "sbox", "obox")}
        t = {k: self._cat(k).cpu().numpy() for k in ("which_t", "rel_t", "scat_t", "ocat_t", "sbox_t", "obox_t")}
        F = self.feature_size
        for r, image in enumerate(images):
            pos = order[seg[r]:seg[r + 1]]
            keep = pos[top[r, :cnt[r]]]
            sel = t["which_t"] == image
            ct = np.nonzero(t["rel_t"][sel] != -1)[0]
            rel_t, sc_t, oc_t = t["rel_t"][sel][ct], t["scat_t"][sel][ct], t["ocat_t"][sel][ct]
            sb_t, ob_t = t["sbox_t"][sel][ct], t["obox_t"][sel][ct]
            for j in keep:
                pj = int(h["pred"][j])
                if len(ct):
                    ok = (h["scat"][j] == sc_t) & (h["ocat"][j] == oc_t) & (pj == rel_t)
                    s_iou = grid_iou_matrix(sb_t, h["sbox"][j][None], F)[:, 0]
                    o_iou = grid_iou_matrix(ob_t, h["obox"][j][None], F)[:, 0]
                    u_iou = _union_iou(h["sbox"][j], h["obox"][j], sb_t, ob_t, F)
                    if np.any(ok & (s_iou >= self.iou_thresh) & (o_iou >= self.iou_thresh)):
                        self.result_per_class_ap[pj] += 1.0
                    if np.any(ok & (u_iou >= self.iou_thresh)):
                        self.result_per_class_ap_union[pj] += 1.0
                self.num_conn_target_per_class_ap[pj] += 1.0
        weight = torch.tensor([1974, 120, 27, 2, 284, 571, 2059, 8, 26, 2, 0, 163, 25, 30, 2, 0, 0, 1, 0, 17, 0, 29, 14, 4,
                               3, 0, 6, 0, 67, 5]) + 1        # reference utils.py:270-274 (data)
        ppc = self.result_per_class_ap / self.num_conn_target_per_class_ap
        not_nan = torch.logical_not(torch.isnan(ppc))
        wmp = torch.nansum(ppc * weight) / torch.sum(weight[not_nan])
        ppcu = self.result_per_class_ap_union / self.num_conn_target_per_class_ap
        wmpu = torch.nansum(ppcu * weight) / torch.sum(weight[not_nan])
        return wmp, wmpu


def _union_iou(sb, ob, sb_t, ob_t, F):
    """IoU of (subject OR object) masks, prediction vs each target (``evaluator.py:97-115``); rasterised."""
    def mask(b):
        m = np.zeros((F, F), dtype=bool)
        m[slice_norm(int(b[2]), F):slice_norm(int(b[3]), F), slice_norm(int(b[0]), F):slice_norm(int(b[1]), F)] = True
        return m
    mp = mask(sb) | mask(ob)
    out = np.zeros(len(sb_t))
    for i in range(len(sb_t)):
        mt = mask(sb_t[i]) | mask(ob_t[i])
        u = (mp | mt).sum()
        out[i] = 0.0 if u == 0 else (mp & mt).sum() / u
    return out


class Evaluator_Top3:
    """Recall@K*: a hit if the target equals any of the three per-super-category argmaxes (``evaluator.py:589-790``)."""

    def __init__(self, args, num_classes, iou_thresh, top_k):
        self.args = args
        self.top_k = list(top_k)
        self.num_classes = num_classes
        self.iou_thresh = iou_thresh
        self.num_connected_target = 0.0
        self.result_dict = {k: 0.0 for k in self.top_k}
        self.result_per_class = {k: torch.zeros(self.num_classes) for k in self.top_k}
        self.num_conn_target_per_class = torch.zeros(self.num_classes)
        self.feature_size = args["models"]["feature_size"]
        self.clear_data()

    def clear_data(self):
        self._l = {k: [] for k in ("which", "conf", "conn", "args3", "rel_t", "scat", "ocat", "sbox", "obox")}

    def accumulate(self, which_in_batch, relation_pred, relation_target, super_relation_pred, connectivity,
                   subject_cat_pred, object_cat_pred, subject_cat_target, object_cat_target,
                   subject_bbox_pred, object_bbox_pred, subject_bbox_target, object_bbox_target, iou_mask):
        a = self.args["models"]
        ng, npos = a["num_geometric"], a["num_possessive"]
        segs = [(0, ng), (ng, ng + npos), (ng + npos, relation_pred.shape[1])]
        mx = [torch.max(relation_pred[:, lo:hi], dim=1) for lo, hi in segs]
        conf = torch.max(torch.vstack([m[0] for m in mx]), dim=0)[0].clone()
        args3 = torch.stack([m[1] + lo for m, (lo, _) in zip(mx, segs)], dim=1)
        self.accumulate_candidates(which_in_batch, conf, args3, relation_target, connectivity, subject_cat_pred,
                                   object_cat_pred, subject_bbox_pred, object_bbox_pred, iou_mask)

    def accumulate_candidates(self, which_in_batch, conf, args3, relation_target, connectivity, subject_cat, object_cat,
                              subject_bbox, object_bbox, iou_mask=None):
        """conf [P] = max over the three candidate confidences, args3 [P,3] = their predicate ids."""
        conf = conf.clone()
        if iou_mask is not None:
            conf[~iou_mask.bool()] = -math.inf
        L = self._l
        L["which"].append(which_in_batch); L["conf"].append(conf); L["conn"].append(connectivity)
        L["args3"].append(args3.long()); L["rel_t"].append(relation_target)
        L["scat"].append(subject_cat); L["ocat"].append(object_cat); L["sbox"].append(subject_bbox); L["obox"].append(object_bbox)

    def compute(self, per_class=False):
        if len(self._l["which"]) > 0:
            cat = lambda k: (torch.vstack(self._l[k]) if self._l[k][0].dim() == 2 else torch.hstack(self._l[k]))
            which = cat("which")
            conf = cat("conf") + cat("conn")
            K = self.top_k[-1]
            images, keep_pos, keep_cnt, top, _, _ = rank_topk_device(conf, which, K)
            rel_all = cat("rel_t")
            sel = torch.nonzero(rel_all != -1).flatten()                       # every candidate row is also a (possible) target row
            if int(sel.numel()) > 0:
                row = torch.searchsorted(images, which[sel])
                tg = dict(row=row.int(), rel=rel_all[sel], scat=cat("scat")[sel], ocat=cat("ocat")[sel], sbox=cat("sbox")[sel], obox=cat("obox")[sel])
                cand = dict(scat=cat("scat"), ocat=cat("ocat"), pred=cat("args3"), sbox=cat("sbox"), obox=cat("obox"))
                hit = recall_hits(cand, keep_pos, keep_cnt, tg, K, self.feature_size, self.iou_thresh)
                rel_h, row_h, cnt_h = tg["rel"].cpu().numpy().astype(np.int64), row.cpu().numpy(), keep_cnt.cpu().numpy()
                num_target = np.bincount(row_h, minlength=len(cnt_h))           # connected targets per image
                for i in range(len(rel_h)):
                    ti = int(rel_h[i])
                    for k in self.top_k:
                        if hit[i] < cnt_h[row_h[i]] and hit[i] < max(k, num_target[row_h[i]]):
                            self.result_dict[k] += 1.0
                            if per_class:
                                self.result_per_class[k][ti] += 1.0
                    self.num_connected_target += 1.0
                    self.num_conn_target_per_class[ti] += 1.0
        recall_k = [self.result_dict[k] / max(self.num_connected_target, 1e-3) for k in self.top_k]
        recall_k_per_class = [self.result_per_class[k] / self.num_conn_target_per_class for k in self.top_k]
        mean_recall_k = [torch.nanmean(r) for r in recall_k_per_class]
        return recall_k, recall_k_per_class, mean_recall_k
