"""CPU oracle for the pairwise relation-prediction path.  TEST INFRASTRUCTURE ONLY.

This file is a literal, un-restructured float32 restatement (PyTorch CPU ops + plain Python
loops) of the reference's algorithm for the hot path.  It is the checker: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.  Nothing under
``scene_graph_commonsense_amd/`` imports it and the product path never falls back to it.

Parity pinning: ``tests/golden/make_golden.py`` imports the real reference from
``/root/reference`` (in the build container only) and stores its outputs for seeded inputs under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this restatement against every one
of those vectors (the reference itself has no tests or golden vectors).

Reference map (all paths relative to the reference repo):
  build_masks              train_test.py:164-169 (= :365-370, evaluate.py:111-116)
  super_class_multihot     utils.py:136-149      (quirk: only first and last list entries are set)
  conv_trunk               model.py:138-150
  concat_labels            model.py:152-168
  classifier_forward       model.py:170-186 (hierarchical), model.py:96-102 (flat)
  bayes_head               model.py:24-34
  overlap_filter           train_test.py:403-408
  class_weights            train_test.py:104-105, utils.py:258-268
  relationship_loss        train_utils.py:116-157, utils.py:28-35
  direction_step_loss      train_utils.py:64-94
  run_pair_loop            train_test.py:174-258 (train) / :373-437 (test), train_utils.py:160-196
  commonsense_step_loss    train_utils.py:36-62 (run_mode train_cs)
  supcon_hierar_loss       sup_contrast/losses.py:85-181 (SupConLossHierar, contrast_mode 'all')
  contrastive term         train_utils.py:28-29,96-99 (hidden/hidden_aug of connected pairs), train_test.py:260-273
  OracleEvaluator          evaluator.py:118-367, :568-583 (incl. accumulate_target :270-275 and the predcls=False branch of compute)
  compare_object_cat       utils.py:355-373
  match_target_sgd         utils.py:294-350
  run_sgdet_loop           evaluate.py:375-440 (SGDET: pair loop over PREDICTED objects, predcls=False evaluator feed)
  OracleEvaluatorTop3      evaluator.py:639-790
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- inputs
def build_masks(bbox: Tensor, feature_size: int) -> Tensor:
    """[n,4] boxes (x0,x1,y0,y1) -> [n,F,F] bool; int() truncation and slice clipping."""
    n = bbox.shape[0]
    mask = torch.zeros(n, feature_size, feature_size, dtype=torch.bool)
    for j in range(n):
        x0, x1, y0, y1 = (int(bbox[j][k]) for k in range(4))
        mask[j, y0:y1, x0:x1] = True
    return mask


def super_class_multihot(s_list: Sequence[Tensor], num_super: int) -> Tensor:
    """Multi-hot over super-classes; for a k-element list only element 0 and k-1 are set."""
    out = torch.zeros(len(s_list), num_super, dtype=torch.int64)
    for r, s in enumerate(s_list):
        out[r, int(s[0])] += 1
        k = len(s)
        if 2 <= k <= 4:
            out[r, int(s[k - 1])] += 1
    return out


def overlap_filter(mask_g: Tensor, mask_e: Tensor) -> Tensor:
    """[b,1,F,F] bool x2 -> [b] bool: sum(or)/sum(and), inf->0, >0."""
    s_or = torch.logical_or(mask_g, mask_e).sum(-1).sum(-1)
    s_and = torch.logical_and(mask_g, mask_e).sum(-1).sum(-1)
    r = (s_or / s_and).flatten()
    r[torch.isinf(r)] = 0
    return r > 0


# --------------------------------------------------------------------------- model
def routed_relu_pool(c: Tensor, code: Tensor) -> Tensor:
    """ReLU + 2x2 max-pool with the routing IMPOSED instead of computed: ``code`` [b,C,H/2,W/2] holds, per pooling window,
    which of its four elements (dy*2+dx) carries the value, or 4 when the ReLU kills the window.  Used only by the
    route-injected backward parity test: with the device's own routing decisions the oracle's autograd walks exactly the
    paths the device's backward walks, so what is left of the gradient difference is arithmetic, not routing."""
    b, C, H, W = c.shape
    win = c.reshape(b, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(b, C, H // 2, W // 2, 4)
    code = code.long()
    out = torch.gather(win, 4, code.clamp(max=3).unsqueeze(-1)).squeeze(-1)
    return out * (code < 4).to(out.dtype)


def conv_trunk(sd: Dict[str, Tensor], h_sub: Tensor, h_obj: Tensor, drop1: Optional[Tensor] = None,
               routes: Optional[dict] = None, capture: Optional[dict] = None) -> Tensor:
    """``routes`` (test hook, see ``routed_relu_pool``): dict(pool2 [b,512,16,16], pool3 [b,1024,8,8] window codes,
    relu1 [b,4096] 0/1 pass mask of fc1's ReLU).  ``capture`` (test hook): dict that receives the PRE-activations in front of the
    three routing decisions (conv2 / conv3 outputs before ReLU + max-pool, fc1 output before its ReLU), detached - how the tests
    show that every route the device takes differently lies within the forward tolerance of the decision boundary."""
    a = torch.tanh(F.conv2d(h_sub, sd["conv1_1.weight"], sd["conv1_1.bias"]))
    b = torch.tanh(F.conv2d(h_obj, sd["conv1_2.weight"], sd["conv1_2.bias"]))
    h = torch.cat((a, b), dim=1)
    h = F.conv2d(h, sd["conv2_1.weight"], sd["conv2_1.bias"], padding=1)
    if capture is not None:
        capture["pool2"] = h.detach()
    h = F.max_pool2d(F.relu(h), 2, 2) if routes is None else routed_relu_pool(h, routes["pool2"])
    h = F.conv2d(h, sd["conv3_1.weight"], sd["conv3_1.bias"], padding=1)
    if capture is not None:
        capture["pool3"] = h.detach()
    h = F.max_pool2d(F.relu(h), 2, 2) if routes is None else routed_relu_pool(h, routes["pool3"])
    h = h.reshape(h.shape[0], -1)
    h = F.linear(h, sd["fc1.weight"], sd["fc1.bias"])
    if capture is not None:
        capture["relu1"] = h.detach()
    h = F.relu(h) if routes is None else h * routes["relu1"]
    if drop1 is not None:          # injected dropout mask already scaled by 1/(1-p)
        h = h * drop1
    return h


def concat_labels(h: Tensor, c1: Tensor, c2: Tensor, s1, s2, num_classes: int, num_super: int) -> Tensor:
    o1 = F.one_hot(c1, num_classes=num_classes)
    o2 = F.one_hot(c2, num_classes=num_classes)
    if s1 is not None:
        m1 = super_class_multihot(s1, num_super)
        m2 = super_class_multihot(s2, num_super)
        return torch.cat((h, o1, o2, m1, m2), dim=1)
    return torch.cat((h, o1, o2), dim=1)


def bayes_head(sd: Dict[str, Tensor], p: Tensor, T=(1.0, 1.0, 1.0)):
    sup = F.log_softmax(F.linear(p, sd["fc5.weight"], sd["fc5.bias"]), dim=1)
    rels = []
    for k, name in enumerate(("fc3_1", "fc3_2", "fc3_3")):
        r = F.linear(p, sd[name + ".weight"], sd[name + ".bias"])
        rels.append(F.log_softmax(r / T[k], dim=1) + sup[:, k].view(-1, 1))
    return rels[0], rels[1], rels[2], sup


def classifier_forward(sd: Dict[str, Tensor], h_sub: Tensor, h_obj: Tensor, c1: Tensor, c2: Tensor, s1, s2,
                       num_classes: int = 150, num_super: int = 17, hierarchical: bool = True,
                       drop1: Optional[Tensor] = None, drop2: Optional[Tensor] = None, T=(1.0, 1.0, 1.0),
                       routes: Optional[dict] = None, capture: Optional[dict] = None):
    """Hierarchical: (rel1, rel2, rel3, super, connectivity[b,1], hidden[b,512]).
    Flat: (relation[b,R] raw logits, connectivity[b,1], hidden).
    ``drop1``/``drop2``: injected dropout masks (already scaled); ``routes``: injected ReLU/max-pool routing (tests only);
    ``capture``: see ``conv_trunk`` (+ ``relu2``: fc2's output before its ReLU)."""
    h = conv_trunk(sd, h_sub, h_obj, drop1, routes, capture)
    hc = concat_labels(h, c1, c2, s1, s2, num_classes, num_super)
    p = F.linear(hc, sd["fc2.weight"], sd["fc2.bias"])
    if capture is not None:
        capture["relu2"] = p.detach()
    p = F.relu(p) if routes is None else p * routes["relu2"]
    if drop2 is not None:
        p = p * drop2
    conn = F.linear(p, sd["fc4.weight"], sd["fc4.bias"])
    if hierarchical:
        r1, r2, r3, sup = bayes_head(sd, p, T)
        return r1, r2, r3, sup, conn, p
    rel = F.linear(p, sd["fc3.weight"], sd["fc3.bias"])
    return rel, conn, p


# --------------------------------------------------------------------------- loss
def class_weights(counts: Tensor) -> Tensor:
    counts = counts.float()
    return 1 - counts / torch.sum(counts)


def relationship_loss(relation: Tensor, super_relation: Optional[Tensor], connected: Tensor, target_row: Tensor,
                      weights: Tensor, ng: int, npos: int, hierarchical: bool = True) -> Tensor:
    tgt = target_row[connected]
    if not hierarchical:
        return F.cross_entropy(relation[connected], tgt, weight=weights)
    sup_t = tgt.clone()
    sup_t[tgt < ng] = 0
    sup_t[torch.logical_and(tgt >= ng, tgt < ng + npos)] = 1
    sup_t[tgt >= ng + npos] = 2
    loss = F.nll_loss(super_relation[connected], sup_t)
    bounds = [(0, ng), (ng, ng + npos), (ng + npos, relation.shape[1])]
    for lo, hi in bounds:
        sel = torch.nonzero(torch.logical_and(tgt >= lo, tgt < hi)).flatten()
        if sel.numel() > 0:
            loss = loss + F.nll_loss(relation[:, lo:hi][connected][sel], tgt[sel] - lo, weight=weights[lo:hi])
    return loss


def direction_step_loss(relation, super_relation, conn, rel_row, dir_row, first_direction: bool, weights,
                        ng, npos, hierarchical=True, lambda_not_connected=1.0):
    """One (graph_iter, edge_iter, direction) step -> (loss_relationship, loss_connectivity)."""
    flag = 1 if first_direction else 0
    not_connected = torch.where(dir_row != flag)[0]
    connected = torch.where(dir_row == flag)[0]
    c = conn[:, 0]
    loss_conn = 0.0
    if len(not_connected) > 0:     # BCE mean over an empty set is NaN -> 0.0 in the reference
        loss_conn = lambda_not_connected * F.binary_cross_entropy_with_logits(
            c[not_connected], torch.zeros(len(not_connected)))
    loss_rel = 0.0
    if len(connected) > 0:
        # overwrite (not add): the not-connected term is dropped whenever the step has a connected pair
        loss_conn = F.binary_cross_entropy_with_logits(c[connected], torch.ones(len(connected)))
        loss_rel = relationship_loss(relation, super_relation, connected, rel_row, weights, ng, npos, hierarchical)
    return loss_rel, loss_conn, connected, not_connected


def commonsense_step_loss(relation: Tensor, cat_sub: Tensor, cat_obj: Tensor, aligned, violated, ng: int, npos: int,
                          hierarchical: bool = True, lambda_weak: float = 0.1, lambda_strong: float = 10.0):
    """Penalty on the most confident prediction of every super-category when its (subject, predicate, object) triplet is
    not in the aligned set (weak) / is in the violated set (strong); means over the flagged candidates of the step."""
    if hierarchical:
        segs = [(0, ng), (ng, ng + npos), (ng + npos, relation.shape[1])]
        probs = torch.hstack([torch.max(F.softmax(relation[:, lo:hi], dim=1), dim=1)[0] for lo, hi in segs])
        pred = torch.hstack([torch.argmax(relation[:, lo:hi], dim=1) + lo for lo, hi in segs])
        trip = torch.hstack((cat_sub.repeat(3).unsqueeze(1), pred.unsqueeze(1), cat_obj.repeat(3).unsqueeze(1)))
    else:
        probs = torch.max(F.softmax(relation, dim=1), dim=1)[0]
        pred = torch.argmax(relation, dim=1)
        trip = torch.hstack((cat_sub.unsqueeze(1), pred.unsqueeze(1), cat_obj.unsqueeze(1)))
    keys = [tuple(t.tolist()) for t in trip]
    not_yes = torch.tensor([k not in aligned for k in keys], dtype=torch.bool)
    in_no = torch.tensor([k in violated for k in keys], dtype=torch.bool)
    loss = 0.0
    if probs[not_yes].numel() > 0:
        loss = loss + lambda_weak * probs[not_yes].mean()
    if probs[in_no].numel() > 0:
        loss = loss + lambda_strong * probs[in_no].mean()
    return loss


def supcon_hierar_loss(features: Tensor, labels: Tensor, temperature: float = 0.07, base_temperature: float = 0.07) -> Tensor:
    """features [M, 2, D] (original and augmented view), labels [M].  Positives: same label (other view included);
    denominator: every other sample under the same parent super-category (hard-coded 15 / 26 boundaries)."""
    M = features.shape[0]
    lab = labels.contiguous().view(-1, 1)
    parent = lab.clone()
    parent[lab < 15] = 0
    parent[(lab >= 15) & (lab < 26)] = 1
    parent[lab >= 26] = 2
    mask_same_parent = torch.eq(parent, parent.T).float()
    mask = torch.eq(lab, lab.T).float()
    contrast = torch.cat(torch.unbind(features, dim=1), dim=0)            # [2M, D]: view 0 rows, then view 1 rows
    a = torch.div(torch.matmul(contrast, contrast.T), temperature)
    logits = a - torch.max(a, dim=1, keepdim=True)[0].detach()
    mask = mask.repeat(2, 2)
    mask_same_parent = mask_same_parent.repeat(2, 2)
    logits_mask = torch.ones_like(mask)
    logits_mask.fill_diagonal_(0)
    mask = mask * logits_mask
    logits_mask = logits_mask * mask_same_parent
    exp_logits = torch.exp(logits) * logits_mask
    log_prob = logits - torch.log(exp_logits.sum(1, keepdim=True) + 1e-7)
    mean_log_prob_pos = (mask * log_prob).sum(1) / (mask.sum(1) + 1e-7)
    loss = -(temperature / base_temperature) * mean_log_prob_pos
    return loss.view(2, M).mean()


# --------------------------------------------------------------------------- pair loop
def run_pair_loop(sd: Dict[str, Tensor], batch, cfg, mode: str = "eval", evaluator=None, evaluator_top3=None,
                  weights: Optional[Tensor] = None, lambda_connectivity: float = 0.1,
                  lambda_not_connected: float = 1.0, overlap_filtering: Optional[bool] = None,
                  max_steps: Optional[int] = None, step_filter=None, image_feature_aug: Optional[Tensor] = None,
                  lambda_contrast: float = 1.0, commonsense=None, lambda_commonsense: float = 1.0, call_hook=None,
                  step_loss_hook=None):
    """The reference's nested (graph_iter, edge_iter) x 2-direction loop.

    ``call_hook(t, b)`` (tests only): for the t-th classifier call (0-based, direction-steps in loop order, b rows) returns a
    dict with any of ``drop1`` [b,4096] / ``drop2`` [b,512] (dropout masks of ``model.py:120-121,149,175`` injected instead
    of drawn - how the training-mode device path is compared), ``routes`` (see ``conv_trunk``) and ``routes_aug`` (the same for
    the augmented view of the contrastive branch; only the rows of connected pairs matter).

    ``step_loss_hook(k, own)`` (tests only, memory): instead of the literal running sums - ``losses += run_rel + lambda_c * run_conn``
    after every direction-step, which keeps the autograd graph of EVERY call alive until the final ``backward()`` (17 GB for 128 calls
    of b = 8) - the k-th direction-step's OWN loss ``own = lr_ + lambda_c * lc_ (+ lambda_cs * cs_)`` is handed to the caller, who
    weights it by the number of running-sum additions it takes part in (T - k for T direction-steps in all) and back-propagates it at
    once.  The same total and the same gradients (``tests/test_oracle_golden.py`` holds the two forms against each other); ``losses``
    is then None.

    mode 'eval' mirrors ``testing()`` (overlap filter on, steps with no overlapping image skipped),
    mode 'train' mirrors ``training()`` (iou_mask all ones, loss with the running-sum quirk; dropout
    only through masks injected by ``call_hook``).
    Returns dict(records=[per direction-step dict], losses=scalar tensor or None).
    """
    Fs = cfg.feature_size
    B = batch.image_feature.shape[0]
    hier = cfg.hierarchical
    ng, npos = cfg.num_geometric, cfg.num_possessive
    if overlap_filtering is None:
        overlap_filtering = (mode == "eval")
    masks = [build_masks(b, Fs) for b in batch.bbox]
    n_obj = torch.as_tensor([len(m) for m in masks])
    relations_target, direction_target = [], []
    for g in range(int(n_obj.max()) - 1):
        keep = torch.nonzero(n_obj - 1 > g).view(-1)
        relations_target.append(torch.vstack([batch.relationships[i][g] for i in keep]).T)
        direction_target.append(torch.vstack([batch.subj_or_obj[i][g] for i in keep]).T)

    records = []
    losses = 0.0
    run_rel, run_conn, run_cs = 0.0, 0.0, 0.0      # commonsense = (aligned dict/set, violated dict/set) for run_mode train_cs
    nsteps = 0
    contrast = image_feature_aug is not None and mode == "train"
    hid_acc = [[] for _ in range(B)]       # per image: [2,512] stacks of (hidden, hidden_aug) of connected pairs
    lab_acc = [[] for _ in range(B)]
    for g in range(int(n_obj.max())):
        keep = torch.nonzero(n_obj > g).view(-1)
        gm = torch.stack([masks[i][g].unsqueeze(0) for i in keep])
        h_graph = torch.cat((batch.image_feature[keep] * gm, batch.image_depth[keep] * gm), dim=1)
        h_graph_aug = torch.cat((image_feature_aug[keep] * gm, batch.image_depth[keep] * gm), dim=1) if contrast else None
        cat_g = torch.tensor([int(batch.categories[i][g]) for i in keep])
        sp_g = [batch.super_categories[i][g] for i in keep] if batch.super_categories is not None else None
        bb_g = torch.stack([batch.bbox[i][g] for i in keep])
        for e in range(g):
            if max_steps is not None and nsteps >= max_steps:
                break
            if step_filter is not None and not step_filter(g, e):
                continue
            em = torch.stack([masks[i][e].unsqueeze(0) for i in keep])
            h_edge = torch.cat((batch.image_feature[keep] * em, batch.image_depth[keep] * em), dim=1)
            h_edge_aug = torch.cat((image_feature_aug[keep] * em, batch.image_depth[keep] * em), dim=1) if contrast else None
            cat_e = torch.tensor([int(batch.categories[i][e]) for i in keep])
            sp_e = [batch.super_categories[i][e] for i in keep] if batch.super_categories is not None else None
            bb_e = torch.stack([batch.bbox[i][e] for i in keep])
            if overlap_filtering:
                iou_mask = overlap_filter(gm, em)
                if torch.sum(iou_mask) == 0:
                    continue
            else:
                iou_mask = torch.ones(len(keep), dtype=torch.bool)
            nsteps += 1
            for first in (True, False):
                hs, ho = (h_graph, h_edge) if first else (h_edge, h_graph)
                cs, co = (cat_g, cat_e) if first else (cat_e, cat_g)
                ss, so = (sp_g, sp_e) if first else (sp_e, sp_g)
                bs, bo = (bb_g, bb_e) if first else (bb_e, bb_g)
                inj = call_hook(len(records), len(keep)) if call_hook is not None else {}
                out = classifier_forward(sd, hs, ho, cs, co, ss, so, cfg.num_classes, cfg.num_super_classes, hier,
                                         drop1=inj.get("drop1"), drop2=inj.get("drop2"), routes=inj.get("routes"),
                                         **({"capture": inj["capture"]} if "capture" in inj else {}))
                if hier:
                    r1, r2, r3, sup, conn, hidden = out
                    relation = torch.cat((r1, r2, r3), dim=1)
                else:
                    relation, conn, hidden = out
                    sup = None
                rel_row = relations_target[g - 1][e]
                dir_row = direction_target[g - 1][e]
                flag = 1 if first else 0
                not_connected = torch.where(dir_row != flag)[0]
                directed = rel_row.clone()
                directed[not_connected] = -1
                if mode == "train":
                    lr_, lc_, connected, _ = direction_step_loss(relation, sup, conn, rel_row, dir_row, first, weights, ng,
                                                                 npos, hier, lambda_not_connected)
                    cs_ = commonsense_step_loss(relation, cs, co, commonsense[0], commonsense[1], ng, npos, hier) if commonsense is not None else 0.0
                    if step_loss_hook is not None:
                        step_loss_hook(len(records), lr_ + lambda_connectivity * lc_ + lambda_commonsense * cs_)
                    else:
                        run_rel = run_rel + lr_
                        run_conn = run_conn + lc_
                        if commonsense is not None:
                            run_cs = run_cs + cs_
                        losses = losses + run_rel + lambda_connectivity * run_conn + lambda_commonsense * run_cs
                    if contrast and len(connected) > 0:
                        hsa, hoa = (h_graph_aug, h_edge_aug) if first else (h_edge_aug, h_graph_aug)
                        ra = inj.get("routes_aug")          # test hook: the device's routing of the augmented trunk (rows of this call)
                        h_aug = conv_trunk(sd, hsa, hoa, routes=ra)
                        hca = concat_labels(h_aug, cs, co, ss, so, cfg.num_classes, cfg.num_super_classes)
                        pa = F.linear(hca, sd["fc2.weight"], sd["fc2.bias"])
                        hidden_aug = F.relu(pa) if ra is None else pa * ra["relu2"]
                        for idx in connected:
                            bi = int(keep[idx])
                            hid_acc[bi].append(torch.stack((hidden[idx], hidden_aug[idx])))
                            lab_acc[bi].append(rel_row[idx])
                logsig = torch.log(torch.sigmoid(conn[:, 0]))
                if evaluator is not None:
                    evaluator.accumulate(keep, relation.detach(), directed, None if sup is None else sup.detach(),
                                         logsig.detach(), cs, co, cs, co, bs, bo, bs, bo, iou_mask)
                if evaluator_top3 is not None and hier:
                    evaluator_top3.accumulate(keep, relation.detach(), directed, sup.detach(), logsig.detach(),
                                              cs, co, cs, co, bs, bo, bs, bo, iou_mask)
                records.append(dict(g=g, e=e, first=first, keep=keep.clone(), relation=relation.detach(),
                                    super_relation=None if sup is None else sup.detach(),
                                    connectivity=conn.detach()[:, 0], hidden=hidden.detach(),
                                    iou_mask=iou_mask.clone(), target=directed))
    loss_contrast = None
    if contrast and any(len(x) > 0 for x in hid_acc):
        feats = torch.cat([torch.stack(x) for x in hid_acc if len(x) > 0], dim=0)
        labs = torch.cat([torch.stack(x) for x in lab_acc if len(x) > 0], dim=0)
        temp = supcon_hierar_loss(feats, labs)
        loss_contrast = 0.0 if torch.isnan(temp) else lambda_contrast * temp
        losses = losses + lambda_contrast * loss_contrast          # lambda applied twice, as in train_test.py:270-273
    return dict(records=records, losses=losses if (mode == "train" and step_loss_hook is None) else None, loss_contrast=loss_contrast)


# --------------------------------------------------------------------------- SGDET (predicted objects)
def compare_object_cat(pred_cat, target_cat) -> bool:
    """utils.py:355-373: object classes that count as the same label when categories are predicted."""
    equiv = [[1, 5, 11, 23, 38, 44, 121, 124, 148, 149], [0, 50], [92, 137]]
    unsymm_equiv = {123: [14, 63, 95, 87, 123], 108: [89, 102, 67, 72, 71, 81, 96, 105, 90, 111, 108], 60: [145, 106, 142, 144, 77, 60]}
    pred_cat, target_cat = int(pred_cat), int(target_cat)
    if pred_cat == target_cat:
        return True
    for group in equiv:
        if pred_cat in group and target_cat in group:
            return True
    for key in unsymm_equiv:
        if pred_cat == key and target_cat in unsymm_equiv[key]:
            return True
        elif target_cat == key and pred_cat in unsymm_equiv[key]:
            return True
    return False


def match_target_sgd(relationships, subj_or_obj, categories_target, bbox_target):
    """utils.py:294-350: the ground-truth triplets of every image as flat per-image tensors (None when an image has none):
    subject/object categories, subject/object boxes [t,4], predicate; direction flag 1 -> (graph, edge), 0 -> (edge, graph).
    Quirk kept: graph_iter runs over range(len(relationships[b])) = 0..n-2, so relations whose graph object is the last object
    of the image are never collected (the list has n-1 rows for g = 1..n-1)."""
    cs_l, co_l, bs_l, bo_l, rt_l = [], [], [], [], []
    for b in range(len(relationships)):
        cs, co, bs, bo, rt = [], [], [], [], []
        for g in range(len(relationships[b])):
            for e in range(g):
                flag = subj_or_obj[b][g - 1][e]
                if flag == 1:
                    s_i, o_i = g, e
                elif flag == 0:
                    s_i, o_i = e, g
                else:
                    continue
                cs.append(categories_target[b][s_i]); co.append(categories_target[b][o_i])
                bs.append(bbox_target[b][s_i]); bo.append(bbox_target[b][o_i]); rt.append(relationships[b][g - 1][e])
        if rt:
            cs_l.append(torch.stack([torch.as_tensor(x).reshape(()) for x in cs])); co_l.append(torch.stack([torch.as_tensor(x).reshape(()) for x in co]))
            bs_l.append(torch.stack(bs).view(-1, 4)); bo_l.append(torch.stack(bo).view(-1, 4))
            rt_l.append(torch.stack([torch.as_tensor(x).reshape(()) for x in rt]))
        else:
            cs_l.append(None); co_l.append(None); bs_l.append(None); bo_l.append(None); rt_l.append(None)
    return cs_l, co_l, bs_l, bo_l, rt_l


def run_sgdet_loop(sd: Dict[str, Tensor], image_feature: Tensor, image_depth: Tensor, categories_pred, cat_pred_confidence,
                   bbox_pred, super_categories_pred, cfg, evaluator):
    """evaluate.py:375-440: the (graph_iter, edge_iter) x 2-direction loop over the PREDICTED objects of every image (one entry
    per image in the lists), overlap filter on, evaluator fed with predcls=False and the category confidences."""
    Fs = cfg.feature_size
    hier = cfg.hierarchical
    masks = [build_masks(b, Fs) for b in bbox_pred]
    n_obj = torch.as_tensor([len(m) for m in masks])
    for g in range(int(n_obj.max())):
        keep = torch.nonzero(n_obj > g).view(-1)
        gm = torch.stack([masks[i][g].unsqueeze(0) for i in keep])
        h_graph = torch.cat((image_feature[keep] * gm, image_depth[keep] * gm), dim=1)
        cat_g = torch.tensor([int(categories_pred[i][g]) for i in keep])
        bb_g = torch.stack([bbox_pred[i][g] for i in keep])
        cf_g = torch.hstack([cat_pred_confidence[i][g] for i in keep])
        for e in range(g):
            em = torch.stack([masks[i][e].unsqueeze(0) for i in keep])
            h_edge = torch.cat((image_feature[keep] * em, image_depth[keep] * em), dim=1)
            cat_e = torch.tensor([int(categories_pred[i][e]) for i in keep])
            bb_e = torch.stack([bbox_pred[i][e] for i in keep])
            cf_e = torch.hstack([cat_pred_confidence[i][e] for i in keep])
            iou_mask = overlap_filter(gm, em)
            if torch.sum(iou_mask) == 0:
                continue
            sp_g = [super_categories_pred[i][g] for i in keep] if super_categories_pred is not None else None
            sp_e = [super_categories_pred[i][e] for i in keep] if super_categories_pred is not None else None
            for first in (True, False):
                hs, ho = (h_graph, h_edge) if first else (h_edge, h_graph)
                cs, co = (cat_g, cat_e) if first else (cat_e, cat_g)
                ss, so = (sp_g, sp_e) if first else (sp_e, sp_g)
                bs, bo = (bb_g, bb_e) if first else (bb_e, bb_g)
                fs, fo_ = (cf_g, cf_e) if first else (cf_e, cf_g)
                out = classifier_forward(sd, hs, ho, cs, co, ss, so, cfg.num_classes, cfg.num_super_classes, hier)
                if hier:
                    r1, r2, r3, sup, conn, _ = out
                    relation = torch.cat((r1, r2, r3), dim=1)
                else:
                    relation, conn, _ = out
                    sup = None
                evaluator.accumulate(keep, relation.detach(), None, None if sup is None else sup.detach(),
                                     torch.log(torch.sigmoid(conn[:, 0])).detach(), cs, co, None, None, bs, bo, None, None,
                                     iou_mask, False, fs, fo_)


# --------------------------------------------------------------------------- evaluator
def grid_iou(bt, bp, feature_size: int) -> float:
    """Rectangles rasterised with int() truncation on the FxF grid; union==0 -> 0."""
    mp = torch.zeros(feature_size, feature_size)
    mp[int(bp[2]):int(bp[3]), int(bp[0]):int(bp[1])] = 1
    mt = torch.zeros(feature_size, feature_size)
    mt[int(bt[2]):int(bt[3]), int(bt[0]):int(bt[1])] = 1
    inter = torch.sum(torch.logical_and(mt, mp))
    union = torch.sum(torch.logical_or(mt, mp))
    return 0 if union == 0 else float(inter) / float(union)


class OracleEvaluator:
    """Recall@K with three candidates per directed pair (hierarchical) or one (flat).

    Deviation from the reference, on purpose and documented in DESIGN.md: the per-image ranking
    uses a *stable* descending sort (ties keep append order); the reference's ``torch.argsort`` is
    unstable on ties, so index parity is only defined under a stable order.
    """

    def __init__(self, cfg, top_k=(20, 50, 100), iou_thresh=0.5, zero_shot_triplets=None, train_triplets=None):
        self.cfg = cfg
        self.hier = cfg.hierarchical
        self.top_k = list(top_k)
        self.iou_thresh = iou_thresh
        self.R = cfg.num_relations
        self.zero_shot = set(zero_shot_triplets) if zero_shot_triplets is not None else None
        self.train_triplets = train_triplets
        self.result_dict = {k: 0.0 for k in self.top_k}
        self.result_per_class = {k: torch.zeros(self.R) for k in self.top_k}
        self.num_connected_target = 0.0
        self.num_conn_target_per_class = torch.zeros(self.R)
        self.result_dict_zs = {k: 0.0 for k in self.top_k}
        self.result_per_class_zs = {k: torch.zeros(self.R) for k in self.top_k}
        self.num_connected_target_zs = 0.0
        self.num_conn_target_per_class_zs = torch.zeros(self.R)
        self.clear_data()

    def clear_data(self):
        self.which, self.conf, self.conn, self.pred = [], [], [], []
        self.scat, self.ocat, self.sbox, self.obox = [], [], [], []
        self.which_t, self.rel_t, self.scat_t, self.ocat_t, self.sbox_t, self.obox_t = [], [], [], [], [], []
        self.last_sorted = {}
        self.targets_by_image = None

    def accumulate_target(self, relation_target, subject_cat_target, object_cat_target, subject_bbox_target, object_bbox_target):
        """evaluator.py:270-275: per-image lists (entries may be None)."""
        self.targets_by_image = (relation_target, subject_cat_target, object_cat_target, subject_bbox_target, object_bbox_target)

    def accumulate(self, which_in_batch, relation_pred, relation_target, super_relation_pred, connectivity,
                   subject_cat_pred, object_cat_pred, subject_cat_target, object_cat_target,
                   subject_bbox_pred, object_bbox_pred, subject_bbox_target, object_bbox_target, iou_mask,
                   predcls=True, cat_subject_confidence=None, cat_object_confidence=None):
        ng, npos = self.cfg.num_geometric, self.cfg.num_possessive
        if self.hier:
            segs = [(0, ng), (ng, ng + npos), (ng + npos, relation_pred.shape[1])]
            conf = torch.hstack([torch.max(relation_pred[:, lo:hi], dim=1)[0] for lo, hi in segs])
            pred = torch.hstack([torch.argmax(relation_pred[:, lo:hi], dim=1) + lo for lo, hi in segs])
            rep = 3
        else:
            conf = torch.max(relation_pred, dim=1)[0].clone()
            pred = torch.argmax(relation_pred, dim=1)
            rep = 1
        if not predcls:
            conf = conf + (cat_subject_confidence + cat_object_confidence).repeat(rep)
        conf = conf.clone()
        conf[~iou_mask.repeat(rep)] = -math.inf
        self.which.append(which_in_batch.repeat(rep)); self.conf.append(conf); self.pred.append(pred)
        self.conn.append(connectivity.repeat(rep))
        self.scat.append(subject_cat_pred.repeat(rep)); self.ocat.append(object_cat_pred.repeat(rep))
        self.sbox.append(subject_bbox_pred.repeat(rep, 1)); self.obox.append(object_bbox_pred.repeat(rep, 1))
        if predcls:
            self.which_t.append(which_in_batch); self.rel_t.append(relation_target)
            self.scat_t.append(subject_cat_target); self.ocat_t.append(object_cat_target)
            self.sbox_t.append(subject_bbox_target); self.obox_t.append(object_bbox_target)

    def flat_state(self):
        cat = torch.hstack
        d = dict(which=cat(self.which), conf=cat(self.conf), conn=cat(self.conn), pred=cat(self.pred),
                 scat=cat(self.scat), ocat=cat(self.ocat), sbox=torch.vstack(self.sbox), obox=torch.vstack(self.obox))
        if self.targets_by_image is None:
            d.update(which_t=cat(self.which_t), rel_t=cat(self.rel_t), scat_t=cat(self.scat_t), ocat_t=cat(self.ocat_t),
                     sbox_t=torch.vstack(self.sbox_t), obox_t=torch.vstack(self.obox_t))
        return d

    def compute(self, per_class=False, predcls=True):
        if len(self.which) == 0:
            return self._ratios()
        s = self.flat_state()
        conf = s["conf"] + s["conn"]
        Fs = self.cfg.feature_size
        for image in torch.unique(s["which"]):
            cur = s["which"] == image
            if self.targets_by_image is not None and self.targets_by_image[0][int(image)] is None:
                continue
            c = conf[cur]
            order = torch.sort(c, descending=True, stable=True)[1]
            this_k = min(self.top_k[-1], len(c))
            keep = order[:this_k]
            self.last_sorted[int(image)] = keep.clone()
            pred, scat, ocat = s["pred"][cur][keep], s["scat"][cur][keep], s["ocat"][cur][keep]
            sbox, obox = s["sbox"][cur][keep], s["obox"][cur][keep]
            if self.targets_by_image is None:
                cur_t = s["which_t"] == image
                rel_t, scat_t, ocat_t = s["rel_t"][cur_t], s["scat_t"][cur_t], s["ocat_t"][cur_t]
                sbox_t, obox_t = s["sbox_t"][cur_t], s["obox_t"][cur_t]
            else:
                rel_t, scat_t, ocat_t, sbox_t, obox_t = (x[int(image)] for x in self.targets_by_image)
            for i in range(len(rel_t)):
                if rel_t[i] == -1:
                    continue
                trip = "%d_%d_%d" % (int(scat_t[i]), int(rel_t[i]), int(ocat_t[i]))
                is_zs = self.zero_shot is not None and trip in self.zero_shot
                for j in range(this_k):
                    if predcls:
                        label_ok = scat_t[i] == scat[j] and ocat_t[i] == ocat[j]
                    else:
                        label_ok = compare_object_cat(scat_t[i], scat[j]) and compare_object_cat(ocat_t[i], ocat[j])
                    if label_ok:
                        if grid_iou(sbox_t[i], sbox[j], Fs) >= self.iou_thresh and \
                                grid_iou(obox_t[i], obox[j], Fs) >= self.iou_thresh:
                            if rel_t[i] == pred[j]:
                                for k in self.top_k:
                                    if j >= k:
                                        continue
                                    self.result_dict[k] += 1.0
                                    if per_class:
                                        self.result_per_class[k][rel_t[i]] += 1.0
                                    if is_zs:
                                        self.result_dict_zs[k] += 1.0
                                        if per_class:
                                            self.result_per_class_zs[k][rel_t[i]] += 1.0
                                break
                self.num_connected_target += 1.0
                self.num_conn_target_per_class[rel_t[i]] += 1.0
                if is_zs:
                    self.num_connected_target_zs += 1.0
                    self.num_conn_target_per_class_zs[rel_t[i]] += 1.0
        return self._ratios()

    def _ratios(self):
        rk = [self.result_dict[k] / max(self.num_connected_target, 1e-3) for k in self.top_k]
        rpc = [self.result_per_class[k] / self.num_conn_target_per_class for k in self.top_k]
        mrk = [torch.nanmean(r) for r in rpc]
        rk_zs = rpc_zs = mrk_zs = None
        if self.cfg.dataset == "vg":
            rk_zs = [self.result_dict_zs[k] / max(self.num_connected_target_zs, 1e-3) for k in self.top_k]
            rpc_zs = [self.result_per_class_zs[k] / self.num_conn_target_per_class_zs for k in self.top_k]
            mrk_zs = [torch.nanmean(r) for r in rpc_zs]
        return rk, rpc, mrk, rk_zs, rpc_zs, mrk_zs


class OracleEvaluatorTop3:
    """Recall@K*: one candidate per directed pair; a hit if the target equals any of the three
    per-super-category argmaxes; counted for k with ``j < max(k, num_target)``."""

    def __init__(self, cfg, top_k=(20, 50, 100), iou_thresh=0.5):
        self.cfg = cfg
        self.top_k = list(top_k)
        self.iou_thresh = iou_thresh
        self.R = cfg.num_relations
        self.result_dict = {k: 0.0 for k in self.top_k}
        self.result_per_class = {k: torch.zeros(self.R) for k in self.top_k}
        self.num_connected_target = 0.0
        self.num_conn_target_per_class = torch.zeros(self.R)
        self.clear_data()

    def clear_data(self):
        self.which, self.conf, self.conn, self.rel, self.rel_t = [], [], [], [], []
        self.scat, self.ocat, self.sbox, self.obox = [], [], [], []

    def accumulate(self, which_in_batch, relation_pred, relation_target, super_relation_pred, connectivity,
                   subject_cat_pred, object_cat_pred, subject_cat_target, object_cat_target,
                   subject_bbox_pred, object_bbox_pred, subject_bbox_target, object_bbox_target, iou_mask):
        ng, npos = self.cfg.num_geometric, self.cfg.num_possessive
        segs = [(0, ng), (ng, ng + npos), (ng + npos, relation_pred.shape[1])]
        conf = torch.max(torch.vstack([torch.max(relation_pred[:, lo:hi], dim=1)[0] for lo, hi in segs]), dim=0)[0]
        conf = conf.clone()
        conf[~iou_mask] = -math.inf
        self.which.append(which_in_batch); self.conf.append(conf); self.conn.append(connectivity)
        self.rel.append(relation_pred); self.rel_t.append(relation_target)
        self.scat.append(subject_cat_pred); self.ocat.append(object_cat_pred)
        self.sbox.append(subject_bbox_pred); self.obox.append(object_bbox_pred)

    def compute(self, per_class=False):
        if len(self.which) > 0:
            ng, npos = self.cfg.num_geometric, self.cfg.num_possessive
            which, conf = torch.hstack(self.which), torch.hstack(self.conf) + torch.hstack(self.conn)
            rel, rel_t = torch.vstack(self.rel), torch.hstack(self.rel_t)
            scat, ocat = torch.hstack(self.scat), torch.hstack(self.ocat)
            sbox, obox = torch.vstack(self.sbox), torch.vstack(self.obox)
            Fs = self.cfg.feature_size
            for image in torch.unique(which):
                cur = which == image
                order = torch.sort(conf[cur], descending=True, stable=True)[1]
                this_k = min(self.top_k[-1], int(cur.sum()))
                keep = order[:this_k]
                r_c, t_c = rel[cur], rel_t[cur]
                sc, oc, sb, ob = scat[cur], ocat[cur], sbox[cur], obox[cur]
                num_target = int(torch.sum(t_c != -1))
                for i in range(len(t_c)):
                    if t_c[i] == -1:
                        continue
                    for jj in range(this_k):
                        j = int(keep[jj])
                        if sc[i] == sc[j] and oc[i] == oc[j] and \
                                grid_iou(sb[i], sb[j], Fs) >= self.iou_thresh and \
                                grid_iou(ob[i], ob[j], Fs) >= self.iou_thresh:
                            a1 = int(torch.argmax(r_c[j][:ng]))
                            a2 = int(torch.argmax(r_c[j][ng:ng + npos])) + ng
                            a3 = int(torch.argmax(r_c[j][ng + npos:])) + ng + npos
                            if int(t_c[i]) in (a1, a2, a3):
                                for k in self.top_k:
                                    if jj >= max(k, num_target):
                                        continue
                                    self.result_dict[k] += 1.0
                                    if per_class:
                                        self.result_per_class[k][t_c[i]] += 1.0
                                break
                    self.num_connected_target += 1.0
                    self.num_conn_target_per_class[t_c[i]] += 1.0
        rk = [self.result_dict[k] / max(self.num_connected_target, 1e-3) for k in self.top_k]
        rpc = [self.result_per_class[k] / self.num_conn_target_per_class for k in self.top_k]
        return rk, rpc, [torch.nanmean(r) for r in rpc]
