"""Per-step drop-ins for the reference's ``train_utils.py``: ``evaluate_one_direction`` (``train_utils.py:160-196``),
``train_one_direction`` (``:21-113``), ``calculate_losses_on_relationships`` (``:116-157``), ``process_image_features``
(``:9-18``).

They let the reference's own ``graph_iter`` / ``edge_iter`` loops (``train_test.py:189-258,386-437``, ``evaluate.py:132-183``)
run unchanged on top of the HIP-backed classifier and evaluators: same positional arguments, same return tuples.  The
classifier call inside is ONE autograd node per step (``model._PairStepFunction``), so ``losses.backward()`` after the loops
trains the model exactly as in the reference.  The per-step loss arithmetic below is a handful of ``[b]``-sized torch ops on
the outputs (b <= batch size); the fused equivalents that do all of it in kernels, once per minibatch, are
``pair_loop.train_minibatch`` / ``evaluate_minibatch`` - use those for throughput, these for drop-in compatibility.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def process_image_features(args, images, detr, rank):
    """``[B,256,32,32]`` encoder features of a minibatch (``train_utils.py:9-18``).  ``detr`` is either the reference's DDP-wrapped
    DETR-101 (``utils.build_detr101``) or this package's ``detr.build_detr101`` (its backbone / input_proj / transformer.encoder are
    called as the reference does) or any
    callable mapping the stacked images to the feature map (precomputed features: pass ``lambda x: x``)."""
    images = torch.stack(list(images)).to(rank)
    core = getattr(detr, "module", detr)
    half = args["models"].get("feature_encoder_dtype")         # "bf16": the frozen extractor under bf16 autocast (detr.DETR.encode); default f32 as the reference
    if half in ("bf16", "bfloat16") and hasattr(core, "encode"):
        return core.encode(images, feature_size=args["models"]["feature_size"], autocast=torch.bfloat16)
    if not hasattr(core, "backbone"):
        feats = core(images)
    else:
        try:
            from utils import nested_tensor_from_tensor_list          # the host repository's own helper (not on the hot path)
        except Exception:
            from .detr import nested_tensor_from_tensor_list          # stand-alone: the local restatement (detr.py)
        maps, pos = core.backbone(nested_tensor_from_tensor_list(list(images)))
        src, mask = maps[-1].decompose()
        tokens = core.input_proj(src).flatten(2).permute(2, 0, 1)
        feats = core.transformer.encoder(tokens, src_key_padding_mask=mask.flatten(1), pos=pos[-1].flatten(2).permute(2, 0, 1))
        feats = feats.permute(1, 2, 0)
    return feats.reshape(-1, args["models"]["num_img_feature"], args["models"]["feature_size"], args["models"]["feature_size"])


def _super_targets(args, tgt):
    ng, npos = args["models"]["num_geometric"], args["models"]["num_possessive"]
    return (tgt >= ng).long() + (tgt >= ng + npos).long()


def calculate_losses_on_relationships(args, relation, super_relation, connected, curr_relations_target, criterion_relationship,
                                      pseudo_label_mask=None, lambda_pseudo=1):
    """NLL on the super-category + class-weighted NLL inside the target's super-category (hierarchical), or one class-weighted
    cross entropy (flat) over the connected rows (``train_utils.py:116-157``, ``utils.py:28-35``)."""
    if connected.numel() == 0:
        return 0.0
    tgt = curr_relations_target[connected]
    if not args["models"]["hierarchical_pred"]:
        return criterion_relationship(relation[connected], tgt)
    ng, npos = args["models"]["num_geometric"], args["models"]["num_possessive"]
    crit = criterion_relationship
    sup_t = _super_targets(args, tgt)
    loss = crit[3](super_relation[connected], sup_t)
    rows = relation[connected]
    for k, (lo, hi) in enumerate(((0, ng), (ng, ng + npos), (ng + npos, relation.shape[1]))):
        sel = torch.nonzero(sup_t == k).flatten()
        if sel.numel() > 0:
            loss = loss + crit[k](rows[sel, lo:hi], tgt[sel] - lo)
    return loss


def _call_classifier(relation_classifier, args, h_sub, h_obj, cat_sub, cat_obj, spcat_sub, spcat_obj, rank, h_sub_aug=None, h_obj_aug=None):
    if args["models"]["hierarchical_pred"]:
        r1, r2, r3, super_relation, connectivity, hidden, hidden_aug = relation_classifier(h_sub, h_obj, cat_sub, cat_obj, spcat_sub,
                                                                                             spcat_obj, rank, h_sub_aug, h_obj_aug)
        return (r1, r2, r3), torch.cat((r1, r2, r3), dim=1), super_relation, connectivity, hidden, hidden_aug
    relation, connectivity, hidden, hidden_aug = relation_classifier(h_sub, h_obj, cat_sub, cat_obj, spcat_sub, spcat_obj, rank, h_sub_aug,
                                                                     h_obj_aug)
    return None, relation, None, connectivity, hidden, hidden_aug


def _connectivity_counts(connectivity, rel_row, dir_row, first_direction):
    flag = 1 if first_direction else 0
    not_connected = torch.where(dir_row != flag)[0]
    connected = torch.where(dir_row == flag)[0]
    conn = connectivity[:, 0]
    connected_pred = torch.nonzero(torch.sigmoid(conn) >= 0.5).flatten()
    precision = torch.sum(rel_row[connected_pred] != -1)
    recall = torch.sum(torch.round(torch.sigmoid(conn[connected]))) if len(connected) > 0 else 0.0
    return not_connected, connected, connected_pred, precision, recall


def _feed_evaluators(args, Recall, Recall_top3, keep_in_batch, relation, directed, super_relation, connectivity, cat_sub, cat_obj, bbox_sub,
                     bbox_obj, iou_mask):
    dev = relation.device
    t = lambda x: x.to(dev) if torch.is_tensor(x) else x
    feed = (t(keep_in_batch), relation.detach(), directed, None if super_relation is None else super_relation.detach(),
            torch.log(torch.sigmoid(connectivity[:, 0].detach())), t(cat_sub), t(cat_obj), t(cat_sub), t(cat_obj), t(bbox_sub), t(bbox_obj),
            t(bbox_sub), t(bbox_obj), t(iou_mask))
    Recall.accumulate(*feed)
    if args["dataset"]["dataset"] == "vg" and args["models"]["hierarchical_pred"] and Recall_top3 is not None:
        Recall_top3.accumulate(*feed)


def train_one_direction(relation_classifier, args, h_sub, h_obj, cat_sub, cat_obj, spcat_sub, spcat_obj, bbox_sub, bbox_obj, h_sub_aug, h_obj_aug,
                        iou_mask, rank, graph_iter, edge_iter, keep_in_batch, Recall, Recall_top3, criterion_relationship, criterion_connectivity,
                        relations_target, direction_target, batch_count, hidden_cat_accumulated, hidden_cat_labels_accumulated,
                        commonsense_aligned_triplets, commonsense_violated_triplets, len_train_loader, first_direction=True):
    """One direction-step of ``training()``: classifier call (an autograd node), losses, connectivity counters, contrastive
    bookkeeping, evaluator feed.  Returns the reference's 10-tuple (``train_utils.py:112-113``)."""
    segs, relation, super_relation, connectivity, hidden, hidden_aug = _call_classifier(
        relation_classifier, args, h_sub, h_obj, cat_sub, cat_obj, spcat_sub, spcat_obj, rank, h_sub_aug, h_obj_aug)
    dev = relation.device
    rel_row = relations_target[graph_iter - 1][edge_iter].to(dev)
    dir_row = direction_target[graph_iter - 1][edge_iter].to(dev)
    cat_sub_d, cat_obj_d = cat_sub.to(dev), cat_obj.to(dev)

    loss_commonsense = 0.0
    if args["training"]["run_mode"] == "train_cs":
        ng, npos = args["models"]["num_geometric"], args["models"]["num_possessive"]
        if segs is not None:
            probs = torch.hstack([torch.max(F.softmax(s, dim=1), dim=1)[0] for s in segs])
            pred = torch.hstack([torch.argmax(s, dim=1) + off for s, off in zip(segs, (0, ng, ng + npos))])
            rep = 3
        else:
            probs, pred, rep = torch.max(F.softmax(relation, dim=1), dim=1)[0], torch.argmax(relation, dim=1), 1
        keys = list(zip(cat_sub_d.repeat(rep).tolist(), pred.tolist(), cat_obj_d.repeat(rep).tolist()))
        weak = torch.tensor([k not in commonsense_aligned_triplets for k in keys], dtype=torch.bool, device=dev)
        strong = torch.tensor([k in commonsense_violated_triplets for k in keys], dtype=torch.bool, device=dev)
        if bool(weak.any()):
            loss_commonsense = loss_commonsense + args["training"]["lambda_cs_weak"] * probs[weak].mean()
        if bool(strong.any()):
            loss_commonsense = loss_commonsense + args["training"]["lambda_cs_strong"] * probs[strong].mean()

    not_connected, connected, connected_pred, connectivity_precision, connectivity_recall = _connectivity_counts(
        connectivity, rel_row, dir_row, first_direction)
    loss_connectivity = 0.0
    if len(not_connected) > 0:                     # the mean over an empty set is NaN -> 0.0 in the reference
        loss_connectivity = args["training"]["lambda_not_connected"] * criterion_connectivity(
            connectivity[not_connected, 0], torch.zeros(len(not_connected), device=dev))
    loss_relationship = 0.0
    if len(connected) > 0:
        # overwrites (does not add to) the not-connected term, as the reference does
        loss_connectivity = criterion_connectivity(connectivity[connected, 0], torch.ones(len(connected), device=dev))
        loss_relationship = calculate_losses_on_relationships(args, relation, super_relation, connected, rel_row, criterion_relationship)
        if hidden_aug is not None:
            pairs = torch.stack((hidden[connected], hidden_aug[connected]), dim=1)          # [c, 2, 512]
            for index, batch_index in enumerate(keep_in_batch[connected.to(keep_in_batch.device)]):
                hidden_cat_accumulated[int(batch_index)].append(pairs[index])
                hidden_cat_labels_accumulated[int(batch_index)].append(rel_row[connected][index])

    directed = rel_row.clone()
    directed[not_connected] = -1
    if (batch_count % args["training"]["eval_freq"] == 0) or (batch_count + 1 == len_train_loader):
        _feed_evaluators(args, Recall, Recall_top3, keep_in_batch, relation, directed, super_relation, connectivity, cat_sub, cat_obj,
                         bbox_sub, bbox_obj, iou_mask)
    return (loss_relationship, loss_connectivity, loss_commonsense, len(not_connected), len(connected), len(connected_pred),
            connectivity_precision, connectivity_recall, hidden_cat_accumulated, hidden_cat_labels_accumulated)


def evaluate_one_direction(relation_classifier, args, h_sub, h_obj, cat_sub, cat_obj, spcat_sub, spcat_obj, bbox_sub, bbox_obj,
                           iou_mask, rank, graph_iter, edge_iter, keep_in_batch, Recall, Recall_top3, relations_target,
                           direction_target, batch_count, len_test_loader, first_direction=True):
    _, relation, super_relation, connectivity, _, _ = _call_classifier(relation_classifier, args, h_sub, h_obj, cat_sub, cat_obj, spcat_sub,
                                                                       spcat_obj, rank)
    dev = relation.device
    dir_row = direction_target[graph_iter - 1][edge_iter].to(dev)
    rel_row = relations_target[graph_iter - 1][edge_iter].to(dev)
    not_connected, connected, connected_pred, connectivity_precision, connectivity_recall = _connectivity_counts(
        connectivity, rel_row, dir_row, first_direction)
    directed = rel_row.clone()
    directed[not_connected] = -1
    if (batch_count % args["training"]["eval_freq_test"] == 0) or (batch_count + 1 == len_test_loader):
        _feed_evaluators(args, Recall, Recall_top3, keep_in_batch, relation, directed, super_relation, connectivity, cat_sub, cat_obj,
                         bbox_sub, bbox_obj, iou_mask)
    return len(not_connected), len(connected), len(connected_pred), connectivity_precision, connectivity_recall
