"""Per-step drop-in for the reference's ``train_utils.evaluate_one_direction`` (``train_utils.py:160-196``).

Lets the reference's own ``graph_iter`` / ``edge_iter`` loops (``train_test.py:386-437``, ``evaluate.py:132-183``) run
unchanged on top of the HIP-backed classifier and evaluators: same positional arguments, same 5-tuple.  The fused
equivalents (one call per minibatch) are ``pair_loop.evaluate_minibatch`` / ``train_minibatch``; per-step *training*
(``train_one_direction``) is not offered - the per-step classifier call builds no autograd graph.
"""
from __future__ import annotations

import torch


def evaluate_one_direction(relation_classifier, args, h_sub, h_obj, cat_sub, cat_obj, spcat_sub, spcat_obj, bbox_sub, bbox_obj,
                           iou_mask, rank, graph_iter, edge_iter, keep_in_batch, Recall, Recall_top3, relations_target,
                           direction_target, batch_count, len_test_loader, first_direction=True):
    if args["models"]["hierarchical_pred"]:
        r1, r2, r3, super_relation, connectivity, _, _ = relation_classifier(h_sub, h_obj, cat_sub, cat_obj, spcat_sub, spcat_obj, rank)
        relation = torch.cat((r1, r2, r3), dim=1)
    else:
        relation, connectivity, _, _ = relation_classifier(h_sub, h_obj, cat_sub, cat_obj, spcat_sub, spcat_obj, rank)
        super_relation = None
    dev = relation.device
    dir_row = direction_target[graph_iter - 1][edge_iter].to(dev)
    rel_row = relations_target[graph_iter - 1][edge_iter].to(dev)
    flag = 1 if first_direction else 0
    not_connected = torch.where(dir_row != flag)[0]
    connected = torch.where(dir_row == flag)[0]
    conn = connectivity[:, 0]
    connected_pred = torch.nonzero(torch.sigmoid(conn) >= 0.5).flatten()
    connectivity_precision = torch.sum(rel_row[connected_pred] != -1)
    connectivity_recall = 0.0
    if len(connected) > 0:
        connectivity_recall = torch.sum(torch.round(torch.sigmoid(conn[connected])))
    directed = rel_row.clone()
    directed[not_connected] = -1
    if (batch_count % args["training"]["eval_freq_test"] == 0) or (batch_count + 1 == len_test_loader):
        t = lambda x: x.to(dev) if torch.is_tensor(x) else x
        feed = (t(keep_in_batch), relation, directed, super_relation, torch.log(torch.sigmoid(conn)), t(cat_sub), t(cat_obj),
                t(cat_sub), t(cat_obj), t(bbox_sub), t(bbox_obj), t(bbox_sub), t(bbox_obj), t(iou_mask))
        Recall.accumulate(*feed)
        if args["dataset"]["dataset"] == "vg" and args["models"]["hierarchical_pred"] and Recall_top3 is not None:
            Recall_top3.accumulate(*feed)
    return len(not_connected), len(connected), len(connected_pred), connectivity_precision, connectivity_recall
