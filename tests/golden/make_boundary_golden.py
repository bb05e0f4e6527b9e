#!/usr/bin/env python3
"""Boundary goldens from the REAL reference (build container only; needs /root/reference): the surfaces of SURVEY 8(b) that
round 1 left unpinned - ``BayesianHead.forward`` (model.py:9-34), ``forward`` with the augmented view (model.py:170-186),
``Evaluator.compute_precision`` (evaluator.py:522-566) and the reference's own per-step TRAINING (``train_one_direction``,
train_utils.py:21-113, + ``losses.backward()``) on the vg_full case.  Only data is stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_boundary_golden.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G                      # noqa: E402  (stubs + reference import helpers of the main generator)

import numpy as np                           # noqa: E402
import torch                                 # noqa: E402

sys.path.insert(0, G.REPO)
from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict   # noqa: E402
from tests.boundary_cases import aug_features, head_inputs, precision_feed                        # noqa: E402


def main():
    ref_model, ref_train, ref_eval = G.import_reference()
    out = {}

    # ---- BayesianHead
    h, sd = head_inputs()
    head = ref_model.BayesianHead(input_dim=512, num_geometric=15, num_possessive=11, num_semantic=24, T1=1, T2=2, T3=0.5)
    head.load_state_dict(sd)
    with torch.no_grad():
        r1, r2, r3, sup = head(h)
    out.update(head_rel1=r1.numpy(), head_rel2=r2.numpy(), head_rel3=r3.numpy(), head_super=sup.numpy())

    # ---- forward() with the augmented view, first two direction-steps of vg_full
    kw, nobj, seed, gain, cfrac, edge = G.CASES["vg_full"]
    cfg = HeadConfig(**kw)
    args = G.ref_args(cfg)
    sdm = make_state_dict(cfg, seed=seed, head_gain=gain)
    batch = make_scene_batch(cfg, nobj, seed=seed, connect_frac=cfrac, edge_boxes=edge)
    model = G.build_ref_model(ref_model, cfg, args, sdm)
    masks = G.ref_masks(batch.bbox, cfg.feature_size)
    feat_aug = aug_features(batch, seed)
    keep = torch.arange(len(nobj))
    gm = torch.stack([masks[i][1].unsqueeze(0) for i in keep]); em = torch.stack([masks[i][0].unsqueeze(0) for i in keep])
    hs = torch.cat((batch.image_feature * gm, batch.image_depth * gm), dim=1)
    ho = torch.cat((batch.image_feature * em, batch.image_depth * em), dim=1)
    hsa = torch.cat((feat_aug * gm, batch.image_depth * gm), dim=1)
    hoa = torch.cat((feat_aug * em, batch.image_depth * em), dim=1)
    cs = torch.tensor([int(batch.categories[i][1]) for i in keep]); co = torch.tensor([int(batch.categories[i][0]) for i in keep])
    ss = [batch.super_categories[i][1] for i in keep]; so = [batch.super_categories[i][0] for i in keep]
    with torch.no_grad():
        o = model(hs, ho, cs, co, ss, so, "cpu", hsa, hoa)
    out.update(aug_rel1=o[0].numpy(), aug_super=o[3].numpy(), aug_conn=o[4].numpy(), aug_pred=o[5].numpy(), aug_pred_aug=o[6].numpy())

    # ---- compute_precision on a synthetic OpenImages feed
    ocfg = HeadConfig(**dict(G.CASES["oiv6_full"][0], hierarchical=False))   # flat: see tests/boundary_cases.py:precision_feed
    oargs = G.ref_args(ocfg)
    ev = ref_eval.Evaluator(args=oargs, num_classes=ocfg.num_relations, iou_thresh=0.5, top_k=[20, 50, 100])
    for a in precision_feed():
        ev.accumulate(*a)
    rec = ev.compute(per_class=True)
    wmap, wmap_u = ev.compute_precision()
    out.update(prec_recall=np.array([float(x) for x in rec[0]]), prec_wmap=np.array([float(wmap), float(wmap_u)]),
               prec_ap=ev.result_per_class_ap.numpy(), prec_ap_union=ev.result_per_class_ap_union.numpy(),
               prec_count=ev.num_conn_target_per_class_ap.numpy())

    # ---- the reference's per-step training on vg_full: train_one_direction + running sums + losses.backward()
    model.eval()                                         # eval-mode numerics (no dropout), autograd on
    for p in model.parameters():
        p.grad = None
    args_t = G.ref_args(cfg)
    args_t["training"]["run_mode"] = "train"
    relations_target, direction_target = G.targets(batch, masks)
    cw = 1 - ref_train.get_num_each_class_reordered(args_t) / torch.sum(ref_train.get_num_each_class_reordered(args_t))
    ng, npos = cfg.num_geometric, cfg.num_possessive
    crit = [torch.nn.NLLLoss(weight=cw[:ng]), torch.nn.NLLLoss(weight=cw[ng:ng + npos]), torch.nn.NLLLoss(weight=cw[ng + npos:]), torch.nn.NLLLoss()]
    crit_c = torch.nn.BCEWithLogitsLoss()
    Recall = ref_eval.Evaluator(args=args_t, num_classes=cfg.num_relations, iou_thresh=0.5, top_k=[20, 50, 100])
    Top3 = ref_eval.Evaluator_Top3(args=args_t, num_classes=cfg.num_relations, iou_thresh=0.5, top_k=[20, 50, 100])
    hid = [[] for _ in nobj]; lab = [[] for _ in nobj]
    losses = lr_run = lc_run = 0.0
    stats = np.zeros(5)
    n_it = torch.as_tensor([len(m) for m in masks])
    for g in range(int(n_it.max())):
        keep = torch.nonzero(n_it > g).view(-1)
        gm = torch.stack([masks[i][g].unsqueeze(0) for i in keep])
        h_g = torch.cat((batch.image_feature[keep] * gm, batch.image_depth[keep] * gm), dim=1)
        h_ga = torch.cat((feat_aug[keep] * gm, batch.image_depth[keep] * gm), dim=1)
        c_g = torch.tensor([int(batch.categories[i][g]) for i in keep]); s_g = [batch.super_categories[i][g] for i in keep]
        b_g = torch.stack([batch.bbox[i][g] for i in keep])
        for e in range(g):
            em = torch.stack([masks[i][e].unsqueeze(0) for i in keep])
            h_e = torch.cat((batch.image_feature[keep] * em, batch.image_depth[keep] * em), dim=1)
            h_ea = torch.cat((feat_aug[keep] * em, batch.image_depth[keep] * em), dim=1)
            c_e = torch.tensor([int(batch.categories[i][e]) for i in keep]); s_e = [batch.super_categories[i][e] for i in keep]
            b_e = torch.stack([batch.bbox[i][e] for i in keep])
            iou_mask = torch.ones(len(keep), dtype=torch.bool)
            for first in (True, False):
                a = (h_g, h_e, c_g, c_e, s_g, s_e, b_g, b_e, h_ga, h_ea) if first else (h_e, h_g, c_e, c_g, s_e, s_g, b_e, b_g, h_ea, h_ga)
                r = ref_train.train_one_direction(model, args_t, *a, iou_mask, "cpu", g, e, keep, Recall, Top3, crit, crit_c, relations_target,
                                                  direction_target, 0, hid, lab, None, None, 1, first_direction=first)
                lr_run = lr_run + r[0]; lc_run = lc_run + r[1]
                stats += np.array([float(r[3]), float(r[4]), float(r[5]), float(r[6]), float(r[7])])
                hid, lab = r[8], r[9]
                losses = losses + lr_run + args_t["training"]["lambda_connectivity"] * lc_run
    losses.backward()
    out["step_loss"] = np.array([float(losses)])
    out["step_stats"] = stats                            # not connected, connected, connected_pred, precision, recall numerators
    for n, p in model.named_parameters():
        flat = p.grad.flatten()
        stride = max(1, flat.numel() // 509)
        out["stepgrad_l2__" + n.replace(".", "__")] = np.array([float(flat.double().norm())])
        out["stepgrad_sample__" + n.replace(".", "__")] = flat[::stride][:509].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "boundary.npz"), **out)
    print("wrote boundary.npz:", {k: v.shape for k, v in out.items() if not k.startswith("stepgrad")})
    print("step loss", float(losses), "stats", stats, "wmap", float(wmap), float(wmap_u), "recall", out["prec_recall"])


if __name__ == "__main__":
    main()
