#!/usr/bin/env python3
"""Golden vectors for the end-to-end SGDET evaluation (DETR outputs -> object front-end -> pair loop over the PREDICTED objects ->
Recall with predcls=False).  Runs only in the build container.

REAL reference code used: ``model.BayesianRelationClassifier``, ``evaluator.Evaluator`` (accumulate with predcls=False and the
category confidences, accumulate_target, compute(per_class=True, predcls=False)), ``utils.match_target_sgd``,
``utils.compare_object_cat`` (through the evaluator).  evaluate.py cannot be imported (tensorboard, process group, DETR download): its inline
front-end block (see make_frontend_golden.py; torchvision.ops.nms is substituted by oracle.frontend_oracle.nms) and its SGDET pair
loop (evaluate.py:375-440) are EXECUTED from the reference's file at generation time (ref_extract.py), not restated here.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_sgdet_golden.py
"""
import importlib.util
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

import numpy as np
import torch


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


mg = _load("make_golden")
mf = _load("make_frontend_golden")
from scene_graph_commonsense_amd.synthetic import hash_uniform        # noqa: E402
from tests import sgdet_case                                           # noqa: E402
from oracle import frontend_oracle as fo                              # noqa: E402


def main():
    cwd = os.getcwd()
    ref_model, ref_train, ref_eval = mg.import_reference()
    import utils as ref_utils
    import torchvision
    torchvision.ops.nms = fo.nms
    os.chdir(cwd)
    cfg, sd, batch, logits, boxes = sgdet_case.make_case()
    args = mg.ref_args(cfg)
    os.chdir(mg.REF)
    model = mg.build_ref_model(ref_model, cfg, args, sd)
    Fs = cfg.feature_size
    out = {}
    # ---- ground-truth relations := the reference model's own predictions on ~60 % of the ground-truth pairs
    gmasks = mg.ref_masks(batch.bbox, Fs)
    with torch.no_grad():
        for b, n in enumerate(sgdet_case.NOBJ):
            for g in range(1, n):
                for e in range(g):
                    u, d = (float(x) for x in hash_uniform(991 + b * 100 + g * 10 + e, 2))
                    if u < 0.6:
                        first = d < 0.5
                        mgm, mem = gmasks[b][g][None, None], gmasks[b][e][None, None]
                        hg = torch.cat((batch.image_feature[b:b + 1] * mgm, batch.image_depth[b:b + 1] * mgm), dim=1)
                        he = torch.cat((batch.image_feature[b:b + 1] * mem, batch.image_depth[b:b + 1] * mem), dim=1)
                        hs, ho = (hg, he) if first else (he, hg)
                        si, oi = (g, e) if first else (e, g)
                        r = model(hs, ho, batch.categories[b][si].view(1), batch.categories[b][oi].view(1),
                                  [batch.super_categories[b][si]], [batch.super_categories[b][oi]], "cpu")
                        batch.relationships[b][g - 1][e] = int(torch.argmax(torch.cat((r[0], r[1], r[2]), dim=1)[0]))
                        batch.subj_or_obj[b][g - 1][e] = 1.0 if first else 0.0
    for b, n in enumerate(sgdet_case.NOBJ):
        for g in range(1, n):
            out["tgt_rel_%d_%d" % (b, g)] = batch.relationships[b][g - 1].numpy().copy()
            out["tgt_dir_%d_%d" % (b, g)] = batch.subj_or_obj[b][g - 1].numpy().copy()
    # ---- front end (inline block restated) on the synthetic DETR outputs
    alp = {i: int(v) for i, v in enumerate(sgdet_case.alp2fre_table())}
    fargs = {'models': {'num_classes': 150, 'topk_cat': 2, 'feature_size': 32, 'nms': 0.5}}
    categories_pred, cat_pred_confidence, bbox_pred, kept, _ = mf.reference_inline_sgdet(
        {'pred_logits': logits.clone(), 'pred_boxes': boxes.clone()}, fargs, alp, torchvision)
    assert kept == list(range(len(sgdet_case.NOBJ)))
    masks_pred = mg.ref_masks(bbox_pred, Fs)
    super_categories_pred = sgdet_case.super_categories_of(categories_pred, cfg)
    for i in range(len(kept)):
        out["fe_cat_%d" % i] = categories_pred[i].numpy(); out["fe_conf_%d" % i] = cat_pred_confidence[i].numpy()
        out["fe_box_%d" % i] = bbox_pred[i].numpy()
    # ---- targets (REAL utils.match_target_sgd)
    bbox_target = [b.clone() for b in batch.bbox]
    cat_subject_target, cat_object_target, bbox_subject_target, bbox_object_target, relation_target = \
        ref_utils.match_target_sgd("cpu", batch.relationships, batch.subj_or_obj, batch.categories, bbox_target)
    # ---- SGDET pair loop (evaluate.py:375-440) with the REAL classifier and evaluator
    Recall = ref_eval.Evaluator(args=args, num_classes=cfg.num_relations, iou_thresh=0.5, top_k=[20, 50, 100])

    class Counted:
        """The reference classifier, counting its calls."""
        def __init__(self, m): self.m, self.n = m, 0
        def __call__(self, *a, **k):
            self.n += 1
            return self.m(*a, **k)
    counted = Counted(model)
    sys.path.insert(0, HERE)
    from ref_extract import reference_block, run_block
    loop, ln = reference_block(os.path.join(mg.REF, "evaluate.py"), "eval_sgd", "num_graph_iter = torch.as_tensor([len(mask) for mask in masks_pred])",
                               "iou_mask, False, cat_edge_confidence, cat_graph_confidence)")
    import torch.nn.functional as F
    fargs_loop = dict(args)
    fargs_loop["training"] = dict(args["training"], eval_freq_test=1)
    ns = dict(torch=torch, F=F, args=fargs_loop, rank="cpu", relation_classifier=counted, Recall=Recall, batch_count=0, test_loader=[0],
              image_feature=batch.image_feature, image_depth=batch.image_depth, masks_pred=masks_pred, categories_pred=categories_pred,
              bbox_pred=bbox_pred, cat_pred_confidence=cat_pred_confidence, super_categories_pred=super_categories_pred)
    with torch.no_grad():
        run_block(loop, ns, "evaluate.py:%d-%d" % ln)            # the reference's own SGDET pair loop, verbatim from its file
    n_calls = counted.n
    Recall.accumulate_target(relation_target, cat_subject_target, cat_object_target, bbox_subject_target, bbox_object_target)
    out["ev_conf"] = Recall.confidence.numpy().copy(); out["ev_conn"] = Recall.connectivity.numpy().copy()
    out["ev_pred"] = Recall.relation_pred.numpy().copy(); out["ev_which"] = Recall.which_in_batch.numpy().copy()
    out["ev_scat"] = Recall.subject_cat_pred.numpy().copy(); out["ev_ocat"] = Recall.object_cat_pred.numpy().copy()
    recall, recall_per_class, mean_recall, recall_zs, _, mean_recall_zs = Recall.compute(per_class=True, predcls=False)
    out["recall"] = np.array([float(r) for r in recall]); out["mean_recall"] = np.array([float(r) for r in mean_recall])
    out["recall_per_class"] = np.stack([r.numpy() for r in recall_per_class])
    out["recall_zs"] = np.array([float(r) for r in recall_zs])
    out["num_connected_target"] = np.array(float(Recall.num_connected_target))
    out["hits"] = np.array([float(Recall.result_dict[k]) for k in (20, 50, 100)])
    out["n_calls"] = np.array(n_calls)
    for i in range(len(kept)):
        out["mt_rel_%d" % i] = relation_target[i].numpy() if relation_target[i] is not None else np.zeros(0)
    os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, "sgdet_vg.npz"), **out)
    print("classifier calls", n_calls, "objects", [len(c) for c in categories_pred], "targets", float(Recall.num_connected_target),
          "hits", out["hits"], "recall", out["recall"])


if __name__ == "__main__":
    main()
