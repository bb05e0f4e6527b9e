#!/usr/bin/env python3
"""Golden vectors for the SGDET / SGCLS object front-end.  Runs only in the build container.

What is REAL reference code here: ``utils.match_object_categories`` and ``utils.iou`` (utils.py:58-74,377-425) and the
class-index table ``dataset_utils.object_class_alp2fre()`` (dataset_utils.py:606-614), imported from /root/reference with
empty stub modules for packages it imports but does not use on this path.  The inline block of evaluate.py:311-366 cannot
be imported (tensorboard, process group, a live DETR): it is restated below line by line around the same torch ops, with
``oracle.frontend_oracle.nms`` standing in for ``torchvision.ops.nms`` (torchvision 0.15.2 is not installed) - that one step
is therefore not pinned by the reference.

Only data is committed: tests/golden/frontend_vg.npz and tests/golden/ref_fixtures/{object_class_alp2fre,relation_class_freq2scat}.npy.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_frontend_golden.py
"""
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

import numpy as np
import torch

from tests.frontend_cases import make_detr_outputs, make_target_boxes      # noqa: E402  (seeded synthetic inputs)
sys.path.insert(0, HERE)
from oracle import frontend_oracle as fo                                   # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    _stub("torchmetrics")
    tv = _stub("torchvision", _is_tracing=lambda: False)
    tv.transforms = _stub("torchvision.transforms")
    tv.ops = _stub("torchvision.ops", nms=fo.nms)
    _stub("openai")
    _stub("cv2")
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    import utils as ref_utils                  # noqa
    import dataset_utils as ref_dataset_utils  # noqa
    os.chdir(cwd)
    return ref_utils, ref_dataset_utils


def reference_inline_sgdet(out_dict, args, object_class_alp2fre_dict, torchvision):
    """The inline front-end block of ``eval_sgd`` (evaluate.py:311-366), EXECUTED from /root/reference/evaluate.py itself (see
    ref_extract.py) - no restatement.  Run in two parts to snapshot the lists before the per-class NMS loop."""
    import torch.nn.functional as F
    from ref_extract import reference_block, run_block
    path = os.path.join(REF, "evaluate.py")
    part1, l1 = reference_block(path, "eval_sgd", "logits_pred = torch.argmax(F.softmax(out_dict['pred_logits']", "masks_pred[i] = masks_pred[i][cat_mask[i]]")
    part2, l2 = reference_block(path, "eval_sgd", "# non-maximum suppression", "masks_pred[i] = masks_pred[i][nms_keep_idx]")
    assert l1[1] < l2[0]
    ns = dict(torch=torch, F=F, torchvision=torchvision, out_dict=out_dict, args=args, object_class_alp2fre_dict=object_class_alp2fre_dict,
              rank="cpu")
    run_block(part1, ns, "evaluate.py:%d-%d" % l1)
    pre_nms = ([c.clone() for c in ns["categories_pred"]], [c.clone() for c in ns["cat_pred_confidence"]], [b.clone() for b in ns["bbox_pred"]])
    run_block(part2, ns, "evaluate.py:%d-%d" % l2)
    kept = [i for i in range(ns["has_object_pred"].shape[0]) if torch.sum(ns["has_object_pred"][i]) > 0]
    return ns["categories_pred"], ns["cat_pred_confidence"], ns["bbox_pred"], kept, pre_nms


def ragged(lst, dtype):
    ptr = np.cumsum([0] + [int(len(x)) for x in lst]).astype(np.int32)
    if len(lst) == 0:
        return np.zeros((0,), dtype), ptr
    flat = torch.cat([torch.as_tensor(x).reshape(len(x), -1) for x in lst]).numpy().astype(dtype)
    return flat, ptr


def main():
    ref_utils, ref_du = import_reference()
    import torchvision
    alp = ref_du.object_class_alp2fre()
    table = np.array([alp[i] for i in range(len(alp))], dtype=np.int32)
    os.makedirs(os.path.join(HERE, "ref_fixtures"), exist_ok=True)
    np.save(os.path.join(HERE, "ref_fixtures", "object_class_alp2fre.npy"), table)
    # predicate re-indexing table of the data loader (dataset_utils.py:647-650; index 50 = the "no relation" slot -> -1)
    np.save(os.path.join(HERE, "ref_fixtures", "relation_class_freq2scat.npy"),
            ref_du.relation_class_freq2scat().numpy().astype(np.int64))
    args = {'models': {'num_classes': 150, 'topk_cat': 2, 'feature_size': 32, 'nms': 0.5}}
    out = {}
    for seed in (1, 2):
        logits, boxes = make_detr_outputs(seed)
        cats, confs, bxs, kept, pre = reference_inline_sgdet({'pred_logits': logits.clone(), 'pred_boxes': boxes.clone()}, args, alp, torchvision)
        tgt = make_target_boxes(seed, bxs)
        m, mc, tm = ref_utils.match_object_categories(cats, confs, bxs, [t.clone() for t in tgt])
        # per ground-truth box: the IoU row maxima, so that the test can tell genuine ties from the repeated-box case
        k = "s%d_" % seed
        out[k + "kept"] = np.array(kept, dtype=np.int32)
        for name, lst, dt in (("cat", cats, np.int64), ("conf", confs, np.float32), ("box", bxs, np.float32),
                              ("pre_cat", pre[0], np.int64), ("pre_conf", pre[1], np.float32), ("pre_box", pre[2], np.float32),
                              ("tgt", tgt, np.float32)):
            out[k + name], out[k + name + "_ptr"] = ragged(lst, dt)
        out[k + "m_cat"], out[k + "m_ptr"] = ragged([torch.stack(x) if len(x) else torch.zeros(0) for x in m], np.int64)
        out[k + "m_conf"], _ = ragged([torch.stack(x) if len(x) else torch.zeros(0) for x in mc], np.float32)
        out[k + "m_tgt"], out[k + "m_tgt_ptr"] = ragged(tm, np.float32)
        # spot values of the reference's own iou()
        pairs = [(0, 0, 0), (0, 1, 2), (1, 0, 1)]
        out[k + "iou_spot"] = np.array([ref_utils.iou(tgt[i][a], bxs[i][b]) for i, a, b in pairs], dtype=np.float64)
        print("seed", seed, "kept images", kept, "objects after NMS", [len(c) for c in cats], "matched", [len(x) for x in m])
    np.savez_compressed(os.path.join(HERE, "frontend_vg.npz"), **out)
    print("wrote", os.path.join(HERE, "frontend_vg.npz"))


if __name__ == "__main__":
    main()
