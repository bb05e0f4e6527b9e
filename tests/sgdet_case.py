"""Seeded end-to-end SGDET case (shared by tests/golden/make_sgdet_golden.py, the CPU oracle test and the GPU parity test):
a synthetic ground-truth minibatch and DETR decoder outputs derived from it - one confident query per ground-truth object
(box converted exactly to cxcywh), a runner-up class per query, a near-duplicate that per-class NMS must remove and two
distractor detections per image."""
import os

import numpy as np
import torch

from scene_graph_commonsense_amd.synthetic import HeadConfig, default_sub2super, make_scene_batch, make_state_dict

HERE = os.path.dirname(os.path.abspath(__file__))
NOBJ = (4, 3, 3)
SEED = 5


def alp2fre_table():
    return np.load(os.path.join(HERE, "golden", "ref_fixtures", "object_class_alp2fre.npy"))


def make_case():
    cfg = HeadConfig()
    sd = make_state_dict(cfg, seed=SEED, head_gain=6.0)
    batch = make_scene_batch(cfg, NOBJ, seed=SEED, connect_frac=0.0, edge_boxes=False)
    alp = alp2fre_table()
    inv = np.argsort(alp)                       # dataset class -> DETR class
    g = torch.Generator().manual_seed(4000 + SEED)
    B, Q, C1 = len(NOBJ), 100, 151
    logits = torch.randn(B, Q, C1, generator=g) * 0.3
    logits[:, :, 150] += 10.0
    boxes = torch.rand(B, Q, 4, generator=g) * 0.2 + 0.4

    def put(b, q, cat, gain, box, runner_up=None):
        logits[b, q, 150] -= 10.0
        logits[b, q, int(inv[cat])] += gain
        if runner_up is not None:
            logits[b, q, int(inv[runner_up])] += gain - 5.0
        x0, x1, y0, y1 = [float(v) for v in box]
        boxes[b, q] = torch.tensor([(x0 + x1) / 64.0, (y0 + y1) / 64.0, (x1 - x0) / 32.0, (y1 - y0) / 32.0])

    for b in range(B):
        for o in range(NOBJ[b]):
            c = int(batch.categories[b][o])
            put(b, 3 * o + 1, c, 14.0, batch.bbox[b][o], runner_up=(c * 7 + 3) % 150)
        c0 = int(batch.categories[b][0])
        bb = batch.bbox[b][0].float() + torch.tensor([0.3, 0.3, 0.2, 0.2])
        put(b, 2, c0, 12.0, bb)                                             # near-duplicate of object 0: suppressed by NMS
        for d in range(2):
            cx = int(torch.randint(0, 150, (1,), generator=g))
            x0 = int(torch.randint(0, 20, (1,), generator=g)); y0 = int(torch.randint(0, 20, (1,), generator=g))
            put(b, 40 + d, cx, 9.0, (x0, x0 + 6 + d, y0, y0 + 5 + d))
    return cfg, sd, batch, logits.contiguous(), boxes.contiguous().clamp(0.0, 1.0)


def apply_stored_targets(batch, gold):
    """Overwrite the ground-truth relations with the ones the golden generator chose (the reference model's own predictions
    on a subset of the ground-truth pairs, so that the recall counters are non-trivial)."""
    for b, n in enumerate(NOBJ):
        for gi in range(1, n):
            batch.relationships[b][gi - 1] = torch.from_numpy(gold["tgt_rel_%d_%d" % (b, gi)].copy())
            batch.subj_or_obj[b][gi - 1] = torch.from_numpy(gold["tgt_dir_%d_%d" % (b, gi)].copy())
    return batch


def super_categories_of(cats_per_image, cfg):
    table = default_sub2super(cfg.num_classes, cfg.num_super_classes)
    return [[torch.as_tensor(table[int(c)]) for c in cats] for cats in cats_per_image]
