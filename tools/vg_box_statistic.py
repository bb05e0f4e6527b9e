#!/usr/bin/env python3
"""Fraction of pair-specific (X) conv3 windows on boxes that went through the REFERENCE'S box pipeline (VERDICT r4 missing 4).

No Visual Genome annotation exists offline, so the raw pixel boxes are drawn from ASSUMED marginals (log-normal area fraction with
a heavy tail, log-normal aspect ratio, uniform position, typical VG image sizes: ``synthetic.vg_like_boxes``); everything after that
is the reference's own arithmetic: ``utils.resize_boxes`` (int truncation to the 32-grid, ``/root/reference/utils.py:38-55``), the
(x0,x1,y0,y1) storage and area ordering of ``dataset_utils.py:113-126``, the loader's ``bbox.int()`` and drop rules
(``dataloader.py:119-129``, through ``annotations.prepare_annotation``).  Three area marginals bracket the unknown truth.  Reported
per setting, over the ordered pairs of the surviving images: share of X windows |R_i n R_j| / 64, the part of it that is LINEAR
(combined, not convolved), the conv-list share (what the three window GEMMs run on), per-object windows per pair.  CPU only.

    python tools/vg_box_statistic.py            # the table of DESIGN.md section 7
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scene_graph_commonsense_amd import annotations as AN                      # noqa: E402
from scene_graph_commonsense_amd import pairs as PR                            # noqa: E402
from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, vg_like_boxes, VG_IMAGE_SIZES      # noqa: E402


def window_shares(boxes_per_image):
    """(X share, linear share of all windows, conv share, per-object windows per ordered pair, ordered pairs)."""
    bb = np.concatenate([PR.normalise_boxes(torch.as_tensor(b), 32) for b in boxes_per_image])
    img_ptr = np.concatenate([[0], np.cumsum([len(b) for b in boxes_per_image])])
    P = int(sum(len(b) * (len(b) - 1) for b in boxes_per_image))
    x = PR.count_shared_windows(bb, img_ptr)
    lin = PR.count_linear_windows(bb, img_ptr)
    obj = PR.count_object_windows(bb)
    return x / (64.0 * P), lin / (64.0 * P), (x - lin + obj) / (64.0 * P), obj / float(P), P


def loader_images(n_images, objects, rng, **kw):
    """Images as the reference's loader would deliver them: object count from ``objects`` (callable), annotation dict in the on-disk
    format, ``prepare_annotation`` with the image size (drop rules)."""
    out, dropped = [], 0
    seed = int(rng.integers(1 << 30))
    k = 0
    while len(out) < n_images:
        n = int(objects(rng))
        k += 1
        boxes = vg_like_boxes(n, seed + k, **kw)
        hw = VG_IMAGE_SIZES[int(np.random.default_rng(seed + k).integers(len(VG_IMAGE_SIZES)))]      # the size vg_like_boxes drew
        annot = dict(categories=torch.zeros(n, dtype=torch.int64), bbox=torch.from_numpy(boxes).float(), image_depth=torch.zeros(1, 32, 32),
                     relationships=[], subj_or_obj=[], super_categories=None)
        a = AN.prepare_annotation(annot, feature_size=32, image_hw=hw, rel_reorder=None)
        if a is None:
            dropped += 1
            continue
        out.append(a["bbox"].numpy())
    return out, dropped / float(k)


def main():
    rng = np.random.default_rng(0)
    cfg = HeadConfig()
    rows = []
    b = make_scene_batch(cfg, [64] * 8, seed=1000)
    rows.append(("bench.py default (SURVEY 8d boxes), 8 x 64",) + window_shares([x.numpy() for x in b.bbox]) + (0.0,))
    # VG150 after the loader's <= 20 filter: a right-skewed count, mean ~ 11 (dataloader.py:118 quotes the tail: 2651 of 60548 above 20)
    vg_count = lambda r: int(np.clip(r.gamma(4.0, 2.9), 2, 20))
    for label, med in (("small objects (median area 3 %)", 0.03), ("median area 6 %", 0.06), ("large objects (median area 12 %)", 0.12)):
        imgs, drop = loader_images(512, vg_count, rng, median_area=med)
        rows.append(("VG-like marginals through the reference's box pipeline, %s, 2-20 objects" % label,) + window_shares(imgs) + (drop,))
        imgs64 = [vg_like_boxes(64, 1000 + 31 * i, median_area=med) for i in range(8)]
        rows.append(("  the same marginals at the benchmark's size, 8 x 64",) + window_shares(imgs64) + (0.0,))
    print("%-100s %8s %8s %8s %9s %10s %8s" % ("boxes", "X", "linear", "conv", "obj/pair", "pairs", "dropped"))
    for label, x, lin, conv, obj, P, drop in rows:
        print("%-100s %8.4f %8.4f %8.4f %9.3f %10d %8.3f" % (label, x, lin, conv, obj, P, drop))


if __name__ == "__main__":
    main()
