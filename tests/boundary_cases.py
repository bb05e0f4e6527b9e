"""Seeded inputs of the boundary goldens (tests/golden/make_boundary_golden.py runs the REAL reference on them)."""
import numpy as np
import torch

from scene_graph_commonsense_amd.synthetic import hash_normal, hash_randint, hash_uniform


def head_inputs(M=37, seed=71):
    """[M,512] non-negative features (the head sits behind a ReLU) and BayesianHead weights with O(1) logits."""
    h = torch.from_numpy(np.maximum(hash_normal(seed, M * 512), 0).reshape(M, 512))
    sd = {}
    for k, (name, rows) in enumerate((("fc3_1", 15), ("fc3_2", 11), ("fc3_3", 24), ("fc5", 3))):
        sd[name + ".weight"] = torch.from_numpy(hash_uniform(seed * 7 + k, rows * 512, -0.25, 0.25).reshape(rows, 512))
        sd[name + ".bias"] = torch.from_numpy(hash_uniform(seed * 7 + k + 100, rows, -0.5, 0.5))
    return h, sd


def aug_features(batch, seed):
    f = batch.image_feature
    noise = torch.from_numpy(hash_normal(seed * 31 + 99, f.numel()).reshape(f.shape))
    return 0.9 * f + 0.3 * noise


def precision_feed(seed=5, n_img=3, steps=14, R=30):
    """A synthetic OpenImages-style evaluator feed for the FLAT head (the reference's ``compute_precision`` indexes its
    targets with the candidate mask, ``evaluator.py:534``, which only has the right length with one candidate per pair):
    ``steps`` accumulate() calls of 1..n_img rows with random logits, connectivity, categories and grid boxes; about half of
    the targets equal the row's prediction, so the weighted mean AP is far from 0 and from 1.  Returns a list of positional-argument tuples for
    ``Evaluator.accumulate`` (CPU tensors)."""
    feed = []
    for t in range(steps):
        s = seed * 1000 + t * 17
        b = 1 + int(hash_randint(s, 1, 0, n_img)[0])
        which = torch.from_numpy(np.sort(hash_randint(s + 1, n_img, 0, 1000).argsort()[:b]).astype(np.int64))
        rel = torch.from_numpy(np.log(np.clip(hash_uniform(s + 2, b * R, 0.0, 1.0), 1e-3, 1)).reshape(b, R).astype(np.float32))
        sup = torch.from_numpy(np.log(np.clip(hash_uniform(s + 3, b * 3, 0.0, 1.0), 1e-3, 1)).reshape(b, 3).astype(np.float32))
        conn = torch.from_numpy(-np.abs(hash_normal(s + 4, b)).astype(np.float32))
        scat = torch.from_numpy(hash_randint(s + 5, b, 0, 6))
        ocat = torch.from_numpy(hash_randint(s + 6, b, 0, 6))
        x0 = hash_randint(s + 7, 2 * b, 0, 20); y0 = hash_randint(s + 8, 2 * b, 0, 20)
        w = hash_randint(s + 9, 2 * b, 4, 12); h = hash_randint(s + 10, 2 * b, 4, 12)
        box = torch.from_numpy(np.stack([x0, x0 + w, y0, y0 + h], axis=1).astype(np.int32))
        sbox, obox = box[:b], box[b:]
        cand = rel.argmax(1)                                       # flat head: one candidate per pair
        pick = hash_randint(s + 11, b, 0, 6)                       # 0..2: target = the prediction, 3: another class, 4..5: not connected
        tgt = torch.full((b,), -1, dtype=torch.int64)
        for r in range(b):
            if pick[r] < 3:
                tgt[r] = cand[r]
            elif pick[r] == 3:
                tgt[r] = (int(cand[r]) + 7) % R
        iou = torch.from_numpy(hash_uniform(s + 12, b, 0.0, 1.0) < 0.85)
        feed.append((which, rel, tgt, sup, conn, scat, ocat, scat, ocat, sbox, obox, sbox, obox, iou))
    return feed
