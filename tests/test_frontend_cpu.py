"""CPU tests of the object front-end oracle (oracle/frontend_oracle.py) against the committed golden vectors
(tests/golden/frontend_vg.npz: the reference's own match_object_categories / iou outputs + the line-by-line restatement of
evaluate.py:311-366) and against hand-computed NMS cases (torchvision is absent: that step is unpinned by the reference)."""
import os

import numpy as np
import torch

from oracle import frontend_oracle as fo
from tests.frontend_cases import make_detr_outputs, make_target_boxes

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "frontend_vg.npz"))
ALP = np.load(os.path.join(HERE, "golden", "ref_fixtures", "object_class_alp2fre.npy"))


def unragged(flat, ptr):
    return [torch.from_numpy(flat[ptr[i]:ptr[i + 1]]) for i in range(len(ptr) - 1)]


def test_class_table_is_a_permutation():
    assert ALP.shape == (151,) and sorted(ALP.tolist()) == list(range(151)) and ALP[150] == 150


def test_candidates_and_nms_match_golden():
    for seed in (1, 2):
        k = "s%d_" % seed
        logits, boxes = make_detr_outputs(seed)
        cats, confs, bxs, kept = fo.detr_candidates(logits, boxes, ALP.tolist(), 150, 2, 32)
        assert kept == GOLD[k + "kept"].tolist() and 3 not in kept
        for name, got in (("pre_cat", cats), ("pre_conf", confs), ("pre_box", bxs)):
            want = unragged(GOLD[k + name], GOLD[k + name + "_ptr"])
            assert len(want) == len(got)
            for w, g in zip(want, got):
                assert torch.equal(w.reshape(g.shape).to(g.dtype), g), name
        c2, f2, b2, _ = fo.frontend_sgdet(logits, boxes, ALP.tolist())
        n_pre = sum(len(c) for c in cats); n_post = sum(len(c) for c in c2)
        assert n_post < n_pre, "the synthetic case must give NMS something to suppress"
        for name, got in (("cat", c2), ("conf", f2), ("box", b2)):
            want = unragged(GOLD[k + name], GOLD[k + name + "_ptr"])
            for w, g in zip(want, got):
                assert torch.equal(w.reshape(g.shape).to(g.dtype), g), name
        for c, f in zip(c2, f2):                       # classes ascending, scores descending inside a class
            assert torch.all(c[1:] >= c[:-1])
            same = c[1:] == c[:-1]
            assert torch.all(f[1:][same] <= f[:-1][same])


def test_matching_matches_the_reference_function():
    for seed in (1, 2):
        k = "s%d_" % seed
        cats = unragged(GOLD[k + "cat"][:, 0], GOLD[k + "cat_ptr"])
        confs = unragged(GOLD[k + "conf"][:, 0], GOLD[k + "conf_ptr"])
        bxs = unragged(GOLD[k + "box"], GOLD[k + "box_ptr"])
        tgt = make_target_boxes(seed, bxs)
        want_t = unragged(GOLD[k + "tgt"], GOLD[k + "tgt_ptr"])
        for a, b in zip(tgt, want_t):
            assert torch.equal(a, b)
        # same tie rule as the reference (torch.topk): exact
        m, mc, tm = fo.match_object_categories(cats, confs, bxs, [t.clone() for t in tgt], stable_ties=False)
        wm = unragged(GOLD[k + "m_cat"][:, 0], GOLD[k + "m_ptr"])
        wc = unragged(GOLD[k + "m_conf"][:, 0], GOLD[k + "m_ptr"])
        wt = unragged(GOLD[k + "m_tgt"], GOLD[k + "m_tgt_ptr"])
        for i in range(len(wm)):
            assert torch.equal(torch.stack(m[i]), wm[i]) and torch.equal(torch.stack(mc[i]), wc[i]) and torch.equal(tm[i], wt[i])
        # stable tie rule (the product path): same multiset of (category, confidence) per ground-truth box, same repeats
        ms, mcs, tms = fo.match_object_categories(cats, confs, bxs, [t.clone() for t in tgt], stable_ties=True)
        n_tie = 0
        for i in range(len(wm)):
            assert torch.equal(tms[i], wt[i]) and len(ms[i]) == len(wm[i])
            got = sorted(zip(torch.stack(ms[i]).tolist(), torch.stack(mcs[i]).tolist()))
            want = sorted(zip(wm[i].tolist(), wc[i].tolist()))
            assert got == want
            n_tie += len(wm[i]) - len(tgt[i])
        assert n_tie > 0, "the case must contain repeated-box ties"
        spots = [(0, 0, 0), (0, 1, 2), (1, 0, 1)]
        for (i, a, b), v in zip(spots, GOLD[k + "iou_spot"]):
            assert fo.iou(tgt[i][a], bxs[i][b]) == v


def test_nms_hand_cases():
    t = torch.tensor
    # B overlaps A by 81/119 = 0.68 > 0.5 -> suppressed; C is disjoint
    boxes = t([[0., 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30]])
    assert fo.nms(boxes, t([0.9, 0.8, 0.7]), 0.5).tolist() == [0, 2]
    # kept indices come back by decreasing score
    assert fo.nms(boxes, t([0.1, 0.8, 0.7]), 0.5).tolist() == [1, 2]
    # IoU exactly at the threshold is NOT suppressed (strict >): inter 2, union 4
    assert fo.nms(t([[0., 0, 2, 2], [0, 0, 2, 1]]), t([0.9, 0.8]), 0.5).tolist() == [0, 1]
    # zero-area boxes give 0/0 = NaN, which compares false: kept
    assert fo.nms(t([[1., 1, 1, 1], [1, 1, 1, 1]]), t([0.9, 0.8]), 0.5).tolist() == [0, 1]
    # equal scores: stable order
    assert fo.nms(t([[0., 0, 1, 1], [5, 5, 6, 6], [9, 9, 10, 10]]), t([0.5, 0.5, 0.5]), 0.5).tolist() == [0, 1, 2]
    # chain: A suppresses B, B would have suppressed C but is gone -> C kept
    boxes = t([[0., 0, 10, 10], [0, 4, 10, 14], [0, 8, 10, 18]])
    assert fo.nms(boxes, t([0.9, 0.8, 0.7]), 0.4).tolist() == [0, 2]
    # per-class: the same two boxes in different classes both survive
    c, f, b, keep = fo.per_class_nms(t([5, 3, 5]), t([0.9, 0.8, 0.7]), t([[0., 10, 0, 10], [0, 10, 0, 10], [1, 11, 1, 11]]), 0.5)
    assert c.tolist() == [3, 5] and keep.tolist() == [1, 0]


def test_nms_oracle_against_independent_restatement_and_hand_vectors():
    """the oracle's NMS pinned by a second restatement of torchvision 0.15.2's CPU kernel that shares no code
    with it (tests/nms_cases.py) and by hand-derived vectors; then both on a randomised stress with integer-grid boxes (exact
    threshold hits and score ties are frequent there)."""
    from tests.nms_cases import HAND_CASES, nms_matrix
    t = lambda x: torch.tensor(x, dtype=torch.float32)
    for boxes, scores, thr, expect, why in HAND_CASES:
        assert nms_matrix(boxes, scores, thr) == expect, ("independent", why)
        assert fo.nms(t(boxes).reshape(-1, 4), t(scores), thr).tolist() == expect, ("oracle", why)
    rng = np.random.default_rng(3)
    for case in range(400):
        n = int(rng.integers(1, 40))
        xy = rng.integers(0, 12, (n, 2)).astype(np.float32)
        wh = rng.integers(0, 7, (n, 2)).astype(np.float32)                 # zero sizes included
        boxes = np.concatenate([xy, xy + wh], axis=1)
        scores = (rng.integers(0, 8, n) / 8.0).astype(np.float32)          # many exact ties
        thr = [0.25, 1.0 / 3.0, 0.5, 0.0, 0.75][case % 5]
        assert fo.nms(torch.from_numpy(boxes), torch.from_numpy(scores), thr).tolist() == nms_matrix(boxes, scores, thr), case
