"""An INDEPENDENT pin for the NMS step of the SGDET front-end (``evaluate.py:347-366`` calls ``torchvision.ops.nms``; torchvision
0.15.2, ``requirements.txt:160``, is not installed here, so round 1's golden used the builder's own ``oracle/frontend_oracle.nms``).

Two things that share no code with that oracle:

* ``nms_matrix`` - a second restatement of torchvision 0.15.2's published CPU kernel (``torchvision/csrc/ops/cpu/nms_kernel.cpp``:
  stable descending sort of the scores, areas (x2-x1)*(y2-y1) and intersections max(0, .) in the box dtype (f32), a box j is
  suppressed by a kept box i when ``inter / (area_i + area_j - inter) > iou_threshold`` with the f32 quotient compared against
  the DOUBLE threshold), written as an IoU matrix + one pass over the sorted order in numpy;
* ``HAND_CASES`` - vectors derived by hand, with the arithmetic in the comments: the ``>`` (not ``>=``) threshold edge, the
  f32-quotient-vs-double-threshold edge, equal scores (stable order), zero-area and inverted boxes (0/0 = NaN never suppresses),
  and a suppression chain (a suppressed box suppresses nobody).
"""
import numpy as np


def nms_matrix(boxes, scores, iou_threshold):
    """boxes [n,4] (x1,y1,x2,y2) f32, scores [n] f32 -> kept indices, highest score first."""
    b = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    s = np.asarray(scores, dtype=np.float32)
    n = len(s)
    if n == 0:
        return []
    order = np.argsort(-s.astype(np.float64), kind="stable")          # stable: equal scores keep index order
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    w = np.maximum(np.float32(0), np.minimum(b[:, None, 2], b[None, :, 2]) - np.maximum(b[:, None, 0], b[None, :, 0]))
    h = np.maximum(np.float32(0), np.minimum(b[:, None, 3], b[None, :, 3]) - np.maximum(b[:, None, 1], b[None, :, 1]))
    inter = (w * h).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = (inter / (area[:, None] + area[None, :] - inter)).astype(np.float32)
    over = iou.astype(np.float64) > float(iou_threshold)             # NaN compares False
    alive = np.ones(n, dtype=bool)
    keep = []
    for i in order:
        if alive[i]:
            keep.append(int(i))
            alive &= ~over[i]                                          # suppress everything this kept box overlaps too much ...
            alive[i] = False                                           # ... (already-visited boxes are never looked at again)
    return keep


# (boxes x1,y1,x2,y2; scores; threshold; expected kept indices in output order; why)
HAND_CASES = [
    # A=[0,0,2,2] area 4, B=[0,0,2,1] area 2: inter 2, union 4, IoU = 0.5 exactly.  0.5 > 0.5 is false -> B survives.
    ([[0, 0, 2, 2], [0, 0, 2, 1]], [0.9, 0.8], 0.5, [0, 1], "IoU == threshold is kept ('>' not '>=')"),
    # same boxes, threshold just below: suppressed
    ([[0, 0, 2, 2], [0, 0, 2, 1]], [0.9, 0.8], 0.49, [0], "IoU 0.5 > 0.49"),
    # A=[0,0,2,1], B=[1,0,3,1]: inter 1, union 3, IoU = 1/3.  In f32 1/3 = 0.3333333432674408 which IS greater than the double
    # threshold 0.3333333333333333 -> B is suppressed (the quotient is f32, the threshold stays double).
    ([[0, 0, 2, 1], [1, 0, 3, 1]], [0.9, 0.8], 1.0 / 3.0, [0], "f32(1/3) > double(1/3)"),
    # ... and with a threshold above the f32 value 0.3333333432674408 it is kept
    ([[0, 0, 2, 1], [1, 0, 3, 1]], [0.9, 0.8], 0.3333334, [0, 1], "f32(1/3) = 0.33333334327 is not > 0.3333334"),
    # equal scores: stable order = index order; box 1 == box 0 -> IoU 1 -> the LOWER index survives
    ([[0, 0, 4, 4], [0, 0, 4, 4], [10, 10, 12, 12]], [0.7, 0.7, 0.7], 0.5, [0, 2], "ties resolved by index"),
    # zero-area boxes: inter 0, union 0 -> 0/0 = NaN, NaN > thr is false: identical degenerate boxes are ALL kept
    ([[1, 1, 1, 1], [1, 1, 1, 1]], [0.9, 0.8], 0.5, [0, 1], "0/0 never suppresses"),
    # zero-area box inside a real one: inter 0 -> IoU 0
    ([[0, 0, 4, 4], [2, 2, 2, 3]], [0.9, 0.8], 0.0, [0, 1], "IoU 0 is not > 0"),
    # chain: A=[0,0,10,10], B=[4,0,14,10] (inter 60, union 140, IoU 0.4286), C=[8,0,18,10] (IoU(B,C) 0.4286, IoU(A,C) = 20/180 = 0.111).
    # thr 0.4: A kills B; B, being dead, kills nobody; C survives because of A only 0.111.
    ([[0, 0, 10, 10], [4, 0, 14, 10], [8, 0, 18, 10]], [0.9, 0.8, 0.7], 0.4, [0, 2], "a suppressed box suppresses nobody"),
    # same with scores reversed: C first, kills B, A survives -> output order by score: C, A
    ([[0, 0, 10, 10], [4, 0, 14, 10], [8, 0, 18, 10]], [0.7, 0.8, 0.9], 0.4, [2, 0], "output is score-ordered"),
    # inverted box (x2 < x1): area -4, inter w = max(0, min(1,4)-max(3,0)) = 0 -> IoU = 0/(16-4-0) = 0
    ([[0, 0, 4, 4], [3, 0, 1, 2]], [0.9, 0.8], 0.1, [0, 1], "negative area does not suppress"),
]
