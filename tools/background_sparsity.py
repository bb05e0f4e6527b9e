#!/usr/bin/env python3
"""Spike: how much of conv3 / fc1 is pair-dependent at all?

Outside the boxes the masked object maps are the constant tanh(b1), so U_i + V_j equals a pair-independent background map
outside D = dil1(box_i) u dil1(box_j) (32-grid; conv2 is 3x3).  After ReLU + 2x2 pool, z_ij (16-grid) differs from the background
only on D16 = windows touching D; conv3 (3x3) output differs on dil1(D16); after the second pool, y_ij (8-grid) differs only on
D8 = windows touching dil1(D16).  A gathered conv3 forward has to compute 4*|D8| of its 256 output pixels per pair, and fc1's
contraction length shrinks to |D8|/64 of its 65536 (the rest is one pair-independent vector W1*Y_bg).

Prints the mean active fraction |D8|/64 over all ordered pairs for the synthetic boxes of bench.py and for heavier-tailed
"VG-like" box statistics (CPU only, no GPU needed).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def active_fraction(boxes):
    """boxes [n,4] int (x0,x1,y0,y1) on the 32-grid -> mean over ordered pairs of |D8|/64, |D16|/256."""
    n = len(boxes)
    d8 = np.zeros((n, 8, 8), dtype=bool)          # per object: D8 of the object alone (union over a pair = OR: all steps are monotone + local)
    d16 = np.zeros((n, 16, 16), dtype=bool)
    for k, (x0, x1, y0, y1) in enumerate(boxes):
        m = np.zeros((32, 32), dtype=bool)
        if x1 > x0 and y1 > y0:
            m[max(y0 - 1, 0):min(y1 + 1, 32), max(x0 - 1, 0):min(x1 + 1, 32)] = True          # dil1(box)
        a = m.reshape(16, 2, 16, 2).any(axis=(1, 3))                                            # D16
        d16[k] = a
        p = np.pad(a, 1)
        b = np.zeros_like(a)
        for dy in range(3):
            for dx in range(3):
                b |= p[dy:dy + 16, dx:dx + 16]                                                  # dil1(D16)
        d8[k] = b.reshape(8, 2, 8, 2).any(axis=(1, 3))                                          # D8
    u8 = (d8[:, None] | d8[None, :]).reshape(n, n, -1).mean(-1)
    u16 = (d16[:, None] | d16[None, :]).reshape(n, n, -1).mean(-1)
    off = ~np.eye(n, dtype=bool)
    return float(u8[off].mean()), float(u16[off].mean())


def vg_like_boxes(n, rng):
    """Objects sorted by area, area fraction log-normal around 6 % with a heavy tail (a few near-full-image boxes such as sky /
    building / wall), aspect ratio log-normal - a stand-in for Visual Genome statistics (no data offline)."""
    area = np.clip(np.exp(rng.normal(np.log(0.06), 1.2, n)), 0.002, 0.95)
    asp = np.exp(rng.normal(0.0, 0.5, n))
    w = np.clip(np.sqrt(area * asp) * 32, 1, 32)
    h = np.clip(np.sqrt(area / asp) * 32, 1, 32)
    x0 = rng.uniform(0, 32 - w)
    y0 = rng.uniform(0, 32 - h)
    b = np.stack([x0, x0 + w, y0, y0 + h], axis=1)
    b = np.stack([np.floor(b[:, 0]), np.ceil(b[:, 1]), np.floor(b[:, 2]), np.ceil(b[:, 3])], axis=1).astype(int)
    return b[np.argsort(-(b[:, 1] - b[:, 0]) * (b[:, 3] - b[:, 2]))]


def main():
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch
    cfg = HeadConfig()
    for n in (20, 36, 64):
        batch = make_scene_batch(cfg, [n] * 8, seed=1000)
        fr = [active_fraction(b.numpy()) for b in batch.bbox]
        print("synthetic boxes (bench.py), N=%d: active fc1/conv3 fraction |D8|/64 = %.3f, z-level |D16|/256 = %.3f, mean box area %.3f"
              % (n, np.mean([f[0] for f in fr]), np.mean([f[1] for f in fr]),
                 np.mean([((b[:, 1] - b[:, 0]) * (b[:, 3] - b[:, 2])).float().mean() / 1024 for b in batch.bbox])))
    rng = np.random.default_rng(0)
    for n in (20, 36):
        fr = [active_fraction(vg_like_boxes(n, rng)) for _ in range(16)]
        print("VG-like boxes, N=%d: |D8|/64 = %.3f, |D16|/256 = %.3f" % (n, np.mean([f[0] for f in fr]), np.mean([f[1] for f in fr])))


if __name__ == "__main__":
    main()
