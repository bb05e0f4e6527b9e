#!/usr/bin/env python3
"""SHA-1 over the loss and every gradient tensor of four fixed scenes (the vg_full fixture and three synthetic minibatches, one with an
image of 100 objects): the check that a kernel rewrite which is meant to leave every sum in place did (GPU box).  Compare the line
across builds / switches; the value depends on the library only."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_cases import load_case                                               # noqa: E402
from tests.test_backward_gpu import run_train                                          # noqa: E402
from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict   # noqa: E402

h = hashlib.sha1()
cfg, sd, batch, gold = load_case("vg_full")
loss, grads = run_train(cfg, sd, batch)
for k in sorted(grads):
    h.update(grads[k].numpy().tobytes())
cfg2 = HeadConfig()
sd2 = make_state_dict(cfg2, seed=3)
for shape, seed in (([30, 64, 9, 41], 4), ([64] * 8, 5), ([100, 3, 70], 6)):
    b2 = make_scene_batch(cfg2, shape, seed=seed, connect_frac=0.05)
    l2, g2 = run_train(cfg2, sd2, b2)
    h.update(str(l2).encode())
    for k in sorted(g2):
        h.update(g2[k].numpy().tobytes())
print("HASH", loss, h.hexdigest())
