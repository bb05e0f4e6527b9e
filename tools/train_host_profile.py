#!/usr/bin/env python3
"""Host-side cost of one realistic training minibatch (GPU box): train_minibatch from the raw SceneBatch (flatten, targets, loss
coefficients, CSR lists) - bench.py builds the scene once outside its timed loop, a real data loader cannot."""
import cProfile
import os
import pstats
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from scene_graph_commonsense_amd.model import BayesianRelationClassifier              # noqa: E402
from scene_graph_commonsense_amd.pair_loop import train_minibatch                     # noqa: E402
from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict  # noqa: E402

cfg = HeadConfig()
model = BayesianRelationClassifier(cfg.args(run_mode="train")).cuda()
model.load_state_dict(make_state_dict(cfg, seed=0))
model.train()
opt = torch.optim.SGD(model.parameters(), lr=1e-9, momentum=0.9, weight_decay=1e-4)
batch = make_scene_batch(cfg, [64] * 8, seed=3, connect_frac=0.02)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    train_minibatch(model, batch, opt)
    torch.cuda.synchronize()
    print("train_minibatch from raw batch: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
train_minibatch(model, batch, opt)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
