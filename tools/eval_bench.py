#!/usr/bin/env python3
"""Time the whole evaluation path of one minibatch (GPU box): fused forward of all ordered pairs + overlap filter + evaluator feed
(evaluate_minibatch) and Evaluator.compute (per-image top-100 ranking + hit matching), 8 images x 64 objects."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from scene_graph_commonsense_amd.evaluator import Evaluator, Evaluator_Top3          # noqa: E402
from scene_graph_commonsense_amd.model import BayesianRelationClassifier              # noqa: E402
from scene_graph_commonsense_amd.pair_loop import evaluate_minibatch                  # noqa: E402
from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict  # noqa: E402

FX = os.path.join(REPO, "tests", "golden", "ref_fixtures") + os.sep
cfg = HeadConfig()
args = cfg.args(fixtures=FX)
model = BayesianRelationClassifier(args).cuda()
model.load_state_dict(make_state_dict(cfg, seed=0))
model.eval()
batch = make_scene_batch(cfg, [64] * 8, seed=3, connect_frac=0.02)
for it in range(6):
    skip = it >= 3
    ev = Evaluator(args, cfg.num_relations, 0.5, [20, 50, 100])
    t3 = Evaluator_Top3(args, cfg.num_relations, 0.5, [20, 50, 100])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    evaluate_minibatch(model, batch, ev, t3, skip_filtered=skip)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    r = ev.compute(per_class=True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    r3 = t3.compute(per_class=True)
    torch.cuda.synchronize(); t3_ = time.perf_counter()
    print("iter %d skip_filtered=%s: evaluate_minibatch %.1f ms, Evaluator.compute %.1f ms, Evaluator_Top3.compute %.1f ms  (32256 pairs, R@20/50/100 %s)"
          % (it, skip, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3_ - t2) * 1e3, [round(float(x), 3) for x in r[0]]))

# the drivers' loop: minibatch k+1 is flattened while the device scores minibatch k (pair_loop.MinibatchLookahead) against the plain loop
from scene_graph_commonsense_amd.pair_loop import MinibatchLookahead          # noqa: E402
from scene_graph_commonsense_amd.pairs import flatten_scene                   # noqa: E402
batches = [make_scene_batch(cfg, [64] * 8, seed=30 + i, connect_frac=0.02) for i in range(4)] * 5
for b in batches[:4]:
    b.image_feature, b.image_depth = b.image_feature.cuda(), b.image_depth.cuda()
for mode in ("plain", "lookahead", "plain", "lookahead"):
    ev = Evaluator(args, cfg.num_relations, 0.5, [20, 50, 100])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if mode == "plain":
        for b in batches:
            evaluate_minibatch(model, b, ev, None, skip_filtered=False)
    else:
        ahead = MinibatchLookahead(batches, lambda i, b: (b, flatten_scene(cfg, b, "cuda:0")))
        for _, (b, sc) in ahead:
            evaluate_minibatch(model, b, ev, None, skip_filtered=False, scene=sc, while_running=ahead.fetch_next)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    r = ev.compute(per_class=True)
    print("%-9s loop over %d minibatches: %.2f ms per minibatch (R@20/50/100 %s)" % (mode, len(batches), (t1 - t0) * 1e3 / len(batches), [round(float(x), 4) for x in r[0]]))

if os.environ.get("EVAL_PROFILE"):
    import cProfile
    import pstats
    ev = Evaluator(args, cfg.num_relations, 0.5, [20, 50, 100])
    pr = cProfile.Profile()
    pr.enable()
    evaluate_minibatch(model, batch, ev, None)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
