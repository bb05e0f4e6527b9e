#!/usr/bin/env python3
"""Box-size sensitivity of the training step (VERDICT r2 item 3 / task 2): the reference's cost does not depend on the boxes
(``model.py:138-150`` runs every pixel of every pair); this implementation's does - conv3 and fc1 run per pair only on the windows
inside BOTH objects' rectangles.  The sweep scales the benchmark's boxes about their centres and times the step on BOTH
organisations at every point (shared windows forced on / per-pair kernels forced), with the peak workspace, so that
``engine.TUNING.shared_max_fraction`` (the switch between them) is a measured number:

    python tools/box_sensitivity.py [--objects 64 --images 8 --steps 3] > profiles/r03_box_sweep.txt

A forced-shared point whose estimated workspace exceeds --max-gb is skipped (a 288 GB part cannot hold the column buffers of a
scene in which every window is pair-specific).
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", type=int, default=64)
    ap.add_argument("--images", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--scales", default="0.5,1.0,1.5,2.0,2.5,3.5,5.0,100")
    ap.add_argument("--max-gb", type=float, default=215.0)
    args = ap.parse_args()
    from bench import scale_boxes
    from scene_graph_commonsense_amd import engine
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.optim import FusedSGD
    from scene_graph_commonsense_amd.pair_loop import train_minibatch
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
    dev = torch.device("cuda:0")
    cfg = HeadConfig()
    model = BayesianRelationClassifier(cfg.args(run_mode="train")).to(dev)
    model.load_state_dict(make_state_dict(cfg, seed=0))
    model.train()
    opt = FusedSGD(model.parameters(), lr=1e-9, momentum=0.9, weight_decay=1e-4)
    eng = model.engine()
    print("# %d images x %d objects; ms per training step (forward + loss + backward + SGD from the raw minibatch), peak GB allocated" %
          (args.images, args.objects))
    print("# scale  pair-specific fraction | shared windows: ms  GB | per-pair kernels: ms  GB | default path")
    for sc in [float(x) for x in args.scales.split(",")]:
        batch = scale_boxes(make_scene_batch(cfg, [args.objects] * args.images, seed=1000, connect_frac=0.02), sc)
        batch.image_feature, batch.image_depth = batch.image_feature.to(dev), batch.image_depth.to(dev)
        scene = flatten_scene(cfg, batch, dev)
        P = scene.n_pairs
        frac = scene.shared_windows / (64.0 * P)
        est_gb = (0.95e6 * P + 0.09e6 * scene.shared_windows) * 1.15 / 1e9
        row = []
        for mode in ("shared", "pairs"):
            if mode == "shared" and est_gb > args.max_gb:
                row.append("   skipped (%3.0f GB est.)" % est_gb)
                continue
            eng.ws.bufs.clear()
            model._weights_version = None
            torch.cuda.empty_cache()
            with engine.tuning(shared_conv3=(mode == "shared"), shared_max_fraction=2.0):
                train_minibatch(model, batch, opt)
                torch.cuda.synchronize()
                torch.cuda.reset_peak_memory_stats(dev)
                t0 = time.time()
                for _ in range(args.steps):
                    train_minibatch(model, batch, opt)
                torch.cuda.synchronize()
                ms = (time.time() - t0) / args.steps * 1e3
            row.append("%8.2f ms %6.1f GB" % (ms, torch.cuda.max_memory_allocated(dev) / 1e9))
        default = "shared" if engine.shared_conv3_enabled(scene.shared_windows, P) else "per-pair"
        print("%6.1f   %.4f | %s | %s | %s" % (sc, frac, row[0], row[1], default), flush=True)


if __name__ == "__main__":
    main()
