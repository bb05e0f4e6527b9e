#!/usr/bin/env python3
"""Peak device memory of the training step at the bench workload, and the largest workspace buffers."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.optim import FusedSGD
    from scene_graph_commonsense_amd.pair_loop import train_minibatch
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
    cfg = HeadConfig()
    model = BayesianRelationClassifier(cfg.args()).cuda()
    model.load_state_dict(make_state_dict(cfg, seed=1, head_gain=4.0))
    model.train()
    opt = FusedSGD(model.parameters(), lr=1e-5, momentum=0.9, weight_decay=1e-4)
    batch = make_scene_batch(cfg, [64] * 8, seed=1000, connect_frac=0.02)
    for _ in range(3):
        train_minibatch(model, batch, opt)
    torch.cuda.synchronize()
    print("peak allocated %.1f GB, reserved %.1f GB" % (torch.cuda.max_memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9))
    eng = model.refresh_weights(backward=True)
    bufs = {}
    for ws in {id(eng.ws): eng.ws, id(eng.scratch): eng.scratch}.values():
        for k, t in ws.bufs.items():
            bufs[k] = t.numel() * t.element_size()
    for k, v in sorted(bufs.items(), key=lambda kv: -kv[1])[:16]:
        print("  %-12s %6.2f GB" % (k, v / 1e9))
    print("  workspace total %.1f GB" % (sum(bufs.values()) / 1e9))


if __name__ == "__main__":
    main()
