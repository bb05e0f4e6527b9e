#!/usr/bin/env python3
"""Which 16-bit rounding carries the forward's hidden-vector error?    CPU only.

The reference graph of one classifier call (``/root/reference/model.py:138-170``) in float64 on a few pairs of the OpenImages 4 x 100
sampled case, with the f16 roundings the device path applies injected ONE AT A TIME (and all together): packed input x, conv1
weights, tanh output a, conv2 weights, conv2 halves U / V (rounded SEPARATELY, as the device stores them), z, conv3 weights, y,
fc1 weights, h1, fc2 weights.  Prints max |hidden - hidden64| / max |hidden64| per source: the measure of
``tests/test_sampled_oracle_gpu.py`` (bar 1e-3).  Accumulation is exact here (float64), so the table isolates operand rounding.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import relhead_oracle as O                                                    # noqa: E402
from scene_graph_commonsense_amd.synthetic import make_state_dict                         # noqa: E402
from tests.sampled_case import HEAD_GAIN, SD_SEED, host_case                              # noqa: E402


def r16(t, on):
    return t.half().double() if on else t


def hidden(sd, hs, ho, lab, src):
    """hidden [b,512] in float64 with the roundings named in ``src`` applied."""
    g = lambda k, tag: r16(sd[k], tag in src)
    hs, ho = r16(hs, "x" in src), r16(ho, "x" in src)
    a = r16(torch.tanh(F.conv2d(hs, g("conv1_1.weight", "w1"), sd["conv1_1.bias"])), "a" in src)
    b = r16(torch.tanh(F.conv2d(ho, g("conv1_2.weight", "w1"), sd["conv1_2.bias"])), "a" in src)
    w2 = g("conv2_1.weight", "w2")
    U = r16(F.conv2d(a, w2[:, :128], None, padding=1), "uv" in src)
    V = r16(F.conv2d(b, w2[:, 128:], sd["conv2_1.bias"], padding=1), "uv" in src)
    z = r16(F.max_pool2d(F.relu(U + V), 2, 2), "z" in src)
    y = r16(F.max_pool2d(F.relu(F.conv2d(z, g("conv3_1.weight", "w3"), sd["conv3_1.bias"], padding=1)), 2, 2), "y" in src)
    h1 = r16(F.relu(F.linear(y.reshape(y.shape[0], -1), g("fc1.weight", "wf1"), sd["fc1.bias"])), "h1" in src)
    w = sd["fc2.weight"]
    wm = r16(w[:, :4096], "wf2" in src)
    return F.relu(F.linear(h1, wm) + F.linear(lab, w[:, 4096:]) + sd["fc2.bias"])


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "configs4_oiv6_4x100"
    n_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    torch.set_num_threads(8)
    hc = host_case(name)
    cfg, batch, pidx = hc["cfg"], hc["batch"], hc["pidx"]
    sd = {k: v.double() for k, v in make_state_dict(cfg, seed=SD_SEED, head_gain=HEAD_GAIN).items()}
    feat = torch.cat([batch.image_feature, batch.image_depth], dim=1).double()
    masks = [O.build_masks(b, cfg.feature_size) for b in batch.bbox]
    rows = [r for r in hc["rows"]][:: max(1, len(hc["rows"]) // n_calls)][:n_calls]
    srcs = ["x", "w1", "a", "w2", "uv", "z", "w3", "y", "wf1", "h1", "wf2"]
    worst = {s: 0.0 for s in srcs + ["all", "all but conv1/conv2 operands (x w1 a w2)", "all but uv", "all but x w1 a w2 uv"]}
    scale = 0.0
    for r in rows:
        img, sub, obj = pidx.image[r], pidx.sub[r], pidx.obj[r]          # global object indices
        off = np.concatenate([[0], np.cumsum(hc["nobj"])])
        hs = torch.stack([feat[i] * masks[i][sg - off[i]] for i, sg in zip(img, sub)])
        ho = torch.stack([feat[i] * masks[i][og - off[i]] for i, og in zip(img, obj)])
        cats = torch.cat(batch.categories)
        lab = torch.cat([F.one_hot(cats[sub], cfg.num_classes), F.one_hot(cats[obj], cfg.num_classes)], 1).double()
        if lab.shape[1] < sd["fc2.weight"].shape[1] - 4096:
            lab = F.pad(lab, (0, sd["fc2.weight"].shape[1] - 4096 - lab.shape[1]))      # VG: super-category columns left zero (labels do not matter here)
        ref = hidden(sd, hs, ho, lab, set())
        sc = float(ref.abs().max())
        scale = max(scale, sc)
        for key in worst:
            if key == "all":
                src = set(srcs)
            elif key.startswith("all but conv1"):
                src = set(srcs) - {"x", "w1", "a", "w2"}
            elif key == "all but uv":
                src = set(srcs) - {"uv"}
            elif key == "all but x w1 a w2 uv":
                src = set(srcs) - {"x", "w1", "a", "w2", "uv"}
            else:
                src = {key}
            worst[key] = max(worst[key], float((hidden(sd, hs, ho, lab, src) - ref).abs().max()) / sc)
    print("# %s, %d reference calls (%d pairs), hidden scale %.3f; max |hidden - f64| / scale per f16 rounding source" % (name, len(rows), sum(len(r) for r in rows), scale))
    for k, v in worst.items():
        print("%-48s %.2e" % (k, v))


if __name__ == "__main__":
    main()
