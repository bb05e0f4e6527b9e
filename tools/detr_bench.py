#!/usr/bin/env python3
"""Time of the DETR-101 feature extractor (scene_graph_commonsense_amd/detr.py, random weights) for one minibatch of 8 images of
1024 x 1024 on the GPU: f32 as the reference runs it, and under bf16 autocast (channels-last)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from scene_graph_commonsense_amd.detr import DETR
    torch.manual_seed(0)
    m = DETR(151).cuda().eval()
    x = torch.randn(8, 3, 1024, 1024, device="cuda")
    for name, kw, fmt in (("f32", {}, torch.contiguous_format), ("bf16 autocast, channels-last", {"autocast": torch.bfloat16}, torch.channels_last)):
        mm = m.to(memory_format=fmt)
        xx = x.contiguous(memory_format=fmt)
        for _ in range(3):
            mm.encode(xx, **kw)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(5):
            mm.encode(xx, **kw)
        torch.cuda.synchronize()
        print("%s: %.1f ms per 8 images" % (name, (time.time() - t) / 5 * 1e3))


if __name__ == "__main__":
    main()
