#!/usr/bin/env python3
"""conv3 data gradient over listed windows: the column form (rows = window pixels, N = 9 x 512, K = 1024) against the patch form
(rows = windows, 16 patch pixels x 512, K = 1024 x 1 / 2 / 4) with the layouts of its operands as parameters (GPU box).

    python tools/dgrad_patch_microbench.py [entries]        (default 228096: the benchmark's list of convolved windows)
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scene_graph_commonsense_amd import _lib

lib = _lib.load()
E = (int(sys.argv[1]) if len(sys.argv) > 1 else 228096) // 256 * 256
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
w3 = torch.randn(1024, 512, 3, 3, device=dev, generator=g) * 0.02
dy = (torch.randn(E, 4, 1024, device=dev, generator=g) * (torch.rand(E, 4, 1024, device=dev, generator=g) < 0.25)).bfloat16()   # 1 of 4 routes live
dy_q = dy.permute(1, 0, 2).contiguous()                       # [4][E][1024]: own-pixel-major
opts = lambda c: [(0, 0)] if c == 0 else ([(1, 2)] if c == 3 else [(0, c), (1, c - 1)])


def w3patch(pad):
    mats = []
    for py in range(4):
        for px in range(4):
            m = torch.cat([w3[:, :, ky, kx].t() for _, ky in opts(py) for _, kx in opts(px)], dim=1)      # [512][K]
            if pad:
                m = torch.cat([m, torch.zeros(512, pad, device=dev)], dim=1)
            mats.append(m.reshape(-1))
    return torch.cat(mats).bfloat16().contiguous()


w3col = w3.permute(2, 3, 1, 0).reshape(9 * 512, 1024).bfloat16().contiguous()
col = torch.empty(E * 4, 9 * 512, dtype=torch.bfloat16, device=dev)
patch = torch.empty(E * 16, 512, dtype=torch.bfloat16, device=dev)
L = ctypes.c_long
flop = 2.0 * E * 4 * 1024 * 9 * 512


def timed(fn, name):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 4
    print("%-64s %7.3f ms  %7.1f TFLOP/s" % (name, ms, flop / ms / 1e9), flush=True)


print("# %d listed windows, %.2f TFLOP; column form stores %.2f GB, patch form %.2f GB" % (E, flop / 1e12, E * 4 * 4608 * 2 / 1e9, E * 16 * 512 * 2 / 1e9))
ref = None
for rep in range(2):
    timed(lambda: _lib.check(lib.sgc_windows_dgrad_cols(_lib.ptr(dy), _lib.ptr(w3col), _lib.ptr(col), E * 4, _lib.stream_ptr()), "cols"), "column form")
    B = w3patch(0)
    out20 = torch.empty(E * 20, 512, dtype=torch.bfloat16, device=dev)
    for gn in (1, 2, 4, 8, 16, 32):          # XCD patch = (32 / gn) M tiles x gn virtual N tiles, unaligned walk
        timed(lambda: _lib.check(lib.sgc_dbg_dgrad_patches(_lib.ptr(dy), _lib.ptr(B), _lib.ptr(out20), E, L(4096), L(1024), gn << 16, 1,
                                                           _lib.stream_ptr()), "patches"), "patch form, K <= 2048, XCD patches of %d x %d tiles" % (32 // gn, gn))
    for split in (0, 1):
        out = torch.empty(E * (20 if split else 16), 512, dtype=torch.bfloat16, device=dev)
        timed(lambda: _lib.check(lib.sgc_dbg_dgrad_patches(_lib.ptr(dy), _lib.ptr(B), _lib.ptr(out), E, L(4096), L(1024), 0, split,
                                                           _lib.stream_ptr()), "patches"),
              "patch form%s" % (", centre pixels in two slots (K <= 2048)" if split else " (K = 1024 / 2048 / 4096)"))
        if rep == 0:
            o = out.float().view(E, -1, 512)
            if split:                                      # fold the two slots of the centre pixels
                idx, k = [], 0
                for pp in range(16):
                    c = (pp >> 2) in (1, 2) and (pp & 3) in (1, 2)
                    idx.append((k, k + 1) if c else (k,))
                    k += 2 if c else 1
                o = torch.stack([sum(o[:, j] for j in t) for t in idx], dim=1)
            if ref is None:
                ref = o
            else:
                print("    max |split - unsplit| / max |unsplit| = %.2e" % float((o - ref).abs().max() / ref.abs().max()))

# ---- weight gradient over the same list: im2col form against the patch form, and the number of K splits of the patch form
z = torch.randn(E // 64, 18, 18, 512, device=dev, generator=g).bfloat16()
gather = (torch.arange(E, device=dev, dtype=torch.int32) % 64) + (torch.arange(E, device=dev, dtype=torch.int32) // 64) * 64
gn = torch.tensor([E], device=dev, dtype=torch.int32)
dy2 = dy.view(E * 4, 1024)
slabs = torch.empty(64, 1024, 9 * 512, device=dev)
n = ctypes.c_int(0)
zcol = torch.empty(E * 4, 9 * 512, dtype=torch.bfloat16, device=dev)
zpatch = torch.empty(E, 16, 512, dtype=torch.bfloat16, device=dev)
for rep in range(2):
    timed(lambda: _lib.check(lib.sgc_windows_im2col(_lib.ptr(z), _lib.ptr(gather), _lib.ptr(gn), E, _lib.ptr(zcol), _lib.stream_ptr()), "im2col"), "im2col (36 rows per window)")
    timed(lambda: _lib.check(lib.sgc_windows_im2patch(_lib.ptr(z), _lib.ptr(gather), _lib.ptr(gn), E, _lib.ptr(zpatch), _lib.stream_ptr()), "im2patch"), "im2patch (16 rows per window)")
    timed(lambda: _lib.check(lib.sgc_windows_wgrad(_lib.ptr(dy2), _lib.ptr(zcol), _lib.ptr(slabs), E * 4, 0, ctypes.byref(n), _lib.stream_ptr()), "wgrad"),
          "weight gradient, im2col form, automatic splits")
    for sp in (0, 4, 7, 14, 28, 56):
        timed(lambda: _lib.check(lib.sgc_windows_wgrad_patch(_lib.ptr(dy2), _lib.ptr(zpatch), _lib.ptr(slabs), E * 4, sp, ctypes.byref(n), _lib.stream_ptr()), "wgrad"),
              "weight gradient, patch form, %s" % ("automatic splits" if sp == 0 else "%d splits" % sp))
        print("    -> %d slabs" % n.value)
