#!/usr/bin/env python3
"""Decompose the conv3 forward launch against its sibling, the data gradient (GPU box): element type (f16 / bf16), shape
(Cin 512 x N 1024: 72 K steps, 4 N tiles per image  vs  Cin 1024 x N 512: 144 K steps, 2 N tiles) and epilogue (plain store
vs bias + ReLU + max-pool with three outputs), all through the same halo-staged ping-pong block.  P pairs = P images of 16x16."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scene_graph_commonsense_amd import _lib

lib = _lib.load()
P = int(os.environ.get("MB_P", "16128"))
g = torch.Generator(device="cuda").manual_seed(0)


def padded(Cin, dtype, relu):
    x = torch.randn(P, 16, 16, Cin, device="cuda", generator=g)
    if relu:
        x = x.clamp(min=0)
    z = torch.zeros(P, 18, 18, Cin, dtype=dtype, device="cuda")
    z[:, 1:17, 1:17] = x.to(dtype)
    return z


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


rows = []
for rep in range(2):
    for elem, dt in ((0, torch.float16), (1, torch.bfloat16)):
        for Cin, N in ((512, 1024), (1024, 512)):
            for relu in (True, False):
                A = padded(Cin, dt, relu)
                W = (torch.randn(N, 9 * Cin, device="cuda", generator=g) * 0.02).to(dt)
                C = torch.empty(P * 256, N, dtype=dt, device="cuda")
                ms = timeit(lambda: _lib.check(lib.sgc_dbg_conv_nt(elem, _lib.ptr(A), _lib.ptr(W), _lib.ptr(C), P, 4, Cin, N, None,
                                                                  _lib.stream_ptr()), "conv"))
                tf = 2.0 * P * 256 * N * 9 * Cin / ms / 1e9
                rows.append((rep, "store", "f16" if elem == 0 else "bf16", Cin, N, "relu-like" if relu else "gaussian", ms, tf))
                del A, W, C
    # the real forward entry point: f16, pool epilogue with 1 / 2 / 3 outputs
    z = padded(512, torch.float16, True)
    W = (torch.randn(1024, 4608, device="cuda", generator=g) * 0.02).half()
    b3 = torch.zeros(1024, device="cuda")
    y = torch.empty(P * 64, 1024, dtype=torch.float16, device="cuda")
    yb = torch.empty(P * 64, 1024, dtype=torch.bfloat16, device="cuda")
    am = torch.empty(P * 64, 1024, dtype=torch.uint8, device="cuda")
    for label, a_, b_ in (("pool y", None, None), ("pool y+argmax", am, None), ("pool y+argmax+bf16", am, yb)):
        ms = timeit(lambda: _lib.check(lib.sgc_conv3_relu_pool(_lib.ptr(z), _lib.ptr(W), _lib.ptr(b3), _lib.ptr(y), _lib.ptr(a_), _lib.ptr(b_),
                                                               P, _lib.stream_ptr()), "conv3"))
        rows.append((rep, label, "f16", 512, 1024, "relu-like", ms, 2.0 * P * 256 * 1024 * 4608 / ms / 1e9))
    del z, W, y, yb, am
for r in rows:
    print("rep %d  %-20s %-4s Cin %4d N %4d %-9s %8.3f ms  %7.1f TFLOP/s" % r)
