#!/usr/bin/env python3
"""Does an HBM-bound kernel hide under an MFMA-bound GEMM on this chip?  (GPU box.)

The ping-pong GEMM block uses 212 VGPRs x 2 waves per SIMD and 128 KiB of LDS: a third wave with <= 80 registers and no LDS fits
beside it, so a streaming kernel on a second HIP stream COULD run in the shadow of the GEMM.  Times, with HIP events around the
whole region: the GEMM alone, a streaming copy alone (and the repo's im2col-like 9-row gather), both launched back to back on ONE
stream, and both on TWO streams.  Also two GEMMs on two streams (what the two-stream backward does with the data- and the
weight-gradient GEMM over the listed windows).
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scene_graph_commonsense_amd import _lib

lib = _lib.load()
M, N, K = 1 << 20, 4608, 1024               # the column form of the conv3 data gradient over the listed windows
A = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
B = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
A2 = (torch.rand(M // 2, K, device="cuda") * 2 - 1).bfloat16()
C2 = torch.empty(M // 2, N, dtype=torch.bfloat16, device="cuda")
src = torch.empty(2 << 30, dtype=torch.uint8, device="cuda").random_()          # 2 GiB read + 2 GiB written per copy
dst = torch.empty_like(src)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def gemm(a, c, m):
    L = ctypes.c_long
    _lib.check(lib.sgc_dbg_gemm_nt(1, _lib.ptr(a), _lib.ptr(B), _lib.ptr(c), m, N, K, L(K), L(K), L(N), _lib.ptr(None), _lib.stream_ptr()), "gemm")


def copies(n=4):
    for _ in range(n):
        dst.copy_(src)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def two_streams(f1, f2):
    def run():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            f1()
        with torch.cuda.stream(s2):
            f2()
        cur.wait_stream(s1); cur.wait_stream(s2)
    return run


g = lambda: gemm(A, C, M)
g2 = lambda: gemm(A2, C2, M // 2)
t_g = timed(g)
t_c = timed(copies)
t_seq = timed(lambda: (g(), copies()))
t_par = timed(two_streams(g, copies))
print("GEMM %dx%dx%d bf16 alone            %7.3f ms  (%.0f TFLOP/s)" % (M, N, K, t_g, 2.0 * M * N * K / t_g / 1e9))
print("4 x 2 GiB copies alone                     %7.3f ms  (%.2f TB/s read+write)" % (t_c, 4 * 2 * (2 << 30) / t_c / 1e9))
print("both, one stream                           %7.3f ms" % t_seq)
print("both, two streams                          %7.3f ms   (max of the two = %.3f, sum = %.3f)" % (t_par, max(t_g, t_c), t_g + t_c))
t_g2 = timed(g2)
t_gg_seq = timed(lambda: (g2(), g()))
t_gg_par = timed(two_streams(g, g2))
print("second GEMM (half the rows) alone          %7.3f ms" % t_g2)
print("two GEMMs, one stream / two streams        %7.3f / %7.3f ms" % (t_gg_seq, t_gg_par))
