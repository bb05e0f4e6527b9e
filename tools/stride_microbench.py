#!/usr/bin/env python3
"""Does a power-of-two operand row pitch cost the ping-pong NT block anything (cache-channel conflicts of its 128-byte row pieces)?
C[M, N] = A[M, K] B[N, K]^T, bf16, with the operands' row pitches padded by `pad` elements.  GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scene_graph_commonsense_amd import _lib

lib = _lib.load()
import ctypes
for (M, N, K) in ((303104, 4096, 1024), (32768, 4096, 4096), (131072, 1024, 4608)):
    for pad_a, pad_b in ((0, 0), (64, 0), (0, 64), (64, 64), (8, 8), (0, 0)):
        lda, ldb = K + pad_a, K + pad_b
        A = (torch.rand(M, lda, device="cuda") * 2 - 1).bfloat16()
        B = (torch.rand(N, ldb, device="cuda") * 2 - 1).bfloat16()
        C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        call = lambda: lib.sgc_dbg_gemm_nt_ld(_lib.ptr(A), _lib.ptr(B), _lib.ptr(C), M, N, K, ctypes.c_long(lda), ctypes.c_long(ldb), _lib.stream_ptr())
        for _ in range(3):
            assert call() == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            call()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        print("M %6d N %4d K %4d  lda K+%-2d ldb K+%-2d  %7.3f ms  %7.1f TFLOP/s" % (M, N, K, pad_a, pad_b, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
        del A, B, C
