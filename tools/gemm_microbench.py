#!/usr/bin/env python3
"""Attribute the time of the 256x256 NT GEMM main loop (GPU box): normal vs no-loads vs no-loads-no-barrier vs
loads-only.  Random bf16 operands (zero-filled data inflates clocks).
Variants 4, 8, 9, 10 (the rejected main loops: 4-stage ring, 4 waves x 128x128, one-phase ping-pong) are compiled only into a
library built with SGC_EXPERIMENTS=1 (``SGC_EXPERIMENTS=1 python -m scene_graph_commonsense_amd.build --force``)."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scene_graph_commonsense_amd import _lib

lib = _lib.load()
M, N, K = 32768, 4096, int(os.environ.get("MB_K", "8192"))
A = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
B = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
names = {0: "normal", 1: "no global loads", 2: "no loads, no barriers", 3: "loads + barriers only", 4: "4-stage ring BK=32", 5: "A staged 1 tile in 9", 6: "ping-pong", 7: "ping-pong, no global loads", 8: "4 waves x 128x128, compiler schedule", 9: "4 waves x 128x128, sched_group_barrier interleave", 10: "ping-pong, one phase per K tile (32 MFMAs per slot)"}
for rep in range(2):
    for abl in [int(x) for x in os.environ.get("MB_ABL", "0,1,2,3,4,5,6,7").split(",")]:
        for _ in range(2):
            lib.sgc_dbg_gemm_nt_abl(abl, _lib.ptr(A), _lib.ptr(B), _lib.ptr(C), M, N, K, _lib.stream_ptr())
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            lib.sgc_dbg_gemm_nt_abl(abl, _lib.ptr(A), _lib.ptr(B), _lib.ptr(C), M, N, K, _lib.stream_ptr())
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        note = ""
        if abl in (0, 6, 8, 9, 10) and rep == 0:          # the variants that compute the real product: spot-check 256 rows
            ref = A[:256].float() @ B.float().t()
            note = "  max rel err %.1e" % ((C[:256].float() - ref).abs().max() / ref.abs().max()).item()
        print("abl=%d %-24s %8.3f ms  %7.1f TFLOP/s-equivalent%s" % (abl, names[abl], ms, 2.0 * M * N * K / ms / 1e9, note))
