#!/usr/bin/env python3
"""Short-K products (K = 1024: fc1 over window-major rows, conv3 data gradient in column form) on the block configurations of
gemm_nt.h: does a second resident workgroup per CU (128 x 128 tiles) hide the tile stores that the 256 x 256 ping-pong block exposes?
Needs a library built with SGC_EXPERIMENTS=1 (SGC_GEMM_CFG selects the block: 1 = 128x128, 2 = 2-stage 256x256, 5 = ping-pong).

    SGC_EXPERIMENTS=1 SGC_GEMM_CFG=1 python tools/short_k_microbench.py [M N K]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scene_graph_commonsense_amd import _lib

lib = _lib.load()
M, N, K = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (294912, 4096, 1024)
A = (torch.rand(M, K, device="cuda") * 2 - 1).half()
B = (torch.rand(N, K, device="cuda") * 0.06 - 0.03).half()
C = torch.empty(M, N, dtype=torch.float16, device="cuda")
L = ctypes.c_long


def run():
    _lib.check(lib.sgc_dbg_gemm_nt(0, _lib.ptr(A), _lib.ptr(B), _lib.ptr(C), M, N, K, L(K), L(K), L(N), None, _lib.stream_ptr()), "gemm")


for rep in range(2):
    run(); run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    ref = A[:256].float() @ B.float().t()
    err = float((C[:256].float() - ref).abs().max() / ref.abs().max())
    print("SGC_GEMM_CFG=%s  %d x %d x %d f16 -> f16: %7.3f ms  %7.1f TFLOP/s  (%.2f GB out, max rel err %.1e)" %
          (os.environ.get("SGC_GEMM_CFG", "-"), M, N, K, ms, 2.0 * M * N * K / ms / 1e9, M * N * 2 / 1e9, err), flush=True)
