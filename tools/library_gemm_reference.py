#!/usr/bin/env python3
"""Context for the roofline fractions: what the vendor GEMM (hipBLASLt / rocBLAS through torch.matmul) reaches on
this chip for the same contraction shapes, random operands.  Measurement only - nothing in the product path uses it."""
import torch

shapes = {  # name: (M, N, K, dtype)
    "fc1_fwd   [32256x65536]x[65536x4096]  f16": (32256, 4096, 65536, torch.float16),
    "fc1_dgrad [32256x4096]x[4096x65536]   bf16": (32256, 65536, 4096, torch.bfloat16),
    "fc1_wgrad [4096x32256]x[32256x65536]  bf16 (TN)": (4096, 65536, 32256, torch.bfloat16),
    "conv3 as plain GEMM [2.06Mx4608]x[4608x1024] f16 (quarter of the launch)": (2064384, 1024, 4608, torch.float16),
    "square 8192^3 bf16": (8192, 8192, 8192, torch.bfloat16),
}
for name, (M, N, K, dt) in shapes.items():
    if "TN" in name:
        a = (torch.rand(K, M, device="cuda") * 2 - 1).to(dt).t()
        b = (torch.rand(K, N, device="cuda") * 2 - 1).to(dt)
    else:
        a = (torch.rand(M, K, device="cuda") * 2 - 1).to(dt)
        b = (torch.rand(N, K, device="cuda") * 2 - 1).to(dt).t()
    for _ in range(2):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        c = a @ b
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("%-75s %8.2f ms  %7.1f TFLOP/s" % (name, ms, 2.0 * M * N * K / ms / 1e9))
    del a, b, c
