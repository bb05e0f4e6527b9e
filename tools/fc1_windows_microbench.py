#!/usr/bin/env python3
"""Where the time of fc1 over the window-major rows goes (GPU box): the grouped product ``owm [rows, 4096] f32 = ywm [rows, 1024] x
W_g^T`` (``sgc_fc1_windows_gemm``: 64 window groups, K = 1024, 5 GB of f32 products per launch at the benchmark's 303 k rows) runs
at ~0.30 of the f16 peak while the same block reaches 0.54-0.58 on long-K products.  Variants: the epilogue (4-byte stores / 16-byte
stores / no stores / f16 output) and the weight layout (the product's [4096][(window, channel)] rows, 2 KB pieces 128 KB apart,
against one contiguous [4096][1024] slab per window group).  Random f16 operands.

    python tools/fc1_windows_microbench.py [rows_per_group]      (default 4736 = 303 104 rows)

The store-policy / tile-contiguous / stagger / wall-clock variants are compiled only into a library built with SGC_EXPERIMENTS=1
(``SGC_EXPERIMENTS=1 python -m scene_graph_commonsense_amd.build --force``); the product library runs them as the plain variant.
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scene_graph_commonsense_amd import _lib

lib = _lib.load()
rpg = (int(sys.argv[1]) if len(sys.argv) > 1 else 4736) // 256 * 256
rows = rpg * 64
dev = "cuda"
ywm = (torch.rand(rows, 1024, device=dev) * 2 - 1).half()
w_rows = (torch.rand(4096, 65536, device=dev) * 0.06 - 0.03).half()                       # the product's layout
w_slab = w_rows.view(4096, 64, 1024).permute(1, 0, 2).contiguous()                       # [group][4096][1024]
tg = torch.arange(64, device=dev, dtype=torch.int32).repeat_interleave(rpg // 256).contiguous()
LDC = int(os.environ.get("MB_LDC", "4096"))          # row pitch of the products in elements (4096: the product's; e.g. 4160 = + 256 B)
MODES = [int(x) for x in os.environ.get("MB_MODES", "").split(",") if x]
STAG, PHASES = [int(x) for x in os.environ.get("MB_STAGGER", "0,1").split(",")]     # sleep units (~4 us) per phase step, phases (power of 2)
o32 = torch.empty(rows, LDC, dtype=torch.float32, device=dev)
o16 = torch.empty(rows, LDC, dtype=torch.float16, device=dev)
flop = 2.0 * rows * 1024 * 4096
modes = {0: "f32, 4-byte stores", 1: "f32, 16-byte stores", 2: "no stores", 3: "f16 output (LDS-staged)", 4: "f32, 16-byte stores nt",
         5: "f32, 16-byte stores sc1", 6: "f32, 16-byte stores sc0 sc1", 7: "f32, 16-byte stores sc0 sc1 nt",
         11: "f32, tile-contiguous output"}
layouts = {"rows [4096][65536]": (w_rows, 65536, 1024), "slabs [64][4096][1024]": (w_slab, 1024, 4096 * 1024)}
L = ctypes.c_long
print("# %d rows (%d per window group), %.2f TFLOP, %.2f GB of f32 products, row pitch %d elements, stagger %d x %d phases" % (rows, rpg, flop / 1e12, rows * 4096 * 4 / 1e9, LDC, STAG, PHASES))
ref = None
clk = torch.zeros(4, dtype=torch.int64, device=dev)        # per-block wall clocks (100 MHz): main loop, store issue, blocks
for rep in range(2):
    for lname, (w, ldb, gs) in layouts.items():
        for mode, mname in modes.items():
            if MODES and mode not in MODES:
                continue
            out = o16 if mode == 3 else o32
            def run():
                _lib.check(lib.sgc_dbg_fc1_windows_gemm(_lib.ptr(ywm), _lib.ptr(w), _lib.ptr(tg), _lib.ptr(out), rows, L(ldb), L(gs), L(LDC), mode, STAG, PHASES, _lib.ptr(clk),
                                                        _lib.stream_ptr()), "sgc_dbg_fc1_windows_gemm")
            run(); run()
            torch.cuda.synchronize()
            clk.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                run()
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 5
            note = ""
            if rep == 0 and mode in (0, 1, 4, 5, 6, 7):
                g = 37
                want = ywm[g * rpg:g * rpg + 256].float() @ w_rows[:, g * 1024:(g + 1) * 1024].float().t()
                got = o32[g * rpg:g * rpg + 256, :4096]
                note = "  max rel err %.1e" % float((got - want).abs().max() / want.abs().max())
                if ref is None:
                    ref = o32[:, :4096].clone()
                else:
                    note += ", bit-identical to the first variant: %s" % bool(torch.equal(ref, o32[:, :4096]))
            c = clk.tolist()
            per = "  per tile: main loop %.1f us, store issue %.1f us, x %.1f tiles per CU = %.2f ms" % (
                c[0] / c[2] / 100.0, c[1] / c[2] / 100.0, c[2] / 5 / 256.0, (c[0] + c[1]) / 5 / 256.0 / 1e5) if c[2] else ""
            print("%-24s %-26s %7.3f ms  %7.1f TFLOP/s%s%s" % (lname, mname, ms, flop / ms / 1e9, per, note), flush=True)
