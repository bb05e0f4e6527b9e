#!/usr/bin/env python3
"""tools/pmc_summary.py output with SQ_INSTS_VALU -> per kernel: duration, the time its vector instructions need with every SIMD issuing
one per 4 cycles (1024 SIMDs, 2.2 GHz), their ratio, and vector instructions per wavefront.  A launch whose ratio is above ~0.4 is bound
by its instruction count: compare the count with what the source should need (profiles/r05_expand_ab.txt: 318 against 170)."""
import csv
import sys

rows = [r for r in csv.reader(l for l in open(sys.argv[1]) if not l.startswith("#"))][1:]
d = {}
for k, n, c, v in rows:
    d.setdefault(k, {})[c] = float(v)
    d[k]["n"] = int(n)
out = []
for k, v in d.items():
    if "SQ_INSTS_VALU" in v and "duration_ms" in v and v["duration_ms"] > 0:
        valu_ms = v["SQ_INSTS_VALU"] * 4 / 1024 / 2.2e9 * 1e3
        out.append((v["duration_ms"] * v["n"], v["duration_ms"], v["n"], valu_ms, valu_ms / v["duration_ms"],
                    v["SQ_INSTS_VALU"] / max(v.get("SQ_WAVES", 0), 1), k[:90]))
out.sort(reverse=True)
print("total_ms   ms/launch  launches  valu_ms@100%  ratio  valu_instr/wave  kernel")
for o in out:
    print("%8.3f %9.3f %8d %12.3f %6.2f %14.0f  %s" % o)
