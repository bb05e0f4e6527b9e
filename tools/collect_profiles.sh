#!/bin/bash
# Collect the profile set kept under profiles/ (run on the GPU box from the repo root):
#   bash tools/collect_profiles.sh <tag>     ->  gpurun_out/<tag>_{bench.json,bench_under_rocprof.json,kernel_stats.csv,pmc_f.csv,pmc_w.csv,pmc_s.csv}
# Counters are collected in their own passes with --kernel-trace only (never together with API / system traces).
set -u
TAG=${1:-rXX}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# per-kernel passes run the backward on ONE stream (durations of launches that overlap on two streams mean little); the bench
# line above is the shipped default (two streams)
export SGC_BWD_STREAMS=0
rm -rf /tmp/prof_ks /tmp/prof_f /tmp/prof_w /tmp/prof_s
rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o ks -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-sensitivity > /tmp/ks.log 2>&1
grep '^{"metric"' /tmp/ks.log | tail -1 > "$OUT/${TAG}_bench_under_rocprof.json"
# WARM-ONLY statistics: the two warm-up steps (code-object load, cold caches, one-time zeroing of the workspace) are dropped, so the
# averages are comparable with the HIP-event means of the bench line (the cold first call spread them by 13 %)
python3 "$R/tools/rocpd_summary.py" "$(find /tmp/prof_ks -name '*.db' | head -1)" "$OUT/${TAG}_kernel_stats.csv" --skip-steps 3     # 1 flatten before the loop + 2 warm-up steps
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -o f --output-format csv -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-sensitivity > /tmp/f.log 2>&1
python3 "$R/tools/pmc_summary.py" "$(find /tmp/prof_f -name '*counter_collection.csv' | head -1)" "$OUT/${TAG}_pmc_f.csv"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -o w --output-format csv -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-sensitivity > /tmp/w.log 2>&1
python3 "$R/tools/pmc_summary.py" "$(find /tmp/prof_w -name '*counter_collection.csv' | head -1)" "$OUT/${TAG}_pmc_w.csv"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 \
    -d /tmp/prof_s -o s --output-format csv -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-sensitivity > /tmp/s.log 2>&1
python3 "$R/tools/pmc_summary.py" "$(find /tmp/prof_s -name '*counter_collection.csv' | head -1)" "$OUT/${TAG}_pmc_s.csv"
# MFMA-busy cycles in their own pass (SQ_VALU_MFMA_BUSY_CYCLES counts cycles, = 32 x N_mfma for the 32x32x16 instructions)
rm -rf /tmp/prof_m
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/prof_m -o m --output-format csv -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-sensitivity > /tmp/m.log 2>&1
python3 "$R/tools/pmc_summary.py" "$(find /tmp/prof_m -name '*counter_collection.csv' | head -1)" "$OUT/${TAG}_pmc_m.csv"
# the bench line comes LAST and reads the counter passes just taken (roofline.traffic is derived from them): shipped defaults again
unset SGC_BWD_STREAMS
mkdir -p "$R/profiles" && cp "$OUT"/${TAG}_pmc_*.csv "$OUT"/${TAG}_kernel_stats.csv "$R/profiles/" 2>/dev/null
cd "$R" && python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/${TAG}_bench.json"
ls -la "$OUT"/${TAG}_*
