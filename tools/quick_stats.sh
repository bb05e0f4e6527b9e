#!/bin/bash
# Warm-only per-kernel statistics of the default bench step on ONE stream (the first part of tools/collect_profiles.sh):
#   bash tools/quick_stats.sh <tag>   ->  gpurun_out/<tag>_kernel_stats.csv
set -u
TAG=${1:-rXX}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export SGC_BWD_STREAMS=0
rm -rf /tmp/prof_ks
rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o ks -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-sensitivity > /tmp/ks.log 2>&1
grep '^{"metric"' /tmp/ks.log | tail -1 > "$OUT/${TAG}_bench_under_rocprof.json"
python3 "$R/tools/rocpd_summary.py" "$(find /tmp/prof_ks -name '*.db' | head -1)" "$OUT/${TAG}_kernel_stats.csv" --skip-steps 3
head -40 "$OUT/${TAG}_kernel_stats.csv" | cut -c1-150
