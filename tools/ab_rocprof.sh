#!/bin/bash
# Per-kernel warm statistics of the step under two settings of one environment switch (single-stream backward):
#   bash tools/ab_rocprof.sh VAR A B   ->  gpurun_out/ab_<VAR>_<value>.csv
VAR=$1; R=$(pwd); OUT=$R/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp SGC_BWD_STREAMS=0
for v in $2 $3; do
    export $VAR=$v
    rm -rf /tmp/ab_prof
    rocprofv3 --kernel-trace --stats -d /tmp/ab_prof -o ab -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-sensitivity > /tmp/ab.log 2>&1
    python3 "$R/tools/rocpd_summary.py" "$(find /tmp/ab_prof -name '*.db' | head -1)" "$OUT/ab_${VAR}_$v.csv" --skip-steps 3
    echo "== $VAR=$v"; head -16 "$OUT/ab_${VAR}_$v.csv" | cut -c1-150; tail -1 "$OUT/ab_${VAR}_$v.csv"
done
