#!/bin/bash
# Kernel timeline of a few bench steps (GPU box, from the repo root):   bash tools/timeline.sh <tag>   -> gpurun_out/<tag>_timeline.csv
TAG=${1:?usage: timeline.sh <tag>}
R=$(pwd); mkdir -p "$R/gpurun_out"; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_prof
rocprofv3 --kernel-trace --stats -d /tmp/tl_prof -o tl -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-sensitivity > /tmp/tl.log 2>&1
python3 "$R/tools/rocpd_timeline.py" "$(find /tmp/tl_prof -name '*.db' | head -1)" "$R/gpurun_out/${TAG}_timeline.csv"
head -1 "$R/gpurun_out/${TAG}_timeline.csv"
