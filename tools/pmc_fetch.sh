#!/bin/bash
# One FETCH_SIZE counter pass over one bench step (GPU box, from the repo root):   bash tools/pmc_fetch.sh <tag>   -> gpurun_out/<tag>_pmc_f.csv
TAG=${1:?usage: pmc_fetch.sh <tag>}
R=$(pwd); mkdir -p "$R/gpurun_out"; cd /tmp && export TMPDIR=/tmp SGC_BWD_STREAMS=0
rm -rf /tmp/prof_f
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -o f --output-format csv -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-sensitivity > /tmp/f.log 2>&1
python3 "$R/tools/pmc_summary.py" "$(find /tmp/prof_f -name '*counter_collection.csv' | head -1)" "$R/gpurun_out/${TAG}_pmc_f.csv"
grep -E "gemm_nt_pp_kernel<1, 0, 0, 0, 1>|gemm_tn_pp_kernel<1, 3, 0>|gemm_nt_pp_kernel<0, 3, 0, 1, 0>" "$R/gpurun_out/${TAG}_pmc_f.csv"
