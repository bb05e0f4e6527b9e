#!/bin/bash
# Vector-instruction counts per kernel of one bench step (GPU box, repo root): SQ_INSTS_VALU and friends in their own --pmc pass,
# summarised as "ms the launch would take with the vector pipes 100 % busy" beside its duration (tools/valu_counters.py).
#   bash tools/valu_counters.sh  ->  gpurun_out/valu_counters.csv, gpurun_out/valu_counters.txt
R=$(pwd); cd /tmp && export TMPDIR=/tmp
export SGC_BWD_STREAMS=0
rm -rf /tmp/prof_x
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/prof_x -o x --output-format csv -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-sensitivity > /tmp/x.log 2>&1
python3 "$R/tools/pmc_summary.py" "$(find /tmp/prof_x -name '*counter_collection.csv' | head -1)" "$R/gpurun_out/valu_counters.csv"
python3 "$R/tools/valu_counters.py" "$R/gpurun_out/valu_counters.csv" > "$R/gpurun_out/valu_counters.txt"
head -40 "$R/gpurun_out/valu_counters.txt"
