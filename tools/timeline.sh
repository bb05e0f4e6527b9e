R=$(pwd); cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_prof
rocprofv3 --kernel-trace --stats -d /tmp/tl_prof -o tl -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-sensitivity > /tmp/tl.log 2>&1
python3 "$R/tools/rocpd_timeline.py" "$(find /tmp/tl_prof -name '*.db' | head -1)" "$R/gpurun_out/r03_final5_timeline_2stream.csv"
head -1 "$R/gpurun_out/r03_final5_timeline_2stream.csv"
