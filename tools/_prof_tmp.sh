cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SGC_BWD_STREAMS=0
rm -rf /tmp/prof_ks
rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o ks -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > /tmp/ks.log 2>&1
python3 "$R/tools/rocpd_summary.py" "$(find /tmp/prof_ks -name '*.db' | head -1)" "$R/gpurun_out/r02_shared_kernel_stats.csv"
head -45 "$R/gpurun_out/r02_shared_kernel_stats.csv" | cut -c1-150
