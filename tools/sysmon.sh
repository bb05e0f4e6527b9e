#!/bin/bash
# Host load while a test suite runs (GPU box):   bash tools/sysmon.sh gpurun_out/sysmon.txt &   ... ; kill %1
OUT=${1:-gpurun_out/sysmon.txt}
T0=$(date +%s)
echo "nproc $(nproc)  mem $(free -g | awk '/Mem:/ {print $2}') GiB  cgroup $(cat /sys/fs/cgroup/memory.max 2>/dev/null)" > "$OUT"
while true; do
    echo "$(( $(date +%s) - T0 )) s  load $(cut -d' ' -f1-3 /proc/loadavg)  used $(free -g | awk '/Mem:/ {print $3}') GiB  shm $(df -BG /dev/shm | awk 'NR==2 {print $3}')  procs $(pgrep -c -f oracle_worker)" >> "$OUT"
    sleep 10
done
