cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "--objects 20 --images 12" "--objects 36 --images 8"; do
  rm -rf /tmp/prof_s
  rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o ks -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline $cfg > /tmp/s.log 2>&1
  grep '^{"metric"' /tmp/s.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'])"
  python3 "$R/tools/rocpd_summary.py" "$(find /tmp/prof_s -name '*.db' | head -1)" /tmp/small.csv
  python3 - <<'PY'
import csv
rows=[r for r in csv.reader(open('/tmp/small.csv')) if r and r[0].isdigit()]
tot=sum(float(r[1]) for r in rows); calls=sum(int(r[0]) for r in rows)
print('kernel time total ms', round(tot,1), 'calls', calls, 'per step (13 steps):', round(tot/13,2), 'ms', calls//13, 'launches')
for r in rows[:12]: print(r[0], r[1], r[2], r[4][:70])
PY
done
