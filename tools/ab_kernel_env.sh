#!/bin/bash
# Alternated A/B of one environment switch, reporting single-stream per-kernel HIP-event times (bench.py's kernels_ms_single_stream):
#   bash tools/ab_kernel_env.sh VAR A B reps key1 key2 ...
VAR=$1; A=$2; B=$3; REPS=$4; shift 4
KEYS="$*"
run() {
    env "$VAR=$1" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-sensitivity 2>/dev/null | KEYS="$KEYS" python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); k=d['kernels_ms_single_stream'] or {}
print('%-22s %7.2f ms/step  ' % ('$VAR=$1', d['ms_per_step']) + '  '.join('%s %.3f' % (n, k.get(n, float('nan'))) for n in os.environ['KEYS'].split()))"
}
for rep in $(seq $REPS); do run "$A"; run "$B"; done
