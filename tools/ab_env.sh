#!/bin/bash
# Alternated A/B of ONE environment switch inside one box (box-to-box spread exceeds most effects):
#   bash tools/ab_env.sh VAR A B [reps] [extra bench.py arguments...]
# C-library switches (csrc/common.h:SgcTuning) need a library built with SGC_EXPERIMENTS=1; engine.TUNING switches work on the product library.
VAR=$1; A=$2; B=$3; REPS=${4:-2}; shift $(( $# < 4 ? $# : 4 ))
run() {
    env "$VAR=$1" python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-sensitivity "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('%-24s %7.2f ms/step  %8.0f pairs/s  ' % ('$VAR=$1', d['ms_per_step'], d['value']) + '  '.join('%s %.2f' % (n, k[n]) for n in ('conv3_fwd_windows', 'fc1_fwd_windows', 'fc1_fwd_assemble', 'expand_dense') if n in k))"
}
for rep in $(seq $REPS); do run "$A" "$@"; run "$B" "$@"; done
