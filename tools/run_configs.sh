#!/bin/bash
# The bench at the other BASELINE.json / DESIGN configurations, one line each (GPU box, from the repo root):
#   bash tools/run_configs.sh > profiles/rNN_configs.txt
run() {
    label=$1; shift
    env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-sensitivity ${ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-62s %8.2f ms/step  %9.0f pairs/s  frac %.3f  step_frac %s  peak %.1f GB' % ('$label', d['ms_per_step'], d['value'], (d['roofline'] or {}).get('frac', 0) if d['roofline'] else 0, (d['roofline'] or {}).get('step_frac'), d['peak_memory_gb']))"
}
ARGS="" run "8x64 default (two-stream backward, GEMMs apart)" SGC_NOOP=1
ARGS="" run "8x64 SGC_TUNING=gemms_apart=0 (round-2 order)" SGC_TUNING=gemms_apart=0
ARGS="" run "8x64 SGC_BWD_STREAMS=0 (one stream)" SGC_BWD_STREAMS=0
ARGS="" run "8x64 default again" SGC_NOOP=1
ARGS="" run "8x64 round-6 switches off: fc1_x16=0" SGC_TUNING=fc1_x16=0
ARGS="" run "8x64 round-6 switches off: wgrad_xcd_k=0 (one M tile per XCD for every K range)" SGC_TUNING=wgrad_xcd_k=0
ARGS="" run "8x64 round-6 switches off: gather_wgrad=0 (patch copy for the whole list)" SGC_TUNING=gather_wgrad=0
ARGS="" run "8x64 all three off (the step of round 5)" SGC_TUNING=fc1_x16=0,wgrad_xcd_k=0,gather_wgrad=0
ARGS="" run "8x64 default, third time" SGC_NOOP=1
ARGS="" run "8x64 column forms of the conv3 window backward (rounds 1-2)" SGC_TUNING=patch_dgrad=0,patch_wgrad=0
ARGS="" run "8x64 SGC_TUNING=shared_linear=0 (every pair convolves its own X windows)" SGC_TUNING=shared_linear=0
ARGS="" run "8x64 SGC_SHARED_LEVEL=2 (no second level)" SGC_SHARED_LEVEL=2
ARGS="" run "8x64 SGC_SHARED_LEVEL=1 (fc1 per pair)" SGC_SHARED_LEVEL=1
ARGS="" run "8x64 SGC_SHARED_LEVEL=0 (everything per pair)" SGC_SHARED_LEVEL=0
ARGS="--forward-only" run "8x64 forward only" SGC_NOOP=1
ARGS="--objects 36" run "8x36 (configs[1])" SGC_NOOP=1
ARGS="--objects 20 --images 12" run "12x20 (the largest case of the reference: N <= 20, batch 12)" SGC_NOOP=1
ARGS="--objects 20 --images 10" run "10x20 (configs[0] size)" SGC_NOOP=1
ARGS="--objects 100 --images 4 --dataset oiv6" run "OpenImages 4x100 (configs[4])" SGC_NOOP=1
ARGS="--images 16" run "16x64" SGC_NOOP=1
