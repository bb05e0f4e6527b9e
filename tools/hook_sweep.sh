#!/bin/bash
# Alternated A/B of the library's kernel-variant hooks inside ONE box (box-to-box spread is larger than most of these effects).
# Needs a library built with SGC_EXPERIMENTS=1 (the product library reads no environment variable, csrc/common.h:SgcTuning):
#   SGC_EXPERIMENTS=1 python -m scene_graph_commonsense_amd.build --force
#   bash tools/hook_sweep.sh        -> one line per run: hook setting, ms/step, per-kernel ms
run() {
    env SGC_EXPERIMENTS=1 "$@" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-sensitivity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('%-28s %7.2f ms  conv3 f/d/w %.2f %.2f %.2f  fc1 f/d/w %.2f %.2f %.2f  expand %.2f unpool %.2f contract %.2f' % ('$*', d['ms_per_step'], k['conv3_fwd'], k['conv3_dgrad'], k['conv3_wgrad'], k['fc1_fwd'], k['fc1_dgrad'], k['fc1_wgrad'], k['expand_dense'], k['unpool'], k['contract']))"
}
for rep in 1 2; do
    run SGC_NOOP=1
    run SGC_HALO_WALK=0
    run SGC_NT_ALIGNED=0
    run SGC_NT_ALIGNED=1
    run SGC_TN_XCD=0
    run SGC_TN_PATCH=0
    run SGC_BWD_STREAMS=0
done
