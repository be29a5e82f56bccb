#!/bin/bash
# Same-box A/B of the training step: the library of another commit (RCF_HIP_LIB) against the tree's, alternating.
#   tools/ab_lib.sh tools/probe/librcf_hip_r4.so [dtype] [steps]
OLD=$1; DT=${2:-f32}; STEPS=${3:-60}
for rep in 1 2; do
  for lib in "$OLD" ""; do
    RCF_HIP_LIB=$lib python bench.py --dtype $DT --steps $STEPS --warmup 3 --no-cpu-baseline --no-side-leg --no-other-configs 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-34s %s %8.2f samples/s  %7.3f ms/step  loss_ok=%s' % ('$lib' or 'tree', '$DT', r['value'], r['ms_per_step'], r['config'].get('loss_check',{}).get('ok')))"
  done
done
