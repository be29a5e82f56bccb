#!/bin/bash
# Same-box A/B of the training step over ONE environment switch of the library / engine, alternating (two rounds).
#   tools/ab_env.sh RCF_WGRAD_TR 0 1 [dtype] [steps]
VAR=$1; A=$2; B=$3; DT=${4:-f32}; STEPS=${5:-40}
for rep in 1 2; do
  for v in "$A" "$B"; do
    env $VAR=$v python bench.py --dtype $DT --steps $STEPS --warmup 3 --no-cpu-baseline --no-side-leg --no-other-configs 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-18s %s %8.2f samples/s  %7.3f ms/step  loss_ok=%s' % ('$VAR=$v', '$DT', r['value'], r['ms_per_step'], r['config'].get('loss_check',{}).get('ok')))"
  done
done
