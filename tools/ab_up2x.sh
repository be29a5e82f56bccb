#!/bin/bash
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --dtype f32 --steps 60 --warmup 3 --no-cpu-baseline --no-side-leg --no-other-configs 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-40s %8.2f samples/s  %7.3f ms/step  loss_ok=%s' % ('$label', r['value'], r['ms_per_step'], r['config'].get('loss_check',{}).get('ok')))"
}
for rep in 1 2; do
  run "prev lib, four launches" RCF_HIP_LIB=tools/probe/librcf_hip_prev.so RCF_UP2X_ONE_LAUNCH=0
  run "tree lib, four launches" RCF_UP2X_ONE_LAUNCH=0
  run "tree lib, merged (default)" RCF_X=1
done
