#!/bin/bash
# Same-box A/B of the training step around the up-2x convolutions: the previous library with four launches per forward, the tree's
# library with the per-phase forms, and the tree's defaults (forward: four phases from one staged tile; weight gradient: four phases
# in one launch).   tools/ab_up2x.sh [dtype] [prev-lib]
DT=${1:-f32}; PREV=${2:-tools/probe/librcf_hip_prev.so}
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --dtype $DT --steps 60 --warmup 3 --no-cpu-baseline --no-side-leg --no-other-configs 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-52s %8.2f samples/s  %7.3f ms/step  loss_ok=%s' % ('$label', r['value'], r['ms_per_step'], r['config'].get('loss_check',{}).get('ok')))"
}
for rep in 1 2; do
  [ -f "$PREV" ] && run "previous library, per-phase launches" RCF_HIP_LIB=$PREV RCF_UP2X_ONE_LAUNCH=0 RCF_UP2X_WGRAD_ONE_LAUNCH=0
  run "tree, per-phase launches" RCF_UP2X_ONE_LAUNCH=0 RCF_UP2X_WGRAD_ONE_LAUNCH=0
  run "tree, merged forward, per-phase weight gradients" RCF_UP2X_WGRAD_ONE_LAUNCH=0
  run "tree, defaults (merged forward + one-launch wgrad)" RCF_X=1
done
