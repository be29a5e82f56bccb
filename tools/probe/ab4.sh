run() { label=$1; shift
  env "$@" python bench.py --dtype f32 --steps 150 --warmup 3 --no-cpu-baseline --no-side-leg --no-other-configs 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-52s %8.2f samples/s  %7.3f ms/step' % ('$label', r['value'], r['ms_per_step']))"
}
for rep in 1 2 3; do
  run "tree, per-phase launches" RCF_UP2X_ONE_LAUNCH=0 RCF_UP2X_WGRAD_ONE_LAUNCH=0
  run "tree, merged forward, per-phase weight gradients" RCF_UP2X_WGRAD_ONE_LAUNCH=0
  run "tree, defaults (merged forward + one-launch wgrad)" RCF_X=1
done
