DT=${1:-f32}
run() { label=$1; shift
  env "$@" python bench.py --dtype $DT --steps 150 --warmup 3 --no-cpu-baseline --no-side-leg --no-other-configs 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-52s %8.2f samples/s  %7.3f ms/step  loss_ok=%s' % ('$label', r['value'], r['ms_per_step'], r['config'].get('loss_check',{}).get('ok')))"
}
for rep in 1 2 3; do
  run "$DT previous library (four reduce launches per phase set)" RCF_HIP_LIB=tools/probe/librcf_hip_prev.so
  run "$DT tree (one reduce launch per phase set)" RCF_X=1
done
