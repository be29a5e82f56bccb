#!/bin/bash
# Same-box A/B of the bf16 legs (training, hipGraph inference, RadarNet) against another build of the library, alternating.
#   tools/ab_lib_legs.sh tools/probe/librcf_hip_prev.so
OLD=$1
for rep in 1 2; do
  for lib in "$OLD" ""; do
    for leg in "--dtype bf16" "--workload infer" "--workload radarnet"; do
      RCF_HIP_LIB=$lib python bench.py $leg --steps 20 --warmup 3 --no-cpu-baseline --no-side-leg --no-other-configs 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-34s %-22s %9.2f %s  %7.3f ms/step' % ('$lib' or 'tree', '$leg', r['value'], r['unit'], r['ms_per_step']))"
    done
  done
done
