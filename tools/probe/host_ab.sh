for dt in f32 bf16; do python bench.py --dtype $dt --steps 20 --warmup 3 --no-cpu-baseline --no-side-leg --no-other-configs 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$dt', r['value'], r['ms_per_step'], 'host_enqueue_ms', r['config'].get('host_enqueue_ms_per_step'))"; done
