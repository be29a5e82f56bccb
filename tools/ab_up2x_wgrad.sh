# A/B: up-2x weight gradient as four 2x2 phase weight gradients (0) against one 3x3 nearest-gather weight gradient (1)
for rep in 1 2; do
for v in 0 1; do
  echo "== RCF_UP2X_WGRAD_DIRECT=$v bf16 training (rep $rep)"
  RCF_UP2X_WGRAD_DIRECT=$v python bench.py --dtype bf16 --steps 15 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(r['value'], r['ms_per_step'], r['config']['loss_check']['ok'])"
  echo "== RCF_UP2X_WGRAD_DIRECT=$v fp32 training (rep $rep)"
  RCF_UP2X_WGRAD_DIRECT=$v python bench.py --steps 15 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(r['value'], r['ms_per_step'], r['config']['loss_check']['ok'])"
done
done
RCF_UP2X_WGRAD_DIRECT=1 timeout 900 python -m pytest tests/test_hip_model.py -q -m gpu -x -k "t1 or t0 or tiny or published" 2>&1 | tail -2
