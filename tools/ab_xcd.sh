# A/B: tiles dealt to the workgroups round-robin (0) against eight contiguous bands, one per XCD (1) -- same box, alternating
for rep in 1 2; do
for b in 0 1; do
  echo "== RCF_XCD_BANDS=$b fp32 training (rep $rep)"
  RCF_XCD_BANDS=$b python bench.py --steps 15 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(r['value'], r['ms_per_step'], r['config']['loss_check'], r['roofline'].get('avg_launch_ms'))"
  echo "== RCF_XCD_BANDS=$b bf16 inference (rep $rep)"
  RCF_XCD_BANDS=$b python bench.py --workload infer --steps 15 --warmup 4 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(r['value'], r['ms_per_step'])"
  echo "== RCF_XCD_BANDS=$b bf16 training (rep $rep)"
  RCF_XCD_BANDS=$b python bench.py --dtype bf16 --steps 15 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(r['value'], r['ms_per_step'], r['config']['loss_check'])"
done
done
echo "== parity under RCF_XCD_BANDS=1"
RCF_XCD_BANDS=1 timeout 1500 python -m pytest tests/test_hip_f16x2.py tests/test_hip_bf16.py tests/test_hip_ops.py -q -m gpu -x 2>&1 | tail -3
