# A/B: the up-2x forward as one launch over the four output phases against four launches (same box)
for one in 1 0; do
  echo "== RCF_UP2X_ONE_LAUNCH=$one fp32 training"
  RCF_UP2X_ONE_LAUNCH=$one python bench.py --steps 15 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(r['value'], r['ms_per_step'], r['config']['loss_check'])"
  echo "== RCF_UP2X_ONE_LAUNCH=$one bf16 inference"
  RCF_UP2X_ONE_LAUNCH=$one python bench.py --workload infer --steps 15 --warmup 4 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(r['value'], r['ms_per_step'])"
  echo "== RCF_UP2X_ONE_LAUNCH=$one bf16 training"
  RCF_UP2X_ONE_LAUNCH=$one python bench.py --dtype bf16 --steps 15 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(r['value'], r['ms_per_step'], r['config']['loss_check'])"
done
