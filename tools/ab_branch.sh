# A/B: the encoder's depth branch on its own stream (1) against the main stream (0); weight gradients on the side stream in both
p() { python -c "import sys, json; r = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(r['value'], r['ms_per_step'], r['config']['loss_check']['ok'])"; }
for rep in 1 2; do
for v in 0 1; do
  echo "== RCF_BRANCH_STREAM=$v fp32 (rep $rep)"; RCF_BRANCH_STREAM=$v python bench.py --steps 40 --warmup 4 --no-cpu-baseline 2>/dev/null | p
  echo "== RCF_BRANCH_STREAM=$v bf16 (rep $rep)"; RCF_BRANCH_STREAM=$v python bench.py --dtype bf16 --steps 40 --warmup 4 --no-cpu-baseline 2>/dev/null | p
done
done
RCF_BRANCH_STREAM=1 timeout 1500 python -m pytest tests/test_hip_model.py tests/test_configs_gpu.py -q -m gpu -x 2>&1 | tail -3
