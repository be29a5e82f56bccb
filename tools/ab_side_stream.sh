# A/B: weight gradients on the main stream (0) against a side stream that overlaps them with the BatchNorm-backward passes (1)
for rep in 1 2; do
for v in 0 1; do
  echo "== RCF_WGRAD_SIDE_STREAM=$v fp32 training (rep $rep)"
  RCF_WGRAD_SIDE_STREAM=$v python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(r['value'], r['ms_per_step'], r['config']['loss_check']['ok'], r['config']['launch'][:40])"
  echo "== RCF_WGRAD_SIDE_STREAM=$v bf16 training (rep $rep)"
  RCF_WGRAD_SIDE_STREAM=$v python bench.py --dtype bf16 --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(r['value'], r['ms_per_step'], r['config']['loss_check']['ok'])"
done
done
echo "== eager (no graph) fp32"
for v in 0 1; do RCF_WGRAD_SIDE_STREAM=$v python bench.py --graph 0 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(r['value'], r['ms_per_step'])"; done
RCF_WGRAD_SIDE_STREAM=1 timeout 1500 python -m pytest tests/test_hip_model.py tests/test_configs_gpu.py -q -m gpu -x 2>&1 | tail -3
