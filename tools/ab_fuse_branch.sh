p() { python -c "import sys, json; r = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(r['value'], r['ms_per_step'], (r.get('config', {}).get('loss_check') or {}).get('ok'))"; }
for rep in 1 2; do
for v in 0 1; do
  echo "== RCF_FUSE_ON_BRANCH=$v fp32 (rep $rep)"; RCF_FUSE_ON_BRANCH=$v python bench.py --steps 40 --warmup 4 --no-cpu-baseline 2>/dev/null | p
  echo "== RCF_FUSE_ON_BRANCH=$v bf16 (rep $rep)"; RCF_FUSE_ON_BRANCH=$v python bench.py --dtype bf16 --steps 40 --warmup 4 --no-cpu-baseline 2>/dev/null | p
done
done
echo "== infer"; for v in 0 1; do RCF_FUSE_ON_BRANCH=$v python bench.py --workload infer --steps 20 --warmup 5 2>/dev/null | p; done
RCF_FUSE_ON_BRANCH=1 timeout 1500 python -m pytest tests/test_hip_model.py tests/test_configs_gpu.py tests/test_hip_bf16.py -q -m gpu -x 2>&1 | tail -3
