# A/B of the wave-specialised two-plane kernels against conv_split_kernel, layer by layer (same box, same run)
export RCF_BENCH_PREC=f16x2
for ws in 1 0; do
  echo "== conv_bench f16x2 RCF_SPLIT_WS=$ws"
  RCF_SPLIT_WS=$ws python tools/conv_bench.py 10
done
