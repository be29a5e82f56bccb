mkdir -p gpurun_out/r4
export RCF_BENCH_PREC=f16x2
for sc in 1 0; do
  echo "== conv_bench f16x2 RCF_BENCH_DATA_SCALE=$sc"
  RCF_BENCH_DATA_SCALE=$sc python tools/conv_bench.py 20 blocks2_img
  RCF_BENCH_DATA_SCALE=$sc python tools/conv_bench.py 20 blocks3_img | tail -1
  RCF_BENCH_DATA_SCALE=$sc python tools/conv_bench.py 10 "deconv0.conv" | tail -1
  RCF_BENCH_DATA_SCALE=$sc python tools/conv_bench.py 10 "deconv1.conv" | tail -1
done
for sc in 1 0; do
  for shp in 8,225,400,64 8,113,200,128 8,900,1600,32; do
    echo "== clock_probe fwd f16x2 shape $shp data scale $sc"
    RCF_PROBE_SHAPE=$shp RCF_BENCH_DATA_SCALE=$sc python tools/clock_probe.py fwd 3
  done
done
echo "== clock_probe wgrad"; RCF_BENCH_DATA_SCALE=1 python tools/clock_probe.py wgrad 3; RCF_BENCH_DATA_SCALE=0 python tools/clock_probe.py wgrad 3
