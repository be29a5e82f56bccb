#!/bin/bash
# Round-6 refresh of profiles/r04_power_bound.txt on the CURRENT kernels: clocks and socket power of the forward and the weight-gradient
# kernels with real and with all-zero operands (same instructions, no toggling in the matrix pipe), fp32 tensors on two fp16 planes and
# bf16 tensors.  GPU box: bash tools/power_refresh.sh > gpurun_out/.../power.txt
for prec in f16x2 bf16; do
  for what in fwd wgrad; do
    for shp in 8,225,400,64 8,900,1600,32; do
      for sc in 1 0; do
        echo "== clock_probe $what $prec shape $shp data scale $sc"
        RCF_BENCH_PREC=$prec RCF_PROBE_SHAPE=$shp RCF_BENCH_DATA_SCALE=$sc python tools/clock_probe.py $what 3 2>&1 | grep -v amdgpu.ids
      done
    done
  done
done
echo "== the weight gradient's previous kernel (RCF_WGRAD_TR=0), fp32 tensors, 64 channels, real data"
RCF_WGRAD_TR=0 RCF_BENCH_PREC=f16x2 RCF_PROBE_SHAPE=8,225,400,64 python tools/clock_probe.py wgrad 3 2>&1 | grep -v amdgpu.ids
