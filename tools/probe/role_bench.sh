export RCF_BENCH_PREC=f16x2
for lib in tools/probe/librcf_hip_role1.so tools/probe/librcf_hip_role2.so; do
  echo "== $lib (role 1 = consumers only, role 2 = producers only)"
  for l in blocks2_img blocks3_img "deconv0.conv" "deconv1.conv"; do RCF_HIP_LIB=$lib python tools/conv_bench.py 10 "$l" | tail -1; done
done
