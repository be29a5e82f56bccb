// Compile-only probe: one instantiation of conv_split_kernel (register / spill / ISA inspection without building the whole library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I radar-camera-fusion-depth_amd/csrc -S --cuda-device-only tools/probe/split_probe.hip -o /tmp/split.s
#define RCF_CONV_KERNELS_ONLY 1
#include "rcf_conv_impl.h"
#ifndef PROBE_CFG
#define PROBE_CFG SplitCfg<3, 2, 16, 0, 2, 1>
#endif
#ifndef PROBE_EPI
#define PROBE_EPI false
#endif
#ifndef PROBE_BST
#define PROBE_BST false
#endif
namespace {
template __global__ void conv_split_kernel<PROBE_CFG, PROBE_EPI, StF32, StF32, PROBE_BST>(ConvArgs);
}
