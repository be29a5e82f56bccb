// Compile-only probe: one instantiation of conv_b16_kernel (register / spill / ISA inspection without building the whole library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I radar-camera-fusion-depth_amd/csrc -S --cuda-device-only tools/probe/dma_big_probe.hip -o /tmp/dma_big.s
#define RCF_CONV_B16 1
#define RCF_CONV_KERNELS_ONLY 1
#include "rcf_conv_impl.h"
namespace {
int num_cus() { return 256; }
#include "rcf_conv_b16_dma.h"
#ifndef PROBE_CFG
#define PROBE_CFG DmaCfg<3, 2, 32, 4>
#endif
#ifndef PROBE_EPI
#define PROBE_EPI false
#endif
#ifndef PROBE_BST
#define PROBE_BST false
#endif
template __global__ void conv_b16_kernel<PROBE_CFG, PROBE_EPI, PROBE_BST>(ConvArgs);
}
