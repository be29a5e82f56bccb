// Compile-only probe: one instantiation of conv_wgrad_split_kernel.
#ifdef PROBE_B16
#define RCF_CONV_B16 1
#endif
#define RCF_CONV_KERNELS_ONLY 1
#include "rcf_conv_impl.h"
#ifndef PROBE_CFG
#define PROBE_CFG WsCfg<2, 2, 3, 8, 2>
#endif
namespace {
template __global__ void conv_wgrad_split_kernel<PROBE_CFG, SAct, SAct>(ConvArgs);
}
