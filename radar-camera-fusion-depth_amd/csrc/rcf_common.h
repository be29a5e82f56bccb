// Shared device helpers for librcf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rcf_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define RCF_LEAKY 0.20f

static inline int rcf_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? RCF_OK : (int)e;
}

__device__ __forceinline__ float rcf_lrelu(float v) { return v > 0.f ? v : RCF_LEAKY * v; }
// derivative of leaky relu expressed on its OUTPUT (slope > 0, so sign(out) == sign(in)); PyTorch's
// leaky_relu_backward uses (x > 0 ? 1 : slope), hence 0 maps to the slope.
__device__ __forceinline__ float rcf_lrelu_grad(float out) { return out > 0.f ? 1.f : RCF_LEAKY; }
__device__ __forceinline__ float rcf_sigmoid(float v) { return 1.f / (1.f + __expf(-v)); }

// MFMA 32x32x2 f32 C/D fragment: value r of lane l sits at row (r&3)+8*(r>>2)+4*(l>>5), column l&31.
__device__ __forceinline__ int rcf_mfma_row(int r, int lane_half) { return (r & 3) + 8 * (r >> 2) + 4 * lane_half; }
