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

// ---- two fp16 operand planes with a per-tensor power-of-two scale (RCF_PREC_F16X2, include/rcf_hip.h) ------------------------------
// An fp32 operand x of a tensor with max|x| = amax is carried into the matrix pipe as  x * s = p0 + p1  with p0 = fp16(x * s) and
// p1 = fp16(x * s - p0), both round-to-nearest-even: 22-23 significant bits, error <= 2^-23 |x| + 2^-25 / s (the second term is the
// fp16 denormal floor: 2^-39 amax).  s is the power of two that puts amax into [2^14, 2^15) -- exact in fp32, no fp16 overflow
// (max 65504), and gfx950's v_mfma_f32_32x32x16_f16 honours fp16 denormal inputs (tools/probe/f16_denorm_probe.hip).  The products
// a0*b0 + a0*b1 + a1*b0 are accumulated in fp32 and the result is multiplied by inv_a * inv_b (two exact multiplications).
struct RcfScale { float s, inv; };
__host__ __device__ inline RcfScale rcf_scale_of_amax(float amax) {
    union { float f; unsigned u; } c;
    c.f = amax;
    const unsigned b = c.u & 0x7fffffffu;
    RcfScale r = {1.f, 1.f};
    if (b == 0u) return r;                       // an all-zero tensor
    int se = 268 - (int)(b >> 23);               // biased exponent of s: 127 + 14 - (E - 127)
    se = se < 1 ? 1 : (se > 253 ? 253 : se);     // keep s and 1/s normal (tensors beyond 2^127 / below 2^-112 are not rescaled further)
    c.u = (unsigned)se << 23; r.s = c.f;
    c.u = (unsigned)(254 - se) << 23; r.inv = c.f;
    return r;
}
// max|x| of a tensor, accumulated by the kernels that write it: wave maximum by shuffles, block maximum through LDS, then at most ONE
// atomic per block on the bit pattern (non-negative floats order like their bit patterns; the slot is zeroed by the host before the
// producers run) -- skipped when the slot already holds a value at least as large.  Same-address atomics retire one per ~4 ns: one
// per WAVE (8192 per launch) cost the BatchNorm passes ~30 us per launch, measured in round 3 (committing early from inside the stream so
// that the tail's guard skips did not help further: what is left is the block's own reduce + one L2 round trip).  All threads of the block must call it.
__device__ __forceinline__ void rcf_amax_commit(float m, float* amax_slot) {
    __shared__ float rcf_amax_lds[16];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const int nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) rcf_amax_lds[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) m = fmaxf(m, rcf_amax_lds[w]);
        const unsigned mb = __float_as_uint(m);
        unsigned* slot = reinterpret_cast<unsigned*>(amax_slot);
        if (mb > __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMax(slot, mb);
    }
}
// |v| of a finite value, 0 for Inf / NaN: the maximum that sets a tensor's scale is taken over its FINITE elements, so one Inf or NaN
// neither flushes the finite elements' planes to zero (s would be 2^-114) nor is lost -- the element itself still reaches the matrix
// pipe as an fp16 Inf / NaN plane and makes every output of its receptive field non-finite, like an fp32 convolution does
// (tests/test_hip_f16x2.py::test_non_finite_operands_propagate)
__device__ __forceinline__ float rcf_abs_finite(float v) {
    const float a = fabsf(v);
    return a <= 3.402823466e+38f ? a : 0.f;
}
__device__ __forceinline__ float rcf_amax4(float m, f32x4 v) {
    return fmaxf(fmaxf(m, fmaxf(rcf_abs_finite(v[0]), rcf_abs_finite(v[1]))), fmaxf(rcf_abs_finite(v[2]), rcf_abs_finite(v[3])));
}

// ---- buffer addressing (gfx950 raw buffers): descriptor over [base, base + 2 GB), byte offsets; a lane whose offset is 0xffffffff is
// out of range -- its store is dropped, its load (also an LDS-DMA load) returns zeros.  The instruction's soffset (an SGPR) is added
// to the address but takes no part in the range check.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rcf_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void rcf_buffer_to_lds16(__amdgpu_buffer_rsrc_t r, void* lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
typedef unsigned rcf_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 rcf_buffer_load_f32x4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

typedef unsigned rcf_u32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rcf_u32x2v rcf_buffer_load_u32x2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
}

// ---- storage type of the NHWC activation / gradient tensors -------------------------------------------------------------------
// StF32: fp32 tensors (the reference's arithmetic).  StB16: bf16 tensors in HBM (BASELINE.json configs 2-4: bf16 storage and MFMA
// operands, fp32 accumulation, fp32 master weights and BatchNorm statistics).  Kernels are templated on the tag and touch such a
// tensor only through the helpers below, with ELEMENT indices; the pointer type stays `float*` at every interface (for StB16 it
// really addresses 2-byte elements).
struct StF32 { static constexpr bool B16 = false; static constexpr int BYTES = 4; };
struct StB16 { static constexpr bool B16 = true; static constexpr int BYTES = 2; };

typedef unsigned rcf_u32x2 __attribute__((ext_vector_type(2)));

// fp32 -> bf16 bits, round to nearest even (NaN payloads are not preserved; the hot path never produces them)
__device__ __forceinline__ unsigned rcf_f2b(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float rcf_b2f(unsigned h) { return __uint_as_float(h << 16); }
// two at once, packed (lo in bits 0-15): one v_cvt_pk_bf16_f32 on gfx950 instead of 2 x (bfe, add3, shift) + or; the same
// round-to-nearest-even on every finite value
typedef __bf16 rcf_bf16x2 __attribute__((ext_vector_type(2)));
typedef float rcf_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned rcf_f2b2(float lo, float hi) {
    const rcf_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, rcf_bf16x2));
}
// the value a bf16 tensor will hold for x (fp32 tensors: x itself)
template <class S>
__device__ __forceinline__ float rcf_round_st(float x) { return S::B16 ? rcf_b2f(rcf_f2b(x)) : x; }

template <class S>
__device__ __forceinline__ f32x4 rcf_ld4(const float* p, size_t i) {
    if constexpr (S::B16) {
        const rcf_u32x2 w = *reinterpret_cast<const rcf_u32x2*>(reinterpret_cast<const unsigned short*>(p) + i);
        f32x4 v;
        v[0] = __uint_as_float(w[0] << 16);
        v[1] = __uint_as_float(w[0] & 0xffff0000u);
        v[2] = __uint_as_float(w[1] << 16);
        v[3] = __uint_as_float(w[1] & 0xffff0000u);
        return v;
    } else {
        return *reinterpret_cast<const f32x4*>(p + i);
    }
}
template <class S>
__device__ __forceinline__ f32x4 rcf_ld4_nt(const float* p, size_t i) {   // non-temporal: streams larger than the MALL
    if constexpr (S::B16) {
        const rcf_u32x2 w = __builtin_nontemporal_load(reinterpret_cast<const rcf_u32x2*>(reinterpret_cast<const unsigned short*>(p) + i));
        f32x4 v;
        v[0] = __uint_as_float(w[0] << 16);
        v[1] = __uint_as_float(w[0] & 0xffff0000u);
        v[2] = __uint_as_float(w[1] << 16);
        v[3] = __uint_as_float(w[1] & 0xffff0000u);
        return v;
    } else {
        return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + i));
    }
}
template <class S>
__device__ __forceinline__ void rcf_st4(float* p, size_t i, f32x4 v) {
    if constexpr (S::B16) {
        rcf_u32x2 w;
        w[0] = rcf_f2b2(v[0], v[1]);
        w[1] = rcf_f2b2(v[2], v[3]);
        *reinterpret_cast<rcf_u32x2*>(reinterpret_cast<unsigned short*>(p) + i) = w;
    } else {
        *reinterpret_cast<f32x4*>(p + i) = v;
    }
}
template <class S>
__device__ __forceinline__ float rcf_ld1(const float* p, size_t i) {
    if constexpr (S::B16) return rcf_b2f(reinterpret_cast<const unsigned short*>(p)[i]);
    else return p[i];
}
template <class S>
__device__ __forceinline__ void rcf_st1(float* p, size_t i, float v) {
    if constexpr (S::B16) reinterpret_cast<unsigned short*>(p)[i] = (unsigned short)rcf_f2b(v);
    else p[i] = v;
}
// element offset -> pointer (for address arithmetic that has to stay in bytes, e.g. LDS-DMA sources)
template <class S>
__device__ __forceinline__ const float* rcf_at(const float* p, size_t i) {
    return reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(p) + i * S::BYTES);
}
template <class S>
static inline const float* rcf_at_host(const float* p, size_t i) {
    return reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(p) + i * S::BYTES);
}
