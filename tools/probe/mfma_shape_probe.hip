// Under the board's power limit, which bf16 MFMA shape delivers more FLOP/s on random operand bits: v_mfma_f32_32x32x16_bf16
// (32768 FLOP, 16 accumulator registers) or v_mfma_f32_16x16x32_bf16 (16384 FLOP, 4 registers)?  Same operand pools, one wave
// per SIMD x 256 CUs, ~0.3 s per shape.  Also the 32x32x16 shape on fp16 operands (the two-plane fp16 split of DESIGN.md section 6).   build: hipcc --offload-arch=gfx950 -O3 mfma_shape_probe.hip -o mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ void operands(bf16x8 (&av)[4], bf16x8 (&bv)[4], bool random) {
    for (int p = 0; p < 4; ++p) {
        unsigned u[4], w[4];
        for (int d = 0; d < 4; ++d) {
            unsigned h = (threadIdx.x * 2654435761u) ^ ((p * 4 + d + 1) * 0x9E3779B9u);
            h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
            u[d] = random ? ((h & 0x807f807fu) | 0x3b803b80u) : 0x3f803f80u;
            w[d] = random ? (((h * 31u) & 0x807f807fu) | 0x3b803b80u) : 0x3f803f80u;
        }
        av[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(u));
        bv[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(w));
    }
}

template <bool RANDOM>
__global__ void __launch_bounds__(256, 1) k32(float* out, int iters) {
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf16x8 av[4], bv[4];
    operands(av, bv, RANDOM);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j % 8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[(j / 2) % 4], bv[(j % 2) + 2 * ((j / 8) % 2)], acc[j % 8], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool RANDOM>
__global__ void __launch_bounds__(256, 1) k16(float* out, int iters) {
    f32x4 acc[16];
    for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    bf16x8 av[4], bv[4];
    operands(av, bv, RANDOM);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j % 16] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[(j / 2) % 4], bv[(j % 2) + 2 * ((j / 8) % 2)], acc[j % 16], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// fp16 operands, same shape: 11 significant bits per operand toggle in the multipliers instead of 8
template <bool RANDOM>
__global__ void __launch_bounds__(256, 1) k32h(float* out, int iters) {
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f16x8 av[4], bv[4];
    for (int p = 0; p < 4; ++p) {
        unsigned u[4], w[4];
        for (int d = 0; d < 4; ++d) {
            unsigned h = (threadIdx.x * 2654435761u) ^ ((p * 4 + d + 1) * 0x9E3779B9u);
            h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
            // fp16 pairs: sign + 10 mantissa bits random, exponent 2^-4 .. 2^-1 (no overflow in 16-deep fp32 sums)
            u[d] = RANDOM ? ((h & 0x83ff83ffu) | 0x2c002c00u) : 0x3c003c00u;
            w[d] = RANDOM ? (((h * 31u) & 0x83ff83ffu) | 0x2c002c00u) : 0x3c003c00u;
        }
        av[p] = __builtin_bit_cast(f16x8, *reinterpret_cast<uint4*>(u));
        bv[p] = __builtin_bit_cast(f16x8, *reinterpret_cast<uint4*>(w));
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j % 8] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[(j / 2) % 4], bv[(j % 2) + 2 * ((j / 8) % 2)], acc[j % 8], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class K>
void run(const char* name, K kern, double flop_per_iter_per_wave, int iters) {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    float last = 0.f, best = 1e9f;
    for (int rep = 0; rep < 40; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&last, e0, e1);
        best = last < best ? last : best;
    }
    const double flop = flop_per_iter_per_wave * iters * 4 * 256;
    printf("%-46s first/best %7.1f TFLOP/s, sustained (40th launch) %7.1f TFLOP/s\n", name, flop / (best * 1e-3) / 1e12, flop / (last * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    run("32x32x16 bf16, constant operands", k32<false>, 16 * 32768.0, 20000);
    run("16x16x32 bf16, constant operands", k16<false>, 32 * 16384.0, 20000);
    run("32x32x16 bf16, random operand bits", k32<true>, 16 * 32768.0, 20000);
    run("16x16x32 bf16, random operand bits", k16<true>, 32 * 16384.0, 20000);
    run("32x32x16 fp16, constant operands", k32h<false>, 16 * 32768.0, 20000);
    run("32x32x16 fp16, random operand bits", k32h<true>, 16 * 32768.0, 20000);
    return 0;
}
