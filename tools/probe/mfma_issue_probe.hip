// How fast can ONE wave per SIMD issue v_mfma_f32_32x32x16_bf16 compared with two?  (wgrad kernel question: its pure MFMA stream
// measured 45 cycles per MFMA at one wave per SIMD.)  build: hipcc --offload-arch=gfx950 -O3 mfma_issue_probe.hip -o mfma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int THREADS, int FILL>
__global__ void __launch_bounds__(THREADS, THREADS / 256) k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    unsigned u[4] = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(u));
    bf16x8 b = a;
    unsigned f = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < FILL; ++q) f = f * 1664525u + 1013904223u;   // independent VALU fillers between MFMAs
        }
    }
    float s = (float)f;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int NACC, int THREADS, int FILL>
void run(const char* name) {
    float* out; hipMalloc(&out, 256 * THREADS * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, THREADS, FILL>), dim3(256), dim3(THREADS), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, THREADS, FILL>), dim3(256), dim3(THREADS), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * NACC * (THREADS / 64) * 256;   // MFMAs
    const double per_simd = mf / 1024.0;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s  %5.1f ns per MFMA per SIMD\n", name, ms, mf * 32768.0 / (ms * 1e-3) / 1e12, ms * 1e6 / per_simd);
    hipFree(out);
}


// Same loop with the accumulators pinned to AGPRs (as in conv_wgrad_split_kernel) and random operand bits (toggle power).
template <int NACC, int THREADS, bool RANDOM, bool AGPR>
__global__ void __launch_bounds__(THREADS, THREADS / 256) k2(float* out, int iters) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf16x8 av[3], bv[3];
    for (int p = 0; p < 3; ++p) {
        unsigned u[4], w[4];
        for (int d = 0; d < 4; ++d) {
            unsigned h = (threadIdx.x * 2654435761u) ^ ((p * 4 + d + 1) * 0x9E3779B9u);
            h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
            // bf16 pairs with small exponents so nothing overflows: sign/mantissa random, exponent ~ 2^-8..2^0
            u[d] = RANDOM ? ((h & 0x807f807fu) | 0x3b803b80u) : 0x3f803f80u;
            w[d] = RANDOM ? (((h * 31u) & 0x807f807fu) | 0x3b803b80u) : 0x3f803f80u;
        }
        av[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(u));
        bv[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(w));
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(av[j % 3]), "v"(bv[(j / 3) % 3]));
            else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[j % 3], bv[(j / 3) % 3], acc[j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int NACC, int THREADS, bool RANDOM, bool AGPR>
void run2(const char* name) {
    float* out; hipMalloc(&out, 256 * THREADS * 4);
    const int iters = 40000;   // ~6 ms: long enough for the power manager to react
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k2<NACC, THREADS, RANDOM, AGPR>), dim3(256), dim3(THREADS), 0, 0, out, 10);
    hipDeviceSynchronize();
    float best = 1e9f, last = 0.f;
    for (int rep = 0; rep < 20; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k2<NACC, THREADS, RANDOM, AGPR>), dim3(256), dim3(THREADS), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; last = ms;
    }
    const double per_simd = (double)iters * NACC * (THREADS / 64) * 256 / 1024.0;
    printf("%-52s best %6.2f ns, sustained (20th launch) %6.2f ns per MFMA per SIMD\n", name, best * 1e6 / per_simd, last * 1e6 / per_simd);
    hipFree(out);
}

// Does the ORDER of operands matter under the power limit?  8 accumulators, pools of 4 random A and 4 random B operands;
// MODE 0: A and B both change at every MFMA; 1: A changes, B fixed for 4; 2: A fixed for 2 while B alternates (conv_split_kernel's
// order); 3: A and B both fixed for 4 MFMAs.
template <int MODE>
__global__ void __launch_bounds__(256, 1) k3(float* out, int iters) {
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf16x8 av[4], bv[4];
    for (int p = 0; p < 4; ++p) {
        unsigned u[4], w[4];
        for (int d = 0; d < 4; ++d) {
            unsigned h = (threadIdx.x * 2654435761u) ^ ((p * 4 + d + 1) * 0x9E3779B9u);
            h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
            u[d] = (h & 0x807f807fu) | 0x3b803b80u;
            w[d] = ((h * 31u) & 0x807f807fu) | 0x3b803b80u;
        }
        av[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(u));
        bv[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(w));
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int ia, ib;
            if (MODE == 0) { ia = j % 4; ib = (j + j / 4) % 4; }
            else if (MODE == 1) { ia = j % 4; ib = (j / 4) % 4; }
            else if (MODE == 2) { ia = (j / 2) % 4; ib = (j % 2) + 2 * ((j / 8) % 2); }
            else { ia = (j / 4) % 4; ib = (j / 4 + 1) % 4; }
            acc[j % 8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ia], bv[ib], acc[j % 8], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run3(const char* name) {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    const int iters = 25000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k3<MODE>), dim3(256), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    float last = 0.f;
    for (int rep = 0; rep < 15; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k3<MODE>), dim3(256), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&last, e0, e1);
    }
    printf("%-60s sustained %6.2f ns per MFMA per SIMD\n", name, last * 1e6 / ((double)iters * 16));
    hipFree(out);
}

// long mode: ./mfma_issue_probe long <0 constant | 1 random> <seconds>   (poll rocm-smi --showpower --showclocks meanwhile)
template <bool RANDOM>
void run_long(double secs) {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double total = 0.0; int n = 0;
    while (total < secs * 1e3) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k2<9, 256, RANDOM, false>), dim3(256), dim3(256), 0, 0, out, 40000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        total += ms; ++n;
    }
    printf("%s operands: %.2f ns per MFMA per SIMD over %.1f s\n", RANDOM ? "random" : "constant", total / n * 1e6 / (40000.0 * 9), total / 1e3);
}

int main(int argc, char** argv) {
    if (argc >= 4 && argv[1][0] == 'l') {
        if (atoi(argv[2])) run_long<true>(atof(argv[3])); else run_long<false>(atof(argv[3]));
        return 0;
    }
    run<9, 256, 0>("1 wave/SIMD, 9 acc, no fillers");
    run<9, 256, 2>("1 wave/SIMD, 9 acc, 2 VALU fillers");
    run<9, 256, 5>("1 wave/SIMD, 9 acc, 5 VALU fillers");
    run<3, 256, 0>("1 wave/SIMD, 3 acc, no fillers");
    run<4, 512, 0>("2 waves/SIMD, 4 acc, no fillers");
    run<4, 512, 2>("2 waves/SIMD, 4 acc, 2 VALU fillers");
    run<4, 512, 5>("2 waves/SIMD, 4 acc, 5 VALU fillers");
    run2<9, 256, false, false>("1 wave/SIMD, 9 acc VGPR, constant operands");
    run2<9, 256, true, false>("1 wave/SIMD, 9 acc VGPR, random operands");
    run2<9, 256, true, true>("1 wave/SIMD, 9 acc AGPR, random operands");
    run2<3, 256, true, true>("1 wave/SIMD, 3 acc AGPR, random operands");
    run2<4, 512, true, false>("2 waves/SIMD, 4 acc VGPR, random operands");
    run3<0>("operand order: A and B change every MFMA");
    run3<1>("operand order: A changes, B fixed for 4");
    run3<2>("operand order: A fixed for 2, B alternates (kernel's order)");
    run3<3>("operand order: A and B fixed for 4");
    return 0;
}
