// How fast can ONE wave per SIMD issue v_mfma_f32_32x32x16_bf16 compared with two?  (wgrad kernel question: its pure MFMA stream
// measured 45 cycles per MFMA at one wave per SIMD.)  build: hipcc --offload-arch=gfx950 -O3 mfma_issue_probe.hip -o mfma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
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

int main() {
    run<9, 256, 0>("1 wave/SIMD, 9 acc, no fillers");
    run<9, 256, 2>("1 wave/SIMD, 9 acc, 2 VALU fillers");
    run<9, 256, 5>("1 wave/SIMD, 9 acc, 5 VALU fillers");
    run<3, 256, 0>("1 wave/SIMD, 3 acc, no fillers");
    run<4, 512, 0>("2 waves/SIMD, 4 acc, no fillers");
    run<4, 512, 2>("2 waves/SIMD, 4 acc, 2 VALU fillers");
    run<4, 512, 5>("2 waves/SIMD, 4 acc, 5 VALU fillers");
    return 0;
}
