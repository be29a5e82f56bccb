// What each kind of activity costs in board power on MI355X (the conv kernels are power-limited).
// usage: ./power_model_probe <mode> <seconds>   and poll `rocm-smi --showpower --showclocks` meanwhile.
//   0 resident waves that only s_sleep      1 dense VALU fp32 FMA on random data     2 ds_read_b128 stream (conflict-free)
//   3 HBM read stream (float4, 4 GB buffer)  4 ds_write_b64 + ds_read_b128 + VALU bf16 split of random data (the staging mix)
// build: hipcc --offload-arch=gfx950 -O3 power_model_probe.hip -o power_model_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 2) k_sleep(float* out, int iters) {
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(64);
    if (iters < 0) out[threadIdx.x] = 1.f;
}

__global__ void __launch_bounds__(256, 2) k_valu(float* out, int iters) {
    float a[8];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x;
    for (int j = 0; j < 8; ++j) { h = h * 1664525u + 1013904223u; a[j] = __uint_as_float((h & 0x007fffffu) | 0x3f800000u) - 1.5f; }
    const float m = a[0] * 0.999f, c = a[1] * 0.001f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = fmaf(a[j], m, c + a[(j + 1) & 7] * 1e-3f);
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += a[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256, 2) k_lds_read(float* out, int iters) {
    __shared__ u32x4 buf[4096];   // 64 KB
    for (int i = threadIdx.x; i < 4096; i += 256) {
        unsigned h = (i + blockIdx.x) * 2654435761u;
        buf[i] = u32x4{h, h * 3u, h * 5u, h * 7u};
    }
    __syncthreads();
    u32x4 acc = {0, 0, 0, 0};
    int idx = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc ^= buf[(idx + r * 256) & 4095];
        }
        idx = (idx + 64) & 4095;
    }
    out[blockIdx.x * 256 + threadIdx.x] = __uint_as_float(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
}

__global__ void __launch_bounds__(256, 2) k_hbm_read(const f32x4* __restrict__ src, float* out, long long n4, int iters) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) acc += src[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

// the conv kernel's staging mix on LDS-resident random data: split fp32 -> 3 bf16 planes, ds_write_b64 x 3, ds_read_b128 x 3
__global__ void __launch_bounds__(256, 2) k_staging(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * 256 * 32];
    f32x4 v;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x;
    for (int j = 0; j < 4; ++j) { h = h * 1664525u + 1013904223u; v[j] = __uint_as_float((h & 0x007fffffu) | 0x3f000000u); }
    u32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        unsigned x0[4], x1[4], x2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = v[e];
            x0[e] = __float_as_uint(x) & 0xffff0000u;
            const float r1 = x - __uint_as_float(x0[e]);
            x1[e] = __float_as_uint(r1) & 0xffff0000u;
            const float r2 = r1 - __uint_as_float(x1[e]);
            x2[e] = __float_as_uint(r2);
            v[e] = x * 1.0001f + 1e-3f;
        }
        unsigned char* dst = lds + threadIdx.x * 8 + ((i & 3) * 2048);
        *reinterpret_cast<uint2*>(dst) = uint2{(x0[0] >> 16) | x0[1], (x0[2] >> 16) | x0[3]};
        *reinterpret_cast<uint2*>(dst + 8192) = uint2{(x1[0] >> 16) | x1[1], (x1[2] >> 16) | x1[3]};
        *reinterpret_cast<uint2*>(dst + 16384) = uint2{(x2[0] >> 16) | (x2[1] & 0xffff0000u), (x2[2] >> 16) | (x2[3] & 0xffff0000u)};
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) acc ^= *reinterpret_cast<const u32x4*>(lds + pl * 8192 + ((threadIdx.x * 16 + i * 4096) & 8191));
    }
    out[blockIdx.x * 256 + threadIdx.x] = __uint_as_float(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 5.0;
    float* out; hipMalloc(&out, 512 * 256 * 4);
    f32x4* src = nullptr;
    const long long n4 = (4LL << 30) / 16;
    if (mode == 3) { hipMalloc(&src, n4 * 16); hipMemset(src, 0x3c, n4 * 16); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double total = 0.0; int n = 0;
    while (total < secs * 1e3) {
        hipEventRecord(e0);
        switch (mode) {
        case 0: hipLaunchKernelGGL(k_sleep, dim3(512), dim3(256), 0, 0, out, 20000); break;
        case 1: hipLaunchKernelGGL(k_valu, dim3(512), dim3(256), 0, 0, out, 20000); break;
        case 2: hipLaunchKernelGGL(k_lds_read, dim3(512), dim3(256), 0, 0, out, 20000); break;
        case 3: hipLaunchKernelGGL(k_hbm_read, dim3(4096), dim3(256), 0, 0, src, out, n4, 4); break;
        default: hipLaunchKernelGGL(k_staging, dim3(512), dim3(256), 0, 0, out, 50000); break;
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        total += ms; ++n;
    }
    const double per = total / n;
    const char* names[] = {"s_sleep", "VALU fma", "ds_read_b128", "HBM read", "staging mix"};
    printf("mode %d (%s): %.3f ms per launch", mode, names[mode], per);
    if (mode == 1) printf(", %.1f TFLOP/s fp32", 512.0 * 256 * 20000 * 64 * 2 / (per * 1e-3) / 1e12);
    if (mode == 2) printf(", %.1f TB/s LDS", 512.0 * 256 * 20000 * 16 * 16 / (per * 1e-3) / 1e12);
    if (mode == 3) printf(", %.2f TB/s HBM", 4.0 * n4 * 16 / (per * 1e-3) / 1e12);
    if (mode == 4) printf(", %.2f G elements/s split+written+read", 512.0 * 256 * 50000 * 4 / (per * 1e-3) / 1e9);
    printf("\n");
    return 0;
}
