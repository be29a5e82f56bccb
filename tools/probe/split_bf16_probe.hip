// Probe (GPU box): C[32x32] = A[32xK] * B[Kx32] three ways -- exact f32 MFMA, bf16x3-split (6 products) on the bf16 MFMA,
// and bf16x3-split with all 9 products -- against an fp64 host reference.  Validates the 32x32x16 bf16 operand layout
// (A: lane l holds row l&31, k = 8*(l>>5)..+7; B: column l&31, same k) and the achievable accuracy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned short f2bf(float x) {   // round to nearest even
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ inline float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

__global__ void probe(const float* A, const float* B, float* Cf32, float* C6, float* C9, int K) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc = {0}, a6 = {0}, a9 = {0};
    for (int k0 = 0; k0 < K; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k0 + h], B[(k0 + h) * 32 + i], acc, 0, 0, 0);
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 ap[3], bp[3];
        for (int e = 0; e < 8; ++e) {
            float a = A[i * K + k0 + 8 * h + e], b = B[(k0 + 8 * h + e) * 32 + i];
            unsigned short s;
            s = f2bf(a); ap[0][e] = __builtin_bit_cast(__bf16, s); a -= bf2f(s);
            s = f2bf(a); ap[1][e] = __builtin_bit_cast(__bf16, s); a -= bf2f(s);
            s = f2bf(a); ap[2][e] = __builtin_bit_cast(__bf16, s);
            s = f2bf(b); bp[0][e] = __builtin_bit_cast(__bf16, s); b -= bf2f(s);
            s = f2bf(b); bp[1][e] = __builtin_bit_cast(__bf16, s); b -= bf2f(s);
            s = f2bf(b); bp[2][e] = __builtin_bit_cast(__bf16, s);
        }
        // small terms first
        for (int x = 2; x >= 0; --x)
            for (int y = 2; y >= 0; --y) {
                a9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[x], bp[y], a9, 0, 0, 0);
                if (x + y <= 2) a6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[x], bp[y], a6, 0, 0, 0);
            }
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        Cf32[row * 32 + i] = acc[r]; C6[row * 32 + i] = a6[r]; C9[row * 32 + i] = a9[r];
    }
}

int main() {
    for (int K : {144, 576, 2304}) {
        std::vector<float> A(32 * K), B(K * 32);
        srand(K);
        for (auto& v : A) v = (float)rand() / RAND_MAX * 2 - 1;
        for (auto& v : B) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.1f + 0.03f;
        float *dA, *dB, *d1, *d2, *d3;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&d1, 4096); hipMalloc(&d2, 4096); hipMalloc(&d3, 4096);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, d1, d2, d3, K);
        std::vector<float> c1(1024), c2(1024), c3(1024);
        hipMemcpy(c1.data(), d1, 4096, hipMemcpyDeviceToHost); hipMemcpy(c2.data(), d2, 4096, hipMemcpyDeviceToHost); hipMemcpy(c3.data(), d3, 4096, hipMemcpyDeviceToHost);
        double e1 = 0, e2 = 0, e3 = 0, mx = 0, sabs = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0, sa = 0;
            for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[k * 32 + j]; sa += fabs((double)A[i * K + k] * B[k * 32 + j]); }
            mx = fmax(mx, fabs(s)); sabs = fmax(sabs, sa);
            e1 = fmax(e1, fabs(c1[i * 32 + j] - s)); e2 = fmax(e2, fabs(c2[i * 32 + j] - s)); e3 = fmax(e3, fabs(c3[i * 32 + j] - s));
        }
        printf("K=%4d  max|C|=%.3f  sum|ab|=%.2f  max abs err: f32-mfma %.3e   bf16x3 6-product %.3e   9-product %.3e   (rel to sum|ab|: %.2e %.2e %.2e)\n",
               K, mx, sabs, e1, e2, e3, e1 / sabs, e2 / sabs, e3 / sabs);
    }
    return 0;
}
