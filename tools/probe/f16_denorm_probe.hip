// Does v_mfma_f32_32x32x16_f16 on gfx950 honour fp16 DENORMAL inputs, and do v_cvt_pkrtz_f16_f32 / v_cvt_pk_f16_f32 produce them?
// The two-plane fp16 tier puts max|x| of a tensor at 2^14..2^15; the second plane of an element 2^e holds bits
// e-11 .. e-21, so elements below 2^7 need fp16 denormals (< 2^-14) to keep their 22 bits.  Prints, for plane values 2^-k:
//   the fp16 bit pattern each conversion produces, and  sum_k A[k] * B  from the MFMA against the exact value.
// build: hipcc --offload-arch=gfx950 -O2 f16_denorm_probe.hip -o f16_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(float* out, unsigned* bits) {
    const int lane = threadIdx.x;
    // conversions: x = 1.2345 * 2^-k for k = 10 .. 26
    if (lane < 17) {
        const float x = ldexpf(1.2345f, -(10 + lane));
        const h2 a = __builtin_amdgcn_cvt_pkrtz(x, -x);
        const f16x2 b = __builtin_convertvector((f32x2){x, -x}, f16x2);   // v_cvt_pk_f16_f32 on gfx950 (round to nearest even)
        bits[lane * 2 + 0] = __builtin_bit_cast(unsigned, a);
        bits[lane * 2 + 1] = __builtin_bit_cast(unsigned, b);
    }
    // MFMA: A[row i][k] = 2^-(12 + i % 14) for k == 0 else 0 ; B[k][col j] = 1024 for k == 0: D[i][j] = 1024 * 2^-(12 + i % 14)
    for (int t = 0; t < 2; ++t) {
        f16x8 av, bv;
        const int i = lane & 31, h = lane >> 5;
        for (int e = 0; e < 8; ++e) { av[e] = (_Float16)0.f; bv[e] = (_Float16)0.f; }
        if (h == 0) {
            const unsigned short ab = t == 0 ? (unsigned short)(1u << (9 - (i % 10)))            // denormals 2^-15 .. 2^-24
                                             : (unsigned short)((15 - 2 - (i % 10)) << 10);      // normals 2^-2 .. 2^-11
            av[0] = __builtin_bit_cast(_Float16, ab);
            bv[0] = (_Float16)1024.f;
        }
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
        // D[row][col = i]: rows r -> 8 * (r / 4) + 4 * h + (r % 4); keep column 0's rows
        if (i == 0)
            for (int r = 0; r < 16; ++r) out[t * 32 + 8 * (r / 4) + 4 * h + (r % 4)] = acc[r];
    }
}

int main() {
    float* out; unsigned* bits;
    hipMalloc(&out, 64 * 4); hipMalloc(&bits, 64 * 4);
    hipMemset(out, 0, 256); hipMemset(bits, 0, 256);
    probe<<<1, 64>>>(out, bits);
    float ho[64]; unsigned hb[64];
    hipMemcpy(ho, out, 256, hipMemcpyDeviceToHost); hipMemcpy(hb, bits, 256, hipMemcpyDeviceToHost);
    printf("conversions of +-1.2345*2^-k (fp16 normals end at 2^-14, denormals at 2^-24):\n");
    for (int k = 0; k < 17; ++k)
        printf("  k=%2d  cvt_pkrtz lo/hi %04x %04x   cvt_pk_f16_f32(rne) lo/hi %04x %04x\n", 10 + k, hb[2 * k] & 0xffff, hb[2 * k] >> 16,
               hb[2 * k + 1] & 0xffff, hb[2 * k + 1] >> 16);
    int den_ok = 1, nor_ok = 1;
    for (int i = 0; i < 32; ++i) {
        const float want_d = 1024.f * ldexpf(1.f, -(15 + (i % 10))), want_n = 1024.f * ldexpf(1.f, -(2 + (i % 10)));
        if (ho[i] != want_d) den_ok = 0;
        if (ho[32 + i] != want_n) nor_ok = 0;
    }
    printf("mfma_f32_32x32x16_f16 with NORMAL fp16 A operands: %s\n", nor_ok ? "exact" : "WRONG (probe bug?)");
    printf("mfma_f32_32x32x16_f16 with DENORMAL fp16 A operands: %s  (row 0: got %g want %g; row 9: got %g want %g)\n",
           den_ok ? "exact -- denormal inputs are honoured" : "NOT exact -- denormal inputs are flushed", ho[0], 1024.f * ldexpf(1.f, -15),
           ho[9], 1024.f * ldexpf(1.f, -24));
    return 0;
}
