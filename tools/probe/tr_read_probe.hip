// ds_read_b64_tr_b16 on gfx950: (1) semantics -- within each 16-lane group lane i supplies the 8-byte address of matrix row i / 4,
// column quad i % 4 of a [4][16] 16-bit matrix and receives column i (elements = rows 0..3); (2) LDS cycles of the wgrad_tr layout
// (pixel pitch 64 B: a 32-lane half reads 256 contiguous bytes) against a 128-B pixel pitch (two pixels per bank row) and a plain
// ds_read_b64 of contiguous addresses.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/tr_read_probe.hip -o tools/probe/tr_read_probe && tools/probe/tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void sem_kernel(const unsigned short* in, unsigned short* out, const int* addr) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = in[i];
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}

template <int MODE>
__global__ void time_kernel(unsigned long long* cyc, unsigned* sink, int pitch) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<unsigned*>(lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int lh = lane >> 5, i16 = lane & 15, g1 = (lane >> 4) & 1;
    int off = (8 * lh + (i16 >> 2)) * pitch + g1 * 32 + (i16 & 3) * 8;
    if (MODE == 2) off = lane * 8;
    const unsigned char* p = lds + off + (threadIdx.x >> 6) * 8192;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            u32x2 v;
            if (MODE == 2) v = *reinterpret_cast<const u32x2*>(p + k * 512);
            else v = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + k * 256)));
            acc += v[0] ^ v[1];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) atomicAdd(cyc, t1 - t0);
    sink[threadIdx.x] = acc;
}

int main() {
    std::vector<unsigned short> in(8192), out(256);
    std::vector<int> addr(64);
    for (int i = 0; i < 8192; ++i) in[i] = (unsigned short)i;
    srand(3);
    for (int l = 0; l < 64; ++l) addr[l] = 4 * (rand() % 2000);   // arbitrary 8-byte aligned addresses (in 16-bit elements)
    unsigned short *din, *dout; int* daddr;
    hipMalloc(&din, 16384); hipMalloc(&dout, 512); hipMalloc(&daddr, 256);
    hipMemcpy(din, in.data(), 16384, hipMemcpyHostToDevice);
    hipMemcpy(daddr, addr.data(), 256, hipMemcpyHostToDevice);
    sem_kernel<<<1, 64>>>(din, dout, daddr);
    hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            const int g = l >> 4, i = l & 15;
            const int srcl = 16 * g + 4 * j + (i >> 2);           // lane that supplied row j's quad holding column i
            const int want = addr[srcl] + (i & 3);
            if (out[l * 4 + j] != want) { if (bad < 8) printf("lane %d elem %d: got %d want %d\n", l, j, out[l * 4 + j], want); ++bad; }
        }
    printf("semantics (lane i of a 16-group supplies row i/4, quad i%%4; receives column i): %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    unsigned long long* dc; unsigned* sink;
    hipMalloc(&dc, 8); hipMalloc(&sink, 1024);
    const char* names[3] = {"tr_b16, pixel pitch  64 B (wgrad_tr layout)", "tr_b16, pixel pitch 128 B", "plain ds_read_b64, contiguous"};
    for (int m = 0; m < 3; ++m) {
        unsigned long long z = 0, c = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemcpy(dc, &z, 8, hipMemcpyHostToDevice);
            if (m == 0) time_kernel<0><<<1, 256>>>(dc, sink, 64);
            else if (m == 1) time_kernel<0><<<1, 256>>>(dc, sink, 128);
            else time_kernel<2><<<1, 256>>>(dc, sink, 64);
            hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        }
        printf("%-46s %.2f s_memtime ticks per wave-instruction (4 waves on one CU)\n", names[m], (double)c / 4.0 / 4096.0);
    }
    return bad ? 1 : 0;
}
