// Standalone timing harness for conv_split_ws_kernel (GPU box; compiles in seconds because it instantiates ONE configuration):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I radar-camera-fusion-depth_amd/csrc [-DRCF_WS_PROBE_ROLE=1|2] [-DRCF_WS_VARIANT=..] \
//         tools/probe/ws_harness.hip -o tools/probe/ws_harness_<tag>
//   ./ws_harness_<tag> [n h w c]        (3x3 stride-1 c -> c forward, 64-co workgroups, random fp32 input, random fp16 weight planes)
// Times the kernel alone (hipEvents, 20 launches) -- wrong results are fine here: this is for A/B of schedule variants, the parity
// tests run on the library build.
#define RCF_CONV_KERNELS_ONLY 1
#include "rcf_conv_impl.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {
#ifndef WSH_NT
#define WSH_NT 2
#endif
using CP = SplitCfg<3, WSH_NT, 32, WSH_NT == 1 ? 2 : 0, 2, 1>;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8, h = argc > 2 ? atoi(argv[2]) : 225, w = argc > 3 ? atoi(argv[3]) : 400, c = argc > 4 ? atoi(argv[4]) : 64;
    const int co = WSH_NT * 32 < c ? c : c;   // c -> c
    const float scale = argc > 5 ? atof(argv[5]) : 1.f;
    ConvArgs a = {};
    const size_t nin = (size_t)n * h * w * c, nout = (size_t)n * h * w * co;
    float *din, *dout;
    unsigned short* dw;
    double* dst;
    const int nchunk = (c + 15) / 16;
    const int ntile_n = (co + CP::BN - 1) / CP::BN;
    const size_t wbytes = (size_t)ntile_n * nchunk * CP::WCHUNK_BYTES;
    CK(hipMalloc(&din, nin * 4)); CK(hipMalloc(&dout, nout * 4)); CK(hipMalloc(&dw, wbytes)); CK(hipMalloc(&dst, 4096 * 2 * 512 * 8));
    std::vector<float> hin(nin);
    srand(1);
    for (size_t i = 0; i < nin; ++i) hin[i] = scale * ((rand() & 0xffff) / 32768.f - 1.f) * 3.f;
    CK(hipMemcpy(din, hin.data(), nin * 4, hipMemcpyHostToDevice));
    std::vector<unsigned short> hw(wbytes / 2);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = scale == 0.f ? 0 : (unsigned short)(0x2000 + (rand() & 0x1fff) + ((rand() & 1) << 15));   // fp16 in +-[2^-7, 2^-3)
    CK(hipMemcpy(dw, hw.data(), wbytes, hipMemcpyHostToDevice));
    a.in1 = din; a.wp = reinterpret_cast<const float*>(dw); a.out = dout; a.stats = dst;
    void* zp = nullptr;
    CK(hipGetSymbolAddress(&zp, HIP_SYMBOL(rcf_zero_page)));
    a.zero = static_cast<const float*>(zp);
    a.n = n; a.h_in = h; a.w_in = w; a.c1 = c; a.c2 = 0; a.h1 = h; a.w1 = w; a.gather1 = 0;
    a.h_out = h; a.w_out = w; a.c_out = co; a.pad = 1; a.pad_x = 1; a.stride = 1; a.gstep = 1; a.accumulate = 0;
    a.os = 1; a.ooy = 0; a.oox = 0; a.ohp = h; a.owp = w; a.ioy = 0; a.iox = 0;
    a.vt = 0; a.hp = h + 1; a.nimg = n; a.inv_hp = 1.f / (h + 1); a.phase_sum = 0; a.wp_phase_stride = 0; a.sy = 1.f; a.sx = 1.f;
    a.tiles_x = (w + CP::PX - 1) / CP::PX; a.tiles_y = (h + CP::TH - 1) / CP::TH; a.ntiles = n * a.tiles_x * a.tiles_y;
    a.nchunk1 = nchunk; a.nchunk2 = 0;
    using L = WsLayout<CP>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_ws_kernel<CP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, L::LDS_BYTES));
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_split_ws_kernel<CP, false>, WS_THREADS, L::LDS_BYTES));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int gx = (per_cu > 2 ? 2 : per_cu) * prop.multiProcessorCount / ntile_n;
    if (gx > a.ntiles) gx = a.ntiles;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((conv_split_ws_kernel<CP, false>), dim3(gx, ntile_n, 1), dim3(WS_THREADS), L::LDS_BYTES, 0, a);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((conv_split_ws_kernel<CP, false>), dim3(gx, ntile_n, 1), dim3(WS_THREADS), L::LDS_BYTES, 0, a);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double gf = 2.0 * n * h * w * (double)co * 9 * c / 1e9;
    printf("%d x %d x %d x %d -> %d: occupancy %d/CU, grid %d x %d, LDS %d B: %.4f ms, %.1f TF/s algorithmic (x3 executed = %.0f)\n", n, h, w, c, co, per_cu, gx,
           ntile_n, L::LDS_BYTES, ms, gf / ms, 3 * gf / ms);
    return 0;
}
