// The remaining (HBM-bound) pieces of the FusionNet training step on gfx950: max pool, nearest-upsample
// backward, the 3x3 C->1 output head fused with the depth map, masked L1 loss, Adam, layout transforms and the
// radar point->grid scatter.  Reference citations are in include/rcf_hip.h next to each entry point.
#include "rcf_common.h"

namespace {

inline unsigned nblk(long long n, int per) { return (unsigned)((n + per - 1) / per); }

// ---------------------------------------------------------------- MaxPool2d(3, 2, 1)
template <class S>
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          unsigned char* __restrict__ idx, int n, int h, int w, int c, int ho,
                                                          int wo) {
    const int c4n = c >> 2;
    const long long total = (long long)n * ho * wo * c4n;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        long long t = g;
        const int cg = t % c4n; t /= c4n;
        const int ox = t % wo; t /= wo;
        const int oy = t % ho;
        const int img = t / ho;
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned bi[4] = {0, 0, 0, 0};
        bool any = false;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if (iy < 0 || iy >= h) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if (ix < 0 || ix >= w) continue;
                const f32x4 v = rcf_ld4<S>(in, (((size_t)img * h + iy) * w + ix) * c + cg * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!any || v[j] > best[j] || v[j] != v[j]) { best[j] = v[j]; bi[j] = ky * 3 + kx; }
                }
                any = true;
            }
        }
        const size_t o = (((size_t)img * ho + oy) * wo + ox) * c + cg * 4;
        rcf_st4<S>(out, o, best);
        *reinterpret_cast<unsigned*>(idx + o) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    }
}

template <class S>
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const float* __restrict__ dout, const unsigned char* __restrict__ idx,
                                                          float* __restrict__ din, int acc, int n, int h, int w, int c, int ho,
                                                          int wo) {
    const int c4n = c >> 2;
    const long long total = (long long)n * h * w * c4n;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        long long t = g;
        const int cg = t % c4n; t /= c4n;
        const int ix = t % w; t /= w;
        const int iy = t % h;
        const int img = t / h;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        // output rows whose 3-window [2oy-1, 2oy+1] contains iy
        const int oy_lo = iy >> 1, oy_hi = (iy + 1) >> 1;
        const int ox_lo = ix >> 1, ox_hi = (ix + 1) >> 1;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            if (oy >= ho) continue;
            const int ky = iy - (oy * 2 - 1);
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                if (ox >= wo) continue;
                const unsigned tap = ky * 3 + (ix - (ox * 2 - 1));
                const size_t o = (((size_t)img * ho + oy) * wo + ox) * c + cg * 4;
                const unsigned pk = *reinterpret_cast<const unsigned*>(idx + o);
                const f32x4 d = rcf_ld4<S>(dout, o);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (((pk >> (8 * j)) & 0xffu) == tap) s[j] += d[j];
            }
        }
        const size_t i = (((size_t)img * h + iy) * w + ix) * c + cg * 4;
        if (acc) s += rcf_ld4<S>(din, i);
        rcf_st4<S>(din, i, s);
    }
}

// ---------------------------------------------------------------- nearest upsample backward
__device__ __forceinline__ int nearest_src(int dst, float scale, int n_src) {
    return min((int)floorf((float)dst * scale), n_src - 1);
}
// [lo, hi) of destination indices that map to source index s (the map is monotone non-decreasing)
__device__ __forceinline__ void nearest_range(int s, float scale, int n_src, int n_dst, int* lo, int* hi) {
    int d = (int)((float)s / scale) - 2;
    if (d < 0) d = 0;
    while (d < n_dst && nearest_src(d, scale, n_src) < s) ++d;
    *lo = d;
    while (d < n_dst && nearest_src(d, scale, n_src) == s) ++d;
    *hi = d;
}

template <class S>
__global__ void __launch_bounds__(256) upsample_bwd_kernel(const float* __restrict__ dup, float* __restrict__ dsrc, int acc,
                                                           int n, int hu, int wu, int hs, int ws, int c, float sy, float sx) {
    const int c4n = c >> 2;
    const long long total = (long long)n * hs * ws * c4n;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        long long t = g;
        const int cg = t % c4n; t /= c4n;
        const int x = t % ws; t /= ws;
        const int y = t % hs;
        const int img = t / hs;
        int y0, y1, x0, x1;
        nearest_range(y, sy, hs, hu, &y0, &y1);
        nearest_range(x, sx, ws, wu, &x0, &x1);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx)
                s += rcf_ld4<S>(dup, (((size_t)img * hu + yy) * wu + xx) * c + cg * 4);
        const size_t i = (((size_t)img * hs + y) * ws + x) * c + cg * 4;
        if (acc) s += rcf_ld4<S>(dsrc, i);
        rcf_st4<S>(dsrc, i, s);
    }
}

// ---------------------------------------------------------------- output head: conv3x3 C->1 + depth map
// One lane per (pixel, 4-channel group): a pixel's C channels are one contiguous run, so each wave load is
// fully coalesced; the C/4 partial dot products are combined with lane shuffles.
template <class S>
__global__ void __launch_bounds__(256) head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                       float* __restrict__ logit, float* __restrict__ depth, int n, int h, int w,
                                                       int c, float dmin, float dmax) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [tap][c]
    const int c4n = c >> 2;
    for (int i = threadIdx.x; i < 9 * c; i += 256) wl[(i % 9) * c + i / 9] = wgt[i];   // OIHW [1][c][3][3] -> [tap][c]
    __syncthreads();
    const long long npix = (long long)n * h * w;
    const long long total = npix * c4n;
    const long long span = (long long)gridDim.x * 256;
    const long long iters = (total + span - 1) / span;
    for (long long it = 0; it < iters; ++it) {
        const long long g = it * span + (long long)blockIdx.x * 256 + threadIdx.x;
        const bool live = g < total;
        const long long p = live ? g / c4n : 0;
        const int cg = live ? (int)(g % c4n) : 0;
        const int px = p % w;
        const int py = (p / w) % h;
        const long long img = p / ((long long)w * h);
        float s = 0.f;
        if (live) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = py - 1 + ky;
                if (iy < 0 || iy >= h) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = px - 1 + kx;
                    if (ix < 0 || ix >= w) continue;
                    const f32x4 v = rcf_ld4<S>(x, (size_t)((img * h + iy) * w + ix) * c + cg * 4);
                    const f32x4 k = *reinterpret_cast<const f32x4*>(wl + (ky * 3 + kx) * c + cg * 4);
                    s += v[0] * k[0] + v[1] * k[1] + v[2] * k[2] + v[3] * k[3];
                }
            }
        }
        for (int off = 1; off < c4n; off <<= 1) s += __shfl_xor(s, off);
        if (live && cg == 0) {
            logit[p] = s;
            depth[p] = dmin / (1.f / (1.f + expf(-s)) + dmin / dmax);
        }
    }
}

__global__ void __launch_bounds__(256) head_bwd_logit_kernel(const float* __restrict__ ddepth, const float* __restrict__ logit,
                                                             float* __restrict__ dlogit, long long n, float dmin, float dmax) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float sg = 1.f / (1.f + expf(-logit[i]));
        const float den = sg + dmin / dmax;
        dlogit[i] = ddepth[i] * (-dmin / (den * den)) * sg * (1.f - sg);
    }
}

template <class S>
__global__ void __launch_bounds__(256) head_bwd_dgrad_kernel(const float* __restrict__ dl, const float* __restrict__ wgt,
                                                             float* __restrict__ dx, int n, int h, int w, int c) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    const int c4n = c >> 2;
    for (int i = threadIdx.x; i < 9 * c; i += 256) wl[(i % 9) * c + i / 9] = wgt[i];
    __syncthreads();
    const long long total = (long long)n * h * w * c4n;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const long long p = g / c4n;
        const int cg = (int)(g % c4n);
        const int px = p % w;
        const int py = (p / w) % h;
        const long long img = p / ((long long)w * h);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        // forward: o[q] = sum x[q + (ky-1,kx-1)] w[tap]  =>  dx[p] = sum_tap dl[p - (ky-1,kx-1)] w[tap]
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int qy = py + 1 - ky;
            if (qy < 0 || qy >= h) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int qx = px + 1 - kx;
                if (qx < 0 || qx >= w) continue;
                const float d = dl[(img * h + qy) * w + qx];
                s += d * *reinterpret_cast<const f32x4*>(wl + (ky * 3 + kx) * c + cg * 4);
            }
        }
        rcf_st4<S>(dx, (size_t)p * c + cg * 4, s);
    }
}

// ---- tile versions of the three head kernels (C <= 64): a block owns an 8 x 32 pixel tile (no per-element divisions), every
// activation element is read exactly once, and what the 3x3 stencil shares between neighbours goes through LDS.
constexpr int HT_H = 8, HT_W = 32, HT_HX = HT_W + 2, HT_HY = HT_H + 2, HT_NP = HT_HX * HT_HY;

template <int CTRL>
__device__ __forceinline__ float head_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the C4N adjacent lanes that hold one pixel's channel quads (every lane ends up with the total)
template <int C4N>
__device__ __forceinline__ float head_lane_sum(float v) {
    if (C4N >= 2) v += head_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
    if (C4N >= 4) v += head_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
    if (C4N >= 8) v += head_dpp<0x141>(v);   // row_half_mirror
    if (C4N >= 16) v += head_dpp<0x140>(v);  // row_mirror
    return v;
}

__device__ __forceinline__ void head_tile_origin(int tile, int h, int w, int* img, int* oy0, int* ox0) {
    const int tiles_x = (w + HT_W - 1) / HT_W, tiles_y = (h + HT_H - 1) / HT_H;
    const int tx = tile % tiles_x;
    const int t2 = tile / tiles_x;
    *img = t2 / tiles_y;
    *oy0 = (t2 % tiles_y) * HT_H;
    *ox0 = tx * HT_W;
}

// forward: per halo pixel the nine per-tap dot products T[pixel][tap] = <x[pixel], w[tap]> (x read once, cross-lane sum by DPP),
// then logit[q] = sum_tap T[q + tap offset][tap] from LDS.
template <int C4N, class S>
__global__ void __launch_bounds__(256) head_fwd_tile_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                            const float* __restrict__ wgt, float* __restrict__ logit,
                                                            float* __restrict__ depth, int n, int h, int w, float dmin, float dmax) {
    constexpr int C = 4 * C4N;
    __shared__ __attribute__((aligned(16))) float wl[9 * C];
    __shared__ float T[HT_NP * 9];
    for (int i = threadIdx.x; i < 9 * C; i += 256) wl[(i % 9) * C + i / 9] = wgt[i];   // OIHW [1][c][3][3] -> [tap][c]
    int img, oy0, ox0;
    head_tile_origin(blockIdx.x, h, w, &img, &oy0, &ox0);
    __syncthreads();
    const int cg = threadIdx.x % C4N;
    f32x4 k[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) k[t] = *reinterpret_cast<const f32x4*>(wl + t * C + cg * 4);
    // coef != nullptr: x is the raw conv output z of the previous block and y = lrelu(z * scale + shift) is applied on load
    // (that block's activation tensor is then never written)
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (coef != nullptr) {
        sc = *reinterpret_cast<const f32x4*>(coef + cg * 4);
        sh = *reinterpret_cast<const f32x4*>(coef + C + cg * 4);
    }
    constexpr int NIT = (HT_NP * C4N + 255) / 256;
#pragma unroll 4
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + 256 * it;
        const int hp = i / C4N;
        const int hy = hp / HT_HX, hx = hp - hy * HT_HX;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        const bool ok = hp < HT_NP && iy >= 0 && iy < h && ix >= 0 && ix < w;
        f32x4 v = rcf_ld4<S>(x, (((size_t)img * h + (ok ? iy : 0)) * w + (ok ? ix : 0)) * C + cg * 4);
        if (coef != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = rcf_lrelu(v[j] * sc[j] + sh[j]);
        }
        if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float s = head_lane_sum<C4N>(v[0] * k[t][0] + v[1] * k[t][1] + v[2] * k[t][2] + v[3] * k[t][3]);
            if (hp < HT_NP && cg == t % C4N) T[hp * 9 + t] = s;
        }
    }
    __syncthreads();
    const int ty = threadIdx.x / HT_W, tx = threadIdx.x % HT_W;
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy < h && ox < w) {
        float s = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) s += T[((ty + ky) * HT_HX + tx + kx) * 9 + ky * 3 + kx];
        const size_t p = ((size_t)img * h + oy) * w + ox;
        logit[p] = s;
        depth[p] = dmin / (1.f / (1.f + expf(-s)) + dmin / dmax);
    }
}

// The same forward for C = 32 (the published decoder's last block) and STORED bf16 activations (inference) on the matrix pipe [r4].
// head_fwd_tile_kernel is VALU-bound -- its nine per-tap dot products need a cross-lane sum each, ~100 VALU per 16 B loaded -- so it takes
// the same 0.42 ms for a bf16 batch of 8 as for the fp32 one (1.8 TB/s) and 1.4-1.6 ms per bf16 inference batch of 32.
// T[pixel][tap] = sum_c x[pixel][c] w[tap][c] is a [pixels x 32] x [32 x 9] product: one wave takes 32 halo pixels at a time, lane (i, h) of
// step s loads the 8 channels [16 s + 8 h, + 8) of pixel i (16 B: exactly its A fragment of v_mfma_f32_32x32x16_bf16), B holds the three
// exact bf16 planes of the fp32 weights (columns 9..31 zero): products exact, fp32 accumulate, 6 MFMAs per 32 pixels.  The D fragment's
// column is the lane's tap; the 3x3 gather over T is the tile kernel's.  0.76 ms per batch of 32 (3.9 TB/s).
// (The same product on v_mfma_f32_32x32x2_f32 for fp32 tensors / BatchNorm-on-load was built and measured: 8-28 % SLOWER than the tile
// kernel -- 16 f32 MFMAs of 64 cycles per 32 pixels -- and is not in the tree.)
typedef __bf16 hd_bf16x8 __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(256) head_fwd_mfma_b16_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                                float* __restrict__ logit, float* __restrict__ depth, int n, int h, int w,
                                                                float dmin, float dmax) {
    constexpr int C = 32;
    constexpr int NG = (HT_NP + 31) / 32;   // groups of 32 halo pixels
    __shared__ float T[NG * 32 * 9];
    int img, oy0, ox0;
    head_tile_origin(blockIdx.x, h, w, &img, &oy0, &ox0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // B fragments of the two 16-channel steps: lane (j = tap, h), k index e of step s <-> channel 16 s + 8 h + e, OIHW [1][c][3][3] ->
    // wgt[c * 9 + tap]; the three exact planes of w (by truncation: 3 x 8 significant bits)
    hd_bf16x8 wb[2][3];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        unsigned pl[3][4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = li < 9 ? wgt[(st * 16 + lh * 8 + e) * 9 + li] : 0.f;
            const unsigned x0 = __float_as_uint(v) & 0xffff0000u;
            const float r1 = v - __uint_as_float(x0);
            const unsigned x1 = __float_as_uint(r1) & 0xffff0000u;
            const unsigned x2 = __float_as_uint(r1 - __uint_as_float(x1)) & 0xffff0000u;
            const unsigned xs[3] = {x0, x1, x2};
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if (e & 1) pl[p][e >> 1] |= xs[p];
                else pl[p][e >> 1] = xs[p] >> 16;
            }
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const uint4 u = {pl[p][0], pl[p][1], pl[p][2], pl[p][3]};
            wb[st][p] = __builtin_bit_cast(hd_bf16x8, u);
        }
    }
    const unsigned short* xb = reinterpret_cast<const unsigned short*>(x);
    for (int g = wave; g < NG; g += 4) {
        const int hp = g * 32 + li;
        const int hy = hp / HT_HX, hx = hp - hy * HT_HX;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        const bool ok = hp < HT_NP && iy >= 0 && iy < h && ix >= 0 && ix < w;
        const size_t base = (((size_t)img * h + (ok ? iy : 0)) * w + (ok ? ix : 0)) * C;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            uint4 u = *reinterpret_cast<const uint4*>(xb + base + st * 16 + lh * 8);
            if (!ok) u = uint4{0u, 0u, 0u, 0u};
            const hd_bf16x8 av = __builtin_bit_cast(hd_bf16x8, u);
#pragma unroll
            for (int p = 2; p >= 0; --p) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, wb[st][p], acc, 0, 0, 0);   // smallest plane first
        }
        if (li < 9) {   // D fragment: value r of lane (tap, lh) is pixel row rcf_mfma_row(r, lh)
#pragma unroll
            for (int r = 0; r < 16; ++r) T[(g * 32 + rcf_mfma_row(r, lh)) * 9 + li] = acc[r];
        }
    }
    __syncthreads();
    const int ty = threadIdx.x / HT_W, tx = threadIdx.x % HT_W;
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy < h && ox < w) {
        float s = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) s += T[((ty + ky) * HT_HX + tx + kx) * 9 + ky * 3 + kx];
        const size_t p = ((size_t)img * h + oy) * w + ox;
        logit[p] = s;
        depth[p] = dmin / (1.f / (1.f + expf(-s)) + dmin / dmax);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The encoder's 'weight_and_project' fusion (src/networks.py:863-866) for INFERENCE on stored bf16 tensors, in one pass [r4]:
//     out = sigmoid(BN_w(W1 d)) * BN_p(W2 d) + img,   BN in eval mode = a per-channel affine map.
// The training arrangement needs the batch statistics of W1 d and W2 d before it can apply them -- two 1x1 convolutions that write zw, zp
// and an elementwise pass that reads them back with img: 2 c_d + 6 c_i elements of traffic per pixel.  With the statistics known the
// two products, the affine maps, the sigmoid gate and the sum are one streaming kernel: c_d + 2 c_i per pixel (2.8x less at
// c_i = 2 c_d).  Structure of conv1x1_b16_kernel (rcf_conv_b16_dma.h): 32 pixels x 16 channels of an NHWC bf16 tensor are the MFMA's
// A operand as they lie in memory, both weight matrices (scaled by their BatchNorm scale while they are loaded) live in registers for
// the wave's lifetime, no LDS, no barrier; blockIdx.y takes 32 NT output channels.  The D fragments of the two products have the
// same (pixel, channel) layout, so the gate needs no exchange; pairs of lanes swap one value to store one dword per pixel pair.
template <int CTRL>
__device__ __forceinline__ unsigned hd_dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

template <int KST, int NT, int MT>
__global__ void __launch_bounds__(256, 2) fuse_wp_b16_kernel(const unsigned short* __restrict__ d, const float* __restrict__ w1,
                                                             const float* __restrict__ coef_w, const float* __restrict__ w2,
                                                             const float* __restrict__ coef_p, const unsigned short* __restrict__ img,
                                                             unsigned short* __restrict__ out, long long npix, int c_i) {
    constexpr int CD = 16 * KST;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5, odd = li & 1;
    const int co0 = blockIdx.y * 32 * NT;
    // B fragments: lane (li = output channel, lh), k index e of step s <-> input channel 16 s + 8 lh + e of OIHW [co][ci][1][1]
    hd_bf16x8 bw[2][KST][NT];
#pragma unroll
    for (int br = 0; br < 2; ++br) {
        const float* w = br ? w2 : w1;
        const float* coef = br ? coef_p : coef_w;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const int co = co0 + ni * 32 + li;
            const bool ok = co < c_i;
            const float sc = ok ? coef[co] : 0.f;   // row 0 of the coefficient table: gamma / sqrt(var + eps)
#pragma unroll
            for (int s = 0; s < KST; ++s) {
                const float* wr = w + (size_t)(ok ? co : 0) * CD + s * 16 + lh * 8;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(wr), hi = *reinterpret_cast<const f32x4*>(wr + 4);
                const uint4 u = {rcf_f2b2(lo[0] * sc, lo[1] * sc), rcf_f2b2(lo[2] * sc, lo[3] * sc), rcf_f2b2(hi[0] * sc, hi[1] * sc),
                                 rcf_f2b2(hi[2] * sc, hi[3] * sc)};
                bw[br][s][ni] = __builtin_bit_cast(hd_bf16x8, u);
            }
        }
    }
    // shifts (row 1: beta - mean * scale) of the channel pair (cp, cp + 1) this lane stores
    float sw[NT][2], sp[NT][2];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const int cp = (co0 + ni * 32 + li) & ~1;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            sw[ni][e] = cp + e < c_i ? coef_w[c_i + cp + e] : 0.f;
            sp[ni][e] = cp + e < c_i ? coef_p[c_i + cp + e] : 0.f;
        }
    }
    const long long nblk32 = (npix + 31) / 32;
    const long long nwaves = (long long)gridDim.x * 4;
    for (long long blk0 = ((long long)blockIdx.x * 4 + wave) * MT; blk0 < nblk32; blk0 += nwaves * MT) {
        // every load of the trip -- the A operands and the image words of all MT blocks -- is in flight before the first MFMA
        uint4 av[MT][KST];
        unsigned iw[MT][8][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const long long p = (blk0 + mt) * 32 + li;
#pragma unroll
            for (int s = 0; s < KST; ++s)
                av[mt][s] = p < npix ? *reinterpret_cast<const uint4*>(d + (size_t)p * CD + s * 16 + lh * 8) : uint4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const long long p = (blk0 + mt) * 32 + rcf_mfma_row(2 * g, lh) + odd;
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) {
                    const int cp = (co0 + ni * 32 + li) & ~1;
                    iw[mt][g][ni] = *reinterpret_cast<const unsigned*>(img + ((p < npix && cp < c_i) ? (size_t)p * c_i + cp : 0));
                }
            }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const long long pb0 = (blk0 + mt) * 32;
            f32x16 acc[2][NT];
#pragma unroll
            for (int br = 0; br < 2; ++br)
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[br][ni][r] = 0.f;
#pragma unroll
                    for (int s = 0; s < KST; ++s)
                        acc[br][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hd_bf16x8, av[mt][s]), bw[br][s][ni],
                                                                              acc[br][ni], 0, 0, 0);
                }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int rj = 2 * g;                                    // accumulator rows rj, rj + 1 are consecutive pixels
                const long long p = pb0 + rcf_mfma_row(rj, lh) + odd;   // the pixel this lane stores
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) {
                    const int cp = (co0 + ni * 32 + li) & ~1;
                    float v[2][2];   // [branch][channel cp, cp + 1] at pixel p
#pragma unroll
                    for (int br = 0; br < 2; ++br) {
                        const float a0 = acc[br][ni][rj], a1 = acc[br][ni][rj + 1];
                        const float mine = odd ? a1 : a0, give = odd ? a0 : a1;
                        const float got = __uint_as_float(hd_dpp_u32<0xB1>(__float_as_uint(give)));
                        v[br][0] = odd ? got : mine;
                        v[br][1] = odd ? mine : got;
                    }
                    const float i0 = __uint_as_float(iw[mt][g][ni] << 16), i1 = __uint_as_float(iw[mt][g][ni] & 0xffff0000u);
                    // the gate on the hardware's exp2 / reciprocal (1 ulp each; the result is rounded to 8 bits): libm's expf and an IEEE
                    // division are ~30 VALU instructions per element, which made this HBM kernel VALU-bound
                    const float g0 = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * (v[0][0] + sw[ni][0])));
                    const float g1 = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * (v[0][1] + sw[ni][1])));
                    const float o0 = g0 * (v[1][0] + sp[ni][0]) + i0;
                    const float o1 = g1 * (v[1][1] + sp[ni][1]) + i1;
                    if (p < npix && cp < c_i) *reinterpret_cast<unsigned*>(out + (size_t)p * c_i + cp) = rcf_f2b2(o0, o1);
                }
            }
        }
    }
}

// input gradient: dx[p] = sum_tap dl[p - (ky-1, kx-1)] w[tap]; the dl halo tile sits in LDS, the weights in registers
template <int C4N, class S>
__global__ void __launch_bounds__(256) head_bwd_dgrad_tile_kernel(const float* __restrict__ dl, const float* __restrict__ wgt,
                                                                  float* __restrict__ dx, int n, int h, int w) {
    constexpr int C = 4 * C4N;
    __shared__ __attribute__((aligned(16))) float wl[9 * C];
    __shared__ float D[HT_NP];
    for (int i = threadIdx.x; i < 9 * C; i += 256) wl[(i % 9) * C + i / 9] = wgt[i];
    int img, oy0, ox0;
    head_tile_origin(blockIdx.x, h, w, &img, &oy0, &ox0);
    for (int hp = threadIdx.x; hp < HT_NP; hp += 256) {
        const int hy = hp / HT_HX, hx = hp - hy * HT_HX;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        D[hp] = (iy >= 0 && iy < h && ix >= 0 && ix < w) ? dl[((size_t)img * h + iy) * w + ix] : 0.f;
    }
    __syncthreads();
    const int cg = threadIdx.x % C4N;
    f32x4 k[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) k[t] = *reinterpret_cast<const f32x4*>(wl + t * C + cg * 4);
#pragma unroll
    for (int it = 0; it < C4N; ++it) {
        const int pix = (threadIdx.x + 256 * it) / C4N;
        const int ty = pix / HT_W, tx = pix % HT_W;
        const int oy = oy0 + ty, ox = ox0 + tx;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) s += D[(ty + 2 - ky) * HT_HX + tx + 2 - kx] * k[ky * 3 + kx];
        if (oy < h && ox < w) rcf_st4<S>(dx, (((size_t)img * h + oy) * w + ox) * C + cg * 4, s);
    }
}

// weight gradient: persistent blocks over tiles; acc[tap] += dl[p - offset] * x[p] with the dl halo tile in LDS
template <int C4N, class S>
__global__ void __launch_bounds__(256) head_bwd_wgrad_tile_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                                  const float* __restrict__ dl, float* __restrict__ ws, int n, int h,
                                                                  int w) {
    constexpr int C = 4 * C4N;
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [256 / C4N][9][C] for the final reduction; D aliases its start
    float* D = sm;
    const int cg = threadIdx.x % C4N;
    float acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = 0.f;
    const int tiles_x = (w + HT_W - 1) / HT_W, tiles_y = (h + HT_H - 1) / HT_H;
    const int ntiles = n * tiles_x * tiles_y;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (coef != nullptr) {   // x is the previous block's raw conv output: apply its BatchNorm + lrelu on load
        sc = *reinterpret_cast<const f32x4*>(coef + cg * 4);
        sh = *reinterpret_cast<const f32x4*>(coef + C + cg * 4);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int img, oy0, ox0;
        head_tile_origin(tile, h, w, &img, &oy0, &ox0);
        f32x4 v[C4N];
#pragma unroll
        for (int it = 0; it < C4N; ++it) {   // this tile's activations: issued before the barrier that waits for the dl tile
            const int pix = (threadIdx.x + 256 * it) / C4N;
            const int oy = oy0 + pix / HT_W, ox = ox0 + pix % HT_W;
            const bool ok = oy < h && ox < w;
            v[it] = rcf_ld4<S>(x, (((size_t)img * h + (ok ? oy : 0)) * w + (ok ? ox : 0)) * C + cg * 4);
            if (coef != nullptr) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[it][j] = rcf_lrelu(v[it][j] * sc[j] + sh[j]);
            }
            if (!ok) v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();   // previous tile's D readers are done
        for (int hp = threadIdx.x; hp < HT_NP; hp += 256) {
            const int hy = hp / HT_HX, hx = hp - hy * HT_HX;
            const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
            D[hp] = (iy >= 0 && iy < h && ix >= 0 && ix < w) ? dl[((size_t)img * h + iy) * w + ix] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < C4N; ++it) {
            const int pix = (threadIdx.x + 256 * it) / C4N;
            const int ty = pix / HT_W, tx = pix % HT_W;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float d = D[(ty + 2 - ky) * HT_HX + tx + 2 - kx];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[ky * 3 + kx][j] += d * v[it][j];
                }
        }
    }
    __syncthreads();
    const int pl = threadIdx.x / C4N;
    constexpr int PPB = 256 / C4N;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) sm[(pl * 9 + t) * C + cg * 4 + j] = acc[t][j];
    __syncthreads();
    for (int e = threadIdx.x; e < 9 * C; e += 256) {
        float s = 0.f;
        for (int r = 0; r < PPB; ++r) s += sm[r * 9 * C + e];
        ws[(size_t)blockIdx.x * 9 * C + e] = s;
    }
}

constexpr int HEAD_WG_BLOCKS = 1024;

template <class S>
__global__ void __launch_bounds__(256) head_bwd_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dl,
                                                             float* __restrict__ ws, int n, int h, int w, int c) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int c4n = c >> 2;
    const int cg = threadIdx.x % c4n;
    const int pl = threadIdx.x / c4n;
    const int ppb = 256 / c4n;
    float acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = 0.f;
    const long long npix = (long long)n * h * w;
    for (long long p = (long long)blockIdx.x * ppb + pl; p < npix; p += (long long)gridDim.x * ppb) {
        const int px = p % w;
        const int py = (p / w) % h;
        const long long img = p / ((long long)w * h);
        const f32x4 v = rcf_ld4<S>(x, (size_t)p * c + cg * 4);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int qy = py + 1 - ky;
            if (qy < 0 || qy >= h) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int qx = px + 1 - kx;
                if (qx < 0 || qx >= w) continue;
                const float d = dl[(img * h + qy) * w + qx];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[ky * 3 + kx][j] += d * v[j];
            }
        }
    }
    // block reduce: sm[pl][tap][c]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) sm[(pl * 9 + t) * c + cg * 4 + j] = acc[t][j];
    __syncthreads();
    for (int e = threadIdx.x; e < 9 * c; e += 256) {
        float s = 0.f;
        for (int r = 0; r < ppb; ++r) s += sm[r * 9 * c + e];
        ws[(size_t)blockIdx.x * 9 * c + e] = s;
    }
}

// ws[nb][tap][c] -> dw OIHW [1][c][3][3]
__global__ void __launch_bounds__(64) head_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nb,
                                                               int c) {
    const int e = blockIdx.x;   // tap*c + ch
    double s = 0.0;
    for (int r = threadIdx.x; r < nb; r += 64) s += (double)ws[(size_t)r * 9 * c + e];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) dw[(e % c) * 9 + e / c] = (float)s;
}

// ---------------------------------------------------------------- masked L1 / L2 / smooth-L1 loss
// (src/fusionnet_model.py:245-275; src/fusionnet_losses.py:4-46: F.l1_loss / F.mse_loss / F.smooth_l1_loss (beta 1) with reduction 'mean'
// over the valid pixels).  KIND 0: |e|, 1: e^2, 2: 0.5 e^2 for |e| < 1 else |e| - 0.5.
constexpr int LOSS_BLOCKS = 1024;
template <int KIND>
__device__ __forceinline__ float loss_term(float e) {
    if (KIND == 1) return e * e;
    if (KIND == 2) { const float a = fabsf(e); return a < 1.f ? 0.5f * e * e : a - 0.5f; }
    return fabsf(e);
}
template <int KIND>
__device__ __forceinline__ float loss_slope(float e) {
    if (KIND == 1) return 2.f * e;
    if (KIND == 2) return fabsf(e) < 1.f ? e : (e > 0.f ? 1.f : -1.f);
    return e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
}

template <int KIND>
__global__ void __launch_bounds__(256) l1_loss_partial_kernel(const float* __restrict__ d, const float* __restrict__ gt,
                                                              const float* __restrict__ lidar, float* __restrict__ ws,
                                                              long long n) {
    __shared__ float sm[4][4];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float dv = d[i], l = lidar[i];
        const float g = l > 0.f ? 0.f : gt[i];   // mask out ground truth where lidar is available
        if (g > 0.f) { s[0] += loss_term<KIND>(dv - g); s[1] += 1.f; }
        if (l > 0.f) { s[2] += loss_term<KIND>(dv - l); s[3] += 1.f; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s[q] += __shfl_xor(s[q], off);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int q = 0; q < 4; ++q) sm[wave][q] = s[q];
    __syncthreads();
    if (threadIdx.x < 4) ws[blockIdx.x * 4 + threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

__global__ void __launch_bounds__(256) l1_loss_final_kernel(const float* __restrict__ ws, int nb, double* __restrict__ sums) {
    __shared__ double sm[4][4];
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int r = threadIdx.x; r < nb; r += 256)
        for (int q = 0; q < 4; ++q) s[q] += (double)ws[r * 4 + q];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s[q] += __shfl_xor(s[q], off);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int q = 0; q < 4; ++q) sm[wave][q] = s[q];
    __syncthreads();
    if (threadIdx.x < 4) sums[threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

__global__ void l1_loss_value_kernel(const double* __restrict__ sums, float w_lidar, float* __restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double ls = sums[0] / sums[1];          // mean over an empty set is NaN, as F.l1_loss
        const double ll = w_lidar > 0.f ? sums[2] / sums[3] : 0.0;
        loss[0] = (float)(ls + (double)w_lidar * ll);
        loss[1] = (float)ls;
        loss[2] = (float)ll;
    }
}

template <int KIND>
__global__ void __launch_bounds__(256) l1_loss_bwd_kernel(const float* __restrict__ d, const float* __restrict__ gt,
                                                          const float* __restrict__ lidar, const double* __restrict__ sums,
                                                          const float* __restrict__ upstream, float w_lidar,
                                                          float* __restrict__ dd, long long n) {
    const float up = upstream ? upstream[0] : 1.f;
    const float kg = (float)((double)up / sums[1]);
    const float kl = w_lidar > 0.f ? (float)((double)up * (double)w_lidar / sums[3]) : 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float dv = d[i], l = lidar[i];
        const float g = l > 0.f ? 0.f : gt[i];
        float r = 0.f;
        if (g > 0.f) r += kg * loss_slope<KIND>(dv - g);
        if (l > 0.f && w_lidar > 0.f) r += kl * loss_slope<KIND>(dv - l);
        dd[i] = r;
    }
}

// ---------------------------------------------------------------- local smoothness loss
// losses.smoothness_loss_func (src/fusionnet_losses.py:48-72; compute_loss with w_smoothness > 0 and loss_smoothness_kernel_size <= 1,
// src/fusionnet_model.py:277-281): mean over N x H x (W-1) of wx |P[y][x] - P[y][x+1]| + mean over N x (H-1) x W of wy |P[y][x] - P[y+1][x]|,
// wx = exp(-mean_c |I[c][y][x] - I[c][y][x+1]|) (wy alike).  image: NCHW (the public tensor), P: [N][H][W].
// Partials (sum x, count x, sum y, count y) per block -> l1_loss_final_kernel -> sums[4] (fp64): local to the rank, like the masked loss.
__device__ __forceinline__ float smooth_weight(const float* __restrict__ img, size_t i0, size_t i1, int c, size_t cstride) {
    float a = 0.f;
    for (int k = 0; k < c; ++k) a += fabsf(img[i0 + k * cstride] - img[i1 + k * cstride]);
    return expf(-a / (float)c);
}

__global__ void __launch_bounds__(256) smoothness_partial_kernel(const float* __restrict__ img, const float* __restrict__ p,
                                                                 float* __restrict__ ws, int n, int c, int h, int w) {
    __shared__ float sm[4][4];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const long long hw = (long long)h * w, tot = (long long)n * hw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const long long im = i / hw;
        const size_t ib = (size_t)im * c * hw + (size_t)y * w + x;   // channel 0 of this pixel in the NCHW image
        const float pv = p[i];
        if (x < w - 1) { s[0] += smooth_weight(img, ib, ib + 1, c, (size_t)hw) * fabsf(pv - p[i + 1]); s[1] += 1.f; }
        if (y < h - 1) { s[2] += smooth_weight(img, ib, ib + w, c, (size_t)hw) * fabsf(pv - p[i + w]); s[3] += 1.f; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s[q] += __shfl_xor(s[q], off);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int q = 0; q < 4; ++q) sm[wave][q] = s[q];
    __syncthreads();
    if (threadIdx.x < 4) ws[blockIdx.x * 4 + threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

__device__ __forceinline__ float sgnf(float e) { return e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f); }

// ddepth += upstream * w_smoothness * d(smoothness)/dP
__global__ void __launch_bounds__(256) smoothness_bwd_kernel(const float* __restrict__ img, const float* __restrict__ p,
                                                             const double* __restrict__ sums, const float* __restrict__ upstream,
                                                             float w_smoothness, float* __restrict__ dd, int n, int c, int h, int w) {
    const float up = (upstream ? upstream[0] : 1.f) * w_smoothness;
    const float kx = (float)((double)up / sums[1]), ky = (float)((double)up / sums[3]);
    const long long hw = (long long)h * w, tot = (long long)n * hw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const long long im = i / hw;
        const size_t ib = (size_t)im * c * hw + (size_t)y * w + x;
        const float pv = p[i];
        float r = 0.f;
        if (x < w - 1) r += kx * smooth_weight(img, ib, ib + 1, c, (size_t)hw) * sgnf(pv - p[i + 1]);
        if (x > 0) r -= kx * smooth_weight(img, ib - 1, ib, c, (size_t)hw) * sgnf(p[i - 1] - pv);
        if (y < h - 1) r += ky * smooth_weight(img, ib, ib + w, c, (size_t)hw) * sgnf(pv - p[i + w]);
        if (y > 0) r -= ky * smooth_weight(img, ib - w, ib, c, (size_t)hw) * sgnf(p[i - w] - pv);
        dd[i] += r;
    }
}

// ---------------------------------------------------------------- OutlierRemoval (7x7 min filter on sparse depth)
__global__ void __launch_bounds__(256) max_reduce_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ out_bits) {
    float m = 0.f;   // depth maps are >= 0: max(depth) >= 0 and non-negative floats order like their bit patterns
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, x[i]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(m));
}

__global__ void __launch_bounds__(256) outlier_removal_kernel(const float* __restrict__ depth, float* __restrict__ out,
                                                              const float* __restrict__ gmax, int h, int w, int k, float threshold) {
    extern __shared__ float tile[];   // (8 + k - 1) x (32 + k - 1) max-filled depth
    const int r = k / 2;
    const int tw = 32 + 2 * r, th = 8 + 2 * r;
    const float max_value = 10.f * gmax[0];
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 8;
    const float* img = depth + (size_t)blockIdx.z * h * w;
    for (int i = threadIdx.x; i < tw * th; i += 256) {
        const int ty = i / tw, tx = i - ty * tw;
        const int y = y0 + ty - r, x = x0 + tx - r;
        float v = max_value;                                   // constant padding with max_value (:617-621)
        if (y >= 0 && y < h && x >= 0 && x < w) {
            const float d = img[(size_t)y * w + x];
            v = d > 0.f ? d : (d <= 0.f ? max_value : d);      // validity_map <= 0 -> max_value (:603-613)
        }
        tile[i] = v;
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= w || y >= h) return;
    float mn = tile[ly * tw + lx];
    for (int dy = 0; dy < k; ++dy)
        for (int dx = 0; dx < k; ++dx) mn = fminf(mn, tile[(ly + dy) * tw + lx + dx]);
    const float d = img[(size_t)y * w + x];
    out[(size_t)blockIdx.z * h * w + (size_t)y * w + x] = (mn < d - threshold) ? d * 0.f : d;   // depth * validity_map_clean
}

// ---------------------------------------------------------------- Adam over one flat arena
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                   float wd, float step_size, float inv_bc2_sqrt) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float gi = g[i];
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

// Graph-capturable form: the step count and the hyper-parameters live in device memory (state[8] = step, lr, beta1, beta2, eps,
// weight_decay, step_size, 1/sqrt(bias_correction2)); adam_tick_kernel advances the count and derives the two bias-correction
// factors in double (as the host does for rcf_adam_step), adam_dev_kernel applies the update with what it finds there.
__global__ void adam_tick_kernel(float* __restrict__ state) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float step = state[0] + 1.f;
    state[0] = step;
    const double bc1 = 1.0 - pow((double)state[2], (double)step);
    const double bc2 = 1.0 - pow((double)state[3], (double)step);
    state[6] = (float)((double)state[1] / bc1);
    state[7] = (float)(1.0 / sqrt(bc2));
}

__global__ void __launch_bounds__(256) adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long long n, const float* __restrict__ state) {
    const float b1 = state[2], b2 = state[3], eps = state[4], wd = state[5], step_size = state[6], inv_bc2_sqrt = state[7];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float gi = g[i];
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

// ---------------------------------------------------------------- layout
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int c,
                                                           long long hw) {
    const long long total = (long long)n * c * hw;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
        const int ch = o % c;
        const long long p = (o / c) % hw;
        const long long img = o / ((long long)c * hw);
        out[o] = in[(img * c + ch) * hw + p];
    }
}
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int c,
                                                           long long hw) {
    const long long total = (long long)n * c * hw;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
        const long long p = o % hw;
        const int ch = (o / hw) % c;
        const long long img = o / ((long long)c * hw);
        out[o] = in[(img * hw + p) * c + ch];
    }
}

// ---------------------------------------------------------------- radar point -> dense map scatter
// One thread per output pixel, looping over the K points in order (K <= ~100).  The canvas is W + 2*pad wide
// (pad = Wc/2); crop k occupies canvas columns [int(x_k) - pad, int(x_k) + pad); the output is canvas columns
// [pad, pad + W).  Crop rows are the bottom `hc` rows of the canvas.
// LOGITS: `crops` holds the correspondence LOGITS (RadarNetModel.forward(return_logits=True)).  The reference thresholds
// sigmoid(logit) < 0.5 (src/radarnet_main.py:563-567); sigmoid is monotonic with sigmoid(0) = 0.5, so the decision is taken on the SIGN of
// the logit -- no transcendental near the threshold, nothing that could differ by an ulp between this device's sigmoid and the CPU's --
// and the response written is 1 / (1 + expf(-logit)) of the surviving values (the maximum over points is still taken on the responses,
// like the reference: saturated responses tie, and the first point wins).
template <bool LOGITS>
__global__ void __launch_bounds__(256) radar_scatter_kernel(const float* __restrict__ crops, const float* __restrict__ pts, int k,
                                                            int h, int w, int hc, int wc, int strict, float* __restrict__ depth,
                                                            float* __restrict__ resp) {
    const int pad = wc / 2;
    const long long total = (long long)h * w;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const int x = g % w;
        const int y = g / w;
        const int cx = x + pad;           // canvas column
        const int cy = y - (h - hc);      // crop row
        float best = 0.f;                 // torch.max over tiles of a zero canvas: first max wins, all-zero -> index 0
        int arg = 0;
        bool first = true;
        for (int i = 0; i < k; ++i) {
            const int x0 = (int)pts[i * 3 + 0] - pad;
            float v = 0.f;
            const int u = cx - x0;
            if (cy >= 0 && u >= 0 && u < 2 * pad && u < wc) {
                v = crops[((size_t)i * hc + cy) * wc + u];
                if (LOGITS) v = v >= 0.f ? 1.f / (1.f + expf(-v)) : 0.f;
                else if (v < 0.5f) v = 0.f;    // thresholding any response less than 0.5 to 0
            }
            if (first || v > best) { best = v; arg = i; first = false; }
        }
        float dz;
        if (strict) {
            // reference: int64 `output` holds the argmax; for point_idx in order: where(output == idx, z_idx, output),
            // z truncated toward zero on the int64 fill, applied in place and sequentially.
            long long o = arg;
            for (int i = 0; i < k; ++i)
                if (o == (long long)i) o = (long long)pts[i * 3 + 2];
            dz = (float)o;
        } else {
            dz = pts[arg * 3 + 2];
        }
        if (best == 0.f) dz = 0.f;        // leave as 0s if we did not predict
        depth[g] = dz;
        resp[g] = best;
    }
}

}   // namespace

template <class S>
static int maxpool3x3s2_fwd_impl(const float* in, float* out, unsigned char* idx, int n, int h, int w, int c, void* stream) {
    if (!in || !out || !idx || n <= 0 || h <= 0 || w <= 0 || c <= 0) return RCF_EINVAL;
    if (c & 3) return RCF_EUNSUPPORTED;
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long long total = (long long)n * ho * wo * (c >> 2);
    unsigned b = nblk(total, 256); if (b > 8192) b = 8192;
    hipLaunchKernelGGL((maxpool_fwd_kernel<S>), dim3(b), dim3(256), 0, (hipStream_t)stream, in, out, idx, n, h, w, c, ho, wo);
    return rcf_launch_status();
}

template <class S>
static int maxpool3x3s2_bwd_impl(const float* dout, const unsigned char* idx, float* din, int din_accumulate, int n, int h,
                                    int w, int c, void* stream) {
    if (!dout || !idx || !din || n <= 0 || h <= 0 || w <= 0 || c <= 0) return RCF_EINVAL;
    if (c & 3) return RCF_EUNSUPPORTED;
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long long total = (long long)n * h * w * (c >> 2);
    unsigned b = nblk(total, 256); if (b > 8192) b = 8192;
    hipLaunchKernelGGL((maxpool_bwd_kernel<S>), dim3(b), dim3(256), 0, (hipStream_t)stream, dout, idx, din, din_accumulate, n, h, w, c,
                       ho, wo);
    return rcf_launch_status();
}

template <class S>
static int upsample_nearest_bwd_impl(const float* dup, float* dsrc, int dsrc_accumulate, int n, int h_up, int w_up,
                                        int h_src, int w_src, int c, void* stream) {
    if (!dup || !dsrc || n <= 0 || h_up <= 0 || w_up <= 0 || h_src <= 0 || w_src <= 0 || c <= 0) return RCF_EINVAL;
    if (c & 3) return RCF_EUNSUPPORTED;
    const long long total = (long long)n * h_src * w_src * (c >> 2);
    unsigned b = nblk(total, 256); if (b > 8192) b = 8192;
    hipLaunchKernelGGL((upsample_bwd_kernel<S>), dim3(b), dim3(256), 0, (hipStream_t)stream, dup, dsrc, dsrc_accumulate, n, h_up, w_up,
                       h_src, w_src, c, (float)h_src / (float)h_up, (float)w_src / (float)w_up);
    return rcf_launch_status();
}

static bool head_c_ok(int c) {
    if (c < 4 || (c & 3)) return false;
    const int c4 = c >> 2;
    return (c4 & (c4 - 1)) == 0 && c4 <= 64;
}

template <class S>
static int head_fwd_core(const float* x, const float* coef, const float* w, float* logit, float* depth, int n, int h, int w_, int c,
                         float min_depth, float max_depth, void* stream);

template <class S>
static int head_fwd_impl(const float* x, const float* w, float* logit, float* depth, int n, int h, int w_, int c,
                            float min_depth, float max_depth, void* stream) {
    return head_fwd_core<S>(x, nullptr, w, logit, depth, n, h, w_, c, min_depth, max_depth, stream);
}

template <class S>
static int head_fwd_bn_impl(const float* z, const float* coef, const float* w, float* logit, float* depth, int n, int h, int w_,
                               int c, float min_depth, float max_depth, void* stream) {
    if (!coef || (c >> 2) > 16) return coef ? RCF_EUNSUPPORTED : RCF_EINVAL;
    return head_fwd_core<S>(z, coef, w, logit, depth, n, h, w_, c, min_depth, max_depth, stream);
}

template <class S>
static int head_fwd_core(const float* x, const float* coef, const float* w, float* logit, float* depth, int n, int h, int w_, int c,
                         float min_depth, float max_depth, void* stream) {
    if (!x || !w || !logit || !depth || n <= 0 || h <= 0 || w_ <= 0) return RCF_EINVAL;
    if (!head_c_ok(c)) return RCF_EUNSUPPORTED;
    const int c4n = c >> 2;
    if (S::B16 && c == 32 && coef == nullptr && getenv("RCF_HEAD_MFMA") == nullptr) {   // stored bf16 activations of the published decoder's
        const unsigned nt = (unsigned)n * ((h + HT_H - 1) / HT_H) * ((w_ + HT_W - 1) / HT_W);   // head: on the matrix pipe (RCF_HEAD_MFMA=0: tile kernel)
        hipLaunchKernelGGL(head_fwd_mfma_b16_kernel, dim3(nt), dim3(256), 0, (hipStream_t)stream, x, w, logit, depth, n, h, w_, min_depth, max_depth);
        return rcf_launch_status();
    }
    if (c4n <= 16) {
        const unsigned nt = (unsigned)n * ((h + HT_H - 1) / HT_H) * ((w_ + HT_W - 1) / HT_W);
        hipStream_t st = (hipStream_t)stream;
#define RCF_HEAD_FWD(N) hipLaunchKernelGGL((head_fwd_tile_kernel<N, S>), dim3(nt), dim3(256), 0, st, x, coef, w, logit, depth, n, h, w_, min_depth, max_depth)
        switch (c4n) {
            case 1: RCF_HEAD_FWD(1); break;
            case 2: RCF_HEAD_FWD(2); break;
            case 4: RCF_HEAD_FWD(4); break;
            case 8: RCF_HEAD_FWD(8); break;
            default: RCF_HEAD_FWD(16); break;
        }
#undef RCF_HEAD_FWD
        return rcf_launch_status();
    }
    const long long total = (long long)n * h * w_ * (c >> 2);
    unsigned b = nblk(total, 256); if (b > 16384) b = 16384;
    hipLaunchKernelGGL((head_fwd_kernel<S>), dim3(b), dim3(256), 9 * c * sizeof(float), (hipStream_t)stream, x, w, logit, depth, n, h,
                       w_, c, min_depth, max_depth);
    return rcf_launch_status();
}

extern "C" int rcf_head_bwd_logit(const float* ddepth, const float* logit, float* dlogit, long long n_pix, float min_depth,
                                  float max_depth, void* stream) {
    if (!ddepth || !logit || !dlogit || n_pix <= 0) return RCF_EINVAL;
    unsigned b = nblk(n_pix, 256); if (b > 8192) b = 8192;
    hipLaunchKernelGGL(head_bwd_logit_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, ddepth, logit, dlogit, n_pix, min_depth,
                       max_depth);
    return rcf_launch_status();
}

template <class S>
static int head_bwd_dgrad_impl(const float* dlogit, const float* w, float* dx, int n, int h, int w_, int c, void* stream) {
    if (!dlogit || !w || !dx || n <= 0 || h <= 0 || w_ <= 0) return RCF_EINVAL;
    if (!head_c_ok(c)) return RCF_EUNSUPPORTED;
    const int c4n = c >> 2;
    if (c4n <= 16) {
        const unsigned nt = (unsigned)n * ((h + HT_H - 1) / HT_H) * ((w_ + HT_W - 1) / HT_W);
        hipStream_t st = (hipStream_t)stream;
#define RCF_HEAD_DG(N) hipLaunchKernelGGL((head_bwd_dgrad_tile_kernel<N, S>), dim3(nt), dim3(256), 0, st, dlogit, w, dx, n, h, w_)
        switch (c4n) {
            case 1: RCF_HEAD_DG(1); break;
            case 2: RCF_HEAD_DG(2); break;
            case 4: RCF_HEAD_DG(4); break;
            case 8: RCF_HEAD_DG(8); break;
            default: RCF_HEAD_DG(16); break;
        }
#undef RCF_HEAD_DG
        return rcf_launch_status();
    }
    const long long total = (long long)n * h * w_ * (c >> 2);
    unsigned b = nblk(total, 256); if (b > 16384) b = 16384;
    hipLaunchKernelGGL((head_bwd_dgrad_kernel<S>), dim3(b), dim3(256), 9 * c * sizeof(float), (hipStream_t)stream, dlogit, w, dx, n, h,
                       w_, c);
    return rcf_launch_status();
}

extern "C" size_t rcf_head_wgrad_workspace_floats(int n, int h, int w_, int c) {
    (void)n; (void)h; (void)w_;
    return (size_t)HEAD_WG_BLOCKS * 9 * (size_t)c;
}

template <class S>
static int head_bwd_wgrad_core(const float* x, const float* coef, const float* dlogit, float* dw, float* workspace, int n, int h,
                               int w_, int c, void* stream);

template <class S>
static int head_bwd_wgrad_impl(const float* x, const float* dlogit, float* dw, float* workspace, int n, int h, int w_, int c,
                                  void* stream) {
    return head_bwd_wgrad_core<S>(x, nullptr, dlogit, dw, workspace, n, h, w_, c, stream);
}

template <class S>
static int head_bwd_wgrad_bn_impl(const float* z, const float* coef, const float* dlogit, float* dw, float* workspace, int n,
                                     int h, int w_, int c, void* stream) {
    if (!coef || (c >> 2) > 16) return coef ? RCF_EUNSUPPORTED : RCF_EINVAL;
    return head_bwd_wgrad_core<S>(z, coef, dlogit, dw, workspace, n, h, w_, c, stream);
}

template <class S>
static int head_bwd_wgrad_core(const float* x, const float* coef, const float* dlogit, float* dw, float* workspace, int n, int h,
                               int w_, int c, void* stream) {
    if (!x || !dlogit || !dw || !workspace || n <= 0 || h <= 0 || w_ <= 0) return RCF_EINVAL;
    if (!head_c_ok(c)) return RCF_EUNSUPPORTED;
    const int ppb = 256 / (c >> 2);
    long long nb = ((long long)n * h * w_ + ppb - 1) / ppb;
    if (nb > HEAD_WG_BLOCKS) nb = HEAD_WG_BLOCKS;
    const int c4n = c >> 2;
    if (c4n <= 16) {
        const long long nt = (long long)n * ((h + HT_H - 1) / HT_H) * ((w_ + HT_W - 1) / HT_W);
        if (nb > nt) nb = nt;
        size_t lds = (size_t)ppb * 9 * c * sizeof(float);
        if (lds < HT_NP * sizeof(float)) lds = HT_NP * sizeof(float);
        hipStream_t st = (hipStream_t)stream;
#define RCF_HEAD_WG(N) hipLaunchKernelGGL((head_bwd_wgrad_tile_kernel<N, S>), dim3((unsigned)nb), dim3(256), lds, st, x, coef, dlogit, workspace, n, h, w_)
        switch (c4n) {
            case 1: RCF_HEAD_WG(1); break;
            case 2: RCF_HEAD_WG(2); break;
            case 4: RCF_HEAD_WG(4); break;
            case 8: RCF_HEAD_WG(8); break;
            default: RCF_HEAD_WG(16); break;
        }
#undef RCF_HEAD_WG
    } else
    hipLaunchKernelGGL((head_bwd_wgrad_kernel<S>), dim3((unsigned)nb), dim3(256), (size_t)ppb * 9 * c * sizeof(float),
                       (hipStream_t)stream, x, dlogit, workspace, n, h, w_, c);
    int rc = rcf_launch_status();
    if (rc != RCF_OK) return rc;
    hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3(9 * c), dim3(64), 0, (hipStream_t)stream, workspace, dw, (int)nb, c);
    return rcf_launch_status();
}

extern "C" size_t rcf_loss_workspace_floats(long long n_pix) {
    (void)n_pix;
    return (size_t)LOSS_BLOCKS * 4;
}

extern "C" int rcf_masked_loss_fwd(const float* depth, const float* gt, const float* lidar, float* workspace, double* sums,
                                   long long n_pix, int kind, void* stream) {
    if (!depth || !gt || !lidar || !workspace || !sums || n_pix <= 0 || kind < 0 || kind > 2) return RCF_EINVAL;
    long long nb = (n_pix + 255) / 256;
    if (nb > LOSS_BLOCKS) nb = LOSS_BLOCKS;
    hipStream_t st = (hipStream_t)stream;
    if (kind == 1) hipLaunchKernelGGL(l1_loss_partial_kernel<1>, dim3((unsigned)nb), dim3(256), 0, st, depth, gt, lidar, workspace, n_pix);
    else if (kind == 2) hipLaunchKernelGGL(l1_loss_partial_kernel<2>, dim3((unsigned)nb), dim3(256), 0, st, depth, gt, lidar, workspace, n_pix);
    else hipLaunchKernelGGL(l1_loss_partial_kernel<0>, dim3((unsigned)nb), dim3(256), 0, st, depth, gt, lidar, workspace, n_pix);
    int rc = rcf_launch_status();
    if (rc != RCF_OK) return rc;
    hipLaunchKernelGGL(l1_loss_final_kernel, dim3(1), dim3(256), 0, st, workspace, (int)nb, sums);
    return rcf_launch_status();
}

extern "C" int rcf_l1_loss_fwd(const float* depth, const float* gt, const float* lidar, float* workspace, double* sums,
                               long long n_pix, void* stream) {
    return rcf_masked_loss_fwd(depth, gt, lidar, workspace, sums, n_pix, RCF_LOSS_L1, stream);
}

extern "C" int rcf_l1_loss_value(const double* sums, float w_lidar, float* loss, void* stream) {
    if (!sums || !loss) return RCF_EINVAL;
    hipLaunchKernelGGL(l1_loss_value_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, w_lidar, loss);
    return rcf_launch_status();
}

extern "C" int rcf_masked_loss_bwd(const float* depth, const float* gt, const float* lidar, const double* sums, const float* upstream,
                                   float w_lidar, float* ddepth, long long n_pix, int kind, void* stream) {
    if (!depth || !gt || !lidar || !sums || !ddepth || n_pix <= 0 || kind < 0 || kind > 2) return RCF_EINVAL;
    unsigned b = nblk(n_pix, 256); if (b > 8192) b = 8192;
    hipStream_t st = (hipStream_t)stream;
    if (kind == 1) hipLaunchKernelGGL(l1_loss_bwd_kernel<1>, dim3(b), dim3(256), 0, st, depth, gt, lidar, sums, upstream, w_lidar, ddepth, n_pix);
    else if (kind == 2) hipLaunchKernelGGL(l1_loss_bwd_kernel<2>, dim3(b), dim3(256), 0, st, depth, gt, lidar, sums, upstream, w_lidar, ddepth, n_pix);
    else hipLaunchKernelGGL(l1_loss_bwd_kernel<0>, dim3(b), dim3(256), 0, st, depth, gt, lidar, sums, upstream, w_lidar, ddepth, n_pix);
    return rcf_launch_status();
}

extern "C" int rcf_l1_loss_bwd(const float* depth, const float* gt, const float* lidar, const double* sums, const float* upstream,
                               float w_lidar, float* ddepth, long long n_pix, void* stream) {
    return rcf_masked_loss_bwd(depth, gt, lidar, sums, upstream, w_lidar, ddepth, n_pix, RCF_LOSS_L1, stream);
}

extern "C" int rcf_smoothness_loss_fwd(const float* image_nchw, const float* depth, float* workspace, double* sums, int n, int c, int h,
                                       int w, void* stream) {
    if (!image_nchw || !depth || !workspace || !sums || n <= 0 || c <= 0 || h <= 0 || w <= 0) return RCF_EINVAL;
    long long nb = ((long long)n * h * w + 255) / 256;
    if (nb > LOSS_BLOCKS) nb = LOSS_BLOCKS;
    hipLaunchKernelGGL(smoothness_partial_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, image_nchw, depth, workspace, n, c, h, w);
    int rc = rcf_launch_status();
    if (rc != RCF_OK) return rc;
    hipLaunchKernelGGL(l1_loss_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, (int)nb, sums);
    return rcf_launch_status();
}

extern "C" int rcf_smoothness_loss_bwd(const float* image_nchw, const float* depth, const double* sums, const float* upstream,
                                       float w_smoothness, float* ddepth, int n, int c, int h, int w, void* stream) {
    if (!image_nchw || !depth || !sums || !ddepth || n <= 0 || c <= 0 || h <= 0 || w <= 0) return RCF_EINVAL;
    unsigned b = nblk((long long)n * h * w, 256); if (b > 8192) b = 8192;
    hipLaunchKernelGGL(smoothness_bwd_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, image_nchw, depth, sums, upstream, w_smoothness,
                       ddepth, n, c, h, w);
    return rcf_launch_status();
}

extern "C" int rcf_outlier_removal(const float* depth, float* out, float* scratch, int n, int h, int w, int kernel_size,
                                   float threshold, void* stream) {
    if (!depth || !out || !scratch || n <= 0 || h <= 0 || w <= 0) return RCF_EINVAL;
    if (kernel_size < 1 || kernel_size > 15 || (kernel_size & 1) == 0) return RCF_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(scratch, 0, sizeof(float), st) != hipSuccess) return rcf_launch_status();
    const long long total = (long long)n * h * w;
    unsigned b = nblk(total, 256); if (b > 2048) b = 2048;
    hipLaunchKernelGGL(max_reduce_kernel, dim3(b), dim3(256), 0, st, depth, total, reinterpret_cast<unsigned*>(scratch));
    int rc = rcf_launch_status();
    if (rc != RCF_OK) return rc;
    const int r = kernel_size / 2;
    const size_t lds = (size_t)(32 + 2 * r) * (8 + 2 * r) * sizeof(float);
    hipLaunchKernelGGL(outlier_removal_kernel, dim3((w + 31) / 32, (h + 7) / 8, n), dim3(256), lds, st, depth, out, scratch, h, w,
                       kernel_size, threshold);
    return rcf_launch_status();
}

extern "C" int rcf_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step <= 0) return RCF_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    unsigned b = nblk(n, 256); if (b > 8192) b = 8192;
    hipLaunchKernelGGL(adam_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                       step_size, inv_bc2_sqrt);
    return rcf_launch_status();
}

extern "C" int rcf_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, float* state, void* stream) {
    if (!p || !g || !m || !v || !state || n <= 0) return RCF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, st, state);
    unsigned b = nblk(n, 256); if (b > 8192) b = 8192;
    hipLaunchKernelGGL(adam_dev_kernel, dim3(b), dim3(256), 0, st, p, g, m, v, n, state);
    return rcf_launch_status();
}

extern "C" int rcf_nchw_to_nhwc(const float* in, float* out, int n, int c, int h, int w, void* stream) {
    if (!in || !out || n <= 0 || c <= 0 || h <= 0 || w <= 0) return RCF_EINVAL;
    const long long total = (long long)n * c * h * w;
    unsigned b = nblk(total, 256); if (b > 8192) b = 8192;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, in, out, n, c, (long long)h * w);
    return rcf_launch_status();
}

extern "C" int rcf_nhwc_to_nchw(const float* in, float* out, int n, int c, int h, int w, void* stream) {
    if (!in || !out || n <= 0 || c <= 0 || h <= 0 || w <= 0) return RCF_EINVAL;
    const long long total = (long long)n * c * h * w;
    unsigned b = nblk(total, 256); if (b > 8192) b = 8192;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, in, out, n, c, (long long)h * w);
    return rcf_launch_status();
}

extern "C" int rcf_radar_scatter(const float* crops, const float* points, int k, int h, int w, int wc, int strict_reference,
                                 float* depth, float* response, void* stream) {
    if (!crops || !points || !depth || !response || k <= 0 || h <= 0 || w <= 0 || wc <= 0 || (wc & 1)) return RCF_EINVAL;
    const long long total = (long long)h * w;
    unsigned b = nblk(total, 256); if (b > 8192) b = 8192;
    hipLaunchKernelGGL(radar_scatter_kernel<false>, dim3(b), dim3(256), 0, (hipStream_t)stream, crops, points, k, h, w, h, wc,
                       strict_reference, depth, response);
    return rcf_launch_status();
}

extern "C" int rcf_radar_scatter_logits(const float* logits, const float* points, int k, int h, int w, int wc, int strict_reference,
                                        float* depth, float* response, void* stream) {
    if (!logits || !points || !depth || !response || k <= 0 || h <= 0 || w <= 0 || wc <= 0 || (wc & 1)) return RCF_EINVAL;
    const long long total = (long long)h * w;
    unsigned b = nblk(total, 256); if (b > 8192) b = 8192;
    hipLaunchKernelGGL(radar_scatter_kernel<true>, dim3(b), dim3(256), 0, (hipStream_t)stream, logits, points, k, h, w, h, wc,
                       strict_reference, depth, response);
    return rcf_launch_status();
}

extern "C" const char* rcf_version(void) { return "rcf_hip 0.3.0 (gfx950; fp32 results from v_mfma_f32_32x32x16_f16 on two scaled fp16 operand planes, or _bf16 on an exact 3-plane split; v_mfma_f32_32x32x2_f32 for 1x1 / stems; bf16 storage mode; weight gradients with producer / consumer waves and ds_read_b64_tr_b16)"; }

extern "C" int rcf_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
    const char* arch = prop.gcnArchName;
    return (arch[0] == 'g' && arch[1] == 'f' && arch[2] == 'x' && arch[3] == '9' && arch[4] == '5' && arch[5] == '0') ? 1 : 0;
}

// ---- BatchNorm folding for inference: w[o][...] * scale[o] ---------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) scale_channels_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                            float* __restrict__ out, int n_out, int inner) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)n_out * inner) return;
    out[i] = w[i] * scale[i / inner];
}
}   // namespace

// ---- max|x| of a tensor (the per-tensor scale of the two-plane fp16 convolutions; rcf_common.h) ------------------------------------
namespace {
// each block takes <= AMAX_SLICE elements: 16-B loads where the slice is aligned, one atomic per wave at the end
constexpr long long AMAX_SLICE = 16384;
__device__ __forceinline__ void amax_slice(const float* __restrict__ x, long long n, long long blk, float* __restrict__ amax) {
    const long long lo = blk * AMAX_SLICE;
    const long long hi = lo + AMAX_SLICE < n ? lo + AMAX_SLICE : n;
    float m = 0.f;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const long long hi4 = lo + ((hi - lo) & ~3ll);
        for (long long i = lo + 4 * threadIdx.x; i < hi4; i += 1024) m = rcf_amax4(m, *reinterpret_cast<const f32x4*>(x + i));
        for (long long i = hi4 + threadIdx.x; i < hi; i += 256) m = fmaxf(m, rcf_abs_finite(x[i]));
    } else {
        for (long long i = lo + threadIdx.x; i < hi; i += 256) m = fmaxf(m, rcf_abs_finite(x[i]));
    }
    rcf_amax_commit(m, amax);
}
__global__ void __launch_bounds__(256) amax_kernel(const float* __restrict__ x, long long n, float* __restrict__ amax) {
    amax_slice(x, n, blockIdx.x, amax);
}
struct AmaxArgs { const float* x; float* amax; long long n; };
constexpr int AMAX_BATCH = 128;
struct AmaxBatch {
    int n;
    unsigned blk_start[AMAX_BATCH + 1];
    AmaxArgs it[AMAX_BATCH];
};
static_assert(sizeof(AmaxBatch) <= 4096, "the batch travels as kernel arguments");
__global__ void __launch_bounds__(256) amax_batch_kernel(AmaxBatch b) {
    int lo = 0, hi = b.n - 1;   // the item of this block: last i with blk_start[i] <= blockIdx.x (wave-uniform binary search)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (b.blk_start[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    amax_slice(b.it[lo].x, b.it[lo].n, blockIdx.x - b.blk_start[lo], b.it[lo].amax);
}
}   // namespace

extern "C" int rcf_amax(const float* x, long long n, float* amax, void* stream) {
    if (!x || !amax || n <= 0) return RCF_EINVAL;
    hipLaunchKernelGGL(amax_kernel, dim3((unsigned)((n + AMAX_SLICE - 1) / AMAX_SLICE)), dim3(256), 0, (hipStream_t)stream, x, n, amax);
    return rcf_launch_status();
}

extern "C" int rcf_amax_batch(const rcf_amax_item* items, int n, void* stream) {
    if (!items || n <= 0) return RCF_EINVAL;
    for (int i = 0; i < n; ++i)
        if (!items[i].x || !items[i].amax || items[i].n <= 0) return RCF_EINVAL;
    static thread_local AmaxBatch b;   // ~4 KB: not on a ctypes caller's stack; one per host thread (two models on two threads do not race)
    int i = 0;
    while (i < n) {
        b.n = 0;
        unsigned nblk = 0;
        while (i < n && b.n < AMAX_BATCH) {
            b.it[b.n].x = items[i].x; b.it[b.n].amax = items[i].amax; b.it[b.n].n = items[i].n;
            b.blk_start[b.n] = nblk;
            nblk += (unsigned)((items[i].n + AMAX_SLICE - 1) / AMAX_SLICE);
            ++b.n;
            ++i;
        }
        for (int j = b.n; j <= AMAX_BATCH; ++j) b.blk_start[j] = nblk;
        hipLaunchKernelGGL(amax_batch_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, b);
    }
    return rcf_launch_status();
}

extern "C" int rcf_scale_channels(const float* w, const float* scale, float* out, int n_out, int inner, void* stream) {
    if (!w || !scale || !out || n_out <= 0 || inner <= 0) return RCF_EINVAL;
    const long long total = (long long)n_out * inner;
    hipLaunchKernelGGL(scale_channels_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, scale, out,
                       n_out, inner);
    return rcf_launch_status();
}

// ---- the 7x7 stride-2 stems as 4x4 stride-1 convolutions on the space-to-depth image (bf16 tensors) -------------------------------
// out[oy] = sum_ky W[ky] x[2 oy + ky - 3]; with ky' = ky + 1: 2 oy + ky' - 4 = 2 (oy + (ky' >> 1) - 2) + (ky' & 1), i.e. row
// oy + t - 2 (t = ky' >> 1 in 0..3) of row phase a = ky' & 1 -- a 4-tap kernel with pad 2 on the image whose channels are the four
// (row, column) phases of a pixel quad: S[n][y][x][a * 8 + b * 4 + c] = img[n][c][2 y + a][2 x + b] (16 channels, c < 4, zero padded).
namespace {
__global__ void __launch_bounds__(256) s2d_image_kernel(const float* __restrict__ img, unsigned short* __restrict__ out, int n, int c,
                                                        int h, int w, int hs, int ws) {
    const long long total = (long long)n * hs * ws * 4;   // one thread per (pixel quad, phase): 4 channels = 8 bytes
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const int ph = (int)(g & 3);
        const long long q = g >> 2;
        const int x = (int)(q % ws);
        const int y = (int)((q / ws) % hs);
        const int im = (int)(q / ((long long)ws * hs));
        const int iy = 2 * y + (ph >> 1), ix = 2 * x + (ph & 1);
        unsigned v[4] = {0u, 0u, 0u, 0u};
        if (iy < h && ix < w) {
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
                if (ch < c) v[ch] = rcf_f2b(img[(((size_t)im * c + ch) * h + iy) * w + ix]);
        }
        rcf_u32x2 pk;
        pk[0] = v[0] | (v[1] << 16);
        pk[1] = v[2] | (v[3] << 16);
        *reinterpret_cast<rcf_u32x2*>(out + (size_t)q * 16 + ph * 4) = pk;
    }
}

// the same image in fp32 (the two-plane fp16 arithmetic of the fp32 configuration), accumulating max|pixel| for its scale
__global__ void __launch_bounds__(256) s2d_image_f32_kernel(const float* __restrict__ img, float* __restrict__ out, int n, int c, int h, int w,
                                                            int hs, int ws, float* __restrict__ amax) {
    const long long total = (long long)n * hs * ws * 4;   // one thread per (pixel quad, phase): 4 channels = 16 bytes
    float am = 0.f;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const int ph = (int)(g & 3);
        const long long q = g >> 2;
        const int x = (int)(q % ws);
        const int y = (int)((q / ws) % hs);
        const int im = (int)(q / ((long long)ws * hs));
        const int iy = 2 * y + (ph >> 1), ix = 2 * x + (ph & 1);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy < h && ix < w) {
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
                if (ch < c) v[ch] = img[(((size_t)im * c + ch) * h + iy) * w + ix];
        }
        *reinterpret_cast<f32x4*>(out + (size_t)q * 16 + ph * 4) = v;
        am = rcf_amax4(am, v);
    }
    if (amax != nullptr) rcf_amax_commit(am, amax);
}

// W4[co][a * 8 + b * 4 + c][t][u] = W7[co][c][2 t + a - 1][2 u + b - 1] (0 outside the 7x7 kernel or for c >= C)
__global__ void __launch_bounds__(256) stem_weights_s2d_kernel(const float* __restrict__ w7, float* __restrict__ w4, int co_n, int c) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= co_n * 16 * 16) return;
    const int u = idx & 3, t = (idx >> 2) & 3, ci = (idx >> 4) & 15, co = idx >> 8;
    const int a = ci >> 3, b = (ci >> 2) & 1, ch = ci & 3;
    const int ky = 2 * t + a - 1, kx = 2 * u + b - 1;
    float v = 0.f;
    if (ch < c && ky >= 0 && ky < 7 && kx >= 0 && kx < 7) v = w7[(((size_t)co * c + ch) * 7 + ky) * 7 + kx];
    w4[idx] = v;
}
}   // namespace

extern "C" int rcf_s2d_image_b16(const float* img_nchw, void* out, int n, int c, int h, int w, void* stream) {
    if (!img_nchw || !out || n <= 0 || c <= 0 || h <= 0 || w <= 0) return RCF_EINVAL;
    if (c > 4) return RCF_EUNSUPPORTED;
    const int hs = (h + 1) / 2, ws = (w + 1) / 2;
    unsigned b = nblk((long long)n * hs * ws * 4, 256); if (b > 16384) b = 16384;
    hipLaunchKernelGGL(s2d_image_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, img_nchw, (unsigned short*)out, n, c, h, w, hs, ws);
    return rcf_launch_status();
}

extern "C" int rcf_s2d_image_f32(const float* img_nchw, float* out, int n, int c, int h, int w, float* amax, void* stream) {
    if (!img_nchw || !out || n <= 0 || c <= 0 || h <= 0 || w <= 0) return RCF_EINVAL;
    if (c > 4) return RCF_EUNSUPPORTED;
    const int hs = (h + 1) / 2, ws = (w + 1) / 2;
    unsigned b = nblk((long long)n * hs * ws * 4, 256); if (b > 2048) b = 2048;   // one atomic per block when amax is given
    hipLaunchKernelGGL(s2d_image_f32_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, img_nchw, out, n, c, h, w, hs, ws, amax);
    return rcf_launch_status();
}

extern "C" int rcf_stem_weights_s2d(const float* w7_oihw, float* w4_oihw, int c_out, int c_in, void* stream) {
    if (!w7_oihw || !w4_oihw || c_out <= 0 || c_in <= 0) return RCF_EINVAL;
    if (c_in > 4) return RCF_EUNSUPPORTED;
    hipLaunchKernelGGL(stem_weights_s2d_kernel, dim3((c_out * 256 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w7_oihw, w4_oihw, c_out, c_in);
    return rcf_launch_status();
}

// ---- element type conversion between the two storages (fp32 <-> bf16, round to nearest even), optionally accumulating into dst
namespace {
template <class SS, class SD>
__global__ void __launch_bounds__(256) convert_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n, int acc) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = rcf_ld1<SS>(src, (size_t)i);
        if (acc) v += rcf_ld1<SD>(dst, (size_t)i);
        rcf_st1<SD>(dst, (size_t)i, v);
    }
}
}   // namespace

extern "C" int rcf_convert(const void* src, int src_storage, void* dst, int dst_storage, long long n, int accumulate, void* stream) {
    if (!src || !dst || n <= 0) return RCF_EINVAL;
    if ((src_storage != RCF_STORE_FP32 && src_storage != RCF_STORE_BF16) || (dst_storage != RCF_STORE_FP32 && dst_storage != RCF_STORE_BF16))
        return RCF_EINVAL;
    unsigned b = nblk(n, 256); if (b > 16384) b = 16384;
    hipStream_t st = (hipStream_t)stream;
    const float* s = (const float*)src;
    float* d = (float*)dst;
    if (src_storage == RCF_STORE_FP32 && dst_storage == RCF_STORE_BF16) hipLaunchKernelGGL((convert_kernel<StF32, StB16>), dim3(b), dim3(256), 0, st, s, d, n, accumulate);
    else if (src_storage == RCF_STORE_BF16 && dst_storage == RCF_STORE_FP32) hipLaunchKernelGGL((convert_kernel<StB16, StF32>), dim3(b), dim3(256), 0, st, s, d, n, accumulate);
    else if (src_storage == RCF_STORE_BF16) hipLaunchKernelGGL((convert_kernel<StB16, StB16>), dim3(b), dim3(256), 0, st, s, d, n, accumulate);
    else hipLaunchKernelGGL((convert_kernel<StF32, StF32>), dim3(b), dim3(256), 0, st, s, d, n, accumulate);
    return rcf_launch_status();
}

// ---- exported instances: NAME for fp32 NHWC tensors, NAME_b16 for bf16 NHWC tensors (same argument lists; see include/rcf_hip.h)
extern "C" int rcf_maxpool3x3s2_fwd(const float* in, float* out, unsigned char* idx, int n, int h, int w, int c, void* stream) { return maxpool3x3s2_fwd_impl<StF32>(in, out, idx, n, h, w, c, stream); }
extern "C" int rcf_maxpool3x3s2_fwd_b16(const float* in, float* out, unsigned char* idx, int n, int h, int w, int c, void* stream) { return maxpool3x3s2_fwd_impl<StB16>(in, out, idx, n, h, w, c, stream); }
extern "C" int rcf_maxpool3x3s2_bwd(const float* dout, const unsigned char* idx, float* din, int din_accumulate, int n, int h,
                                    int w, int c, void* stream) { return maxpool3x3s2_bwd_impl<StF32>(dout, idx, din, din_accumulate, n, h, w, c, stream); }
extern "C" int rcf_maxpool3x3s2_bwd_b16(const float* dout, const unsigned char* idx, float* din, int din_accumulate, int n, int h,
                                    int w, int c, void* stream) { return maxpool3x3s2_bwd_impl<StB16>(dout, idx, din, din_accumulate, n, h, w, c, stream); }
extern "C" int rcf_upsample_nearest_bwd(const float* dup, float* dsrc, int dsrc_accumulate, int n, int h_up, int w_up,
                                        int h_src, int w_src, int c, void* stream) { return upsample_nearest_bwd_impl<StF32>(dup, dsrc, dsrc_accumulate, n, h_up, w_up, h_src, w_src, c, stream); }
extern "C" int rcf_upsample_nearest_bwd_b16(const float* dup, float* dsrc, int dsrc_accumulate, int n, int h_up, int w_up,
                                        int h_src, int w_src, int c, void* stream) { return upsample_nearest_bwd_impl<StB16>(dup, dsrc, dsrc_accumulate, n, h_up, w_up, h_src, w_src, c, stream); }
extern "C" int rcf_head_fwd(const float* x, const float* w, float* logit, float* depth, int n, int h, int w_, int c,
                            float min_depth, float max_depth, void* stream) { return head_fwd_impl<StF32>(x, w, logit, depth, n, h, w_, c, min_depth, max_depth, stream); }
template <int KST, int NT, int MT>
static int launch_fuse_wp(const float* d, const float* w1, const float* coef_w, const float* w2, const float* coef_p, const float* img,
                          float* out, long long n_pix, int c_i, void* stream) {
    const int gy = (c_i + 32 * NT - 1) / (32 * NT);
    const long long trips = (n_pix + 32 * 4 * MT - 1) / (32 * 4 * MT);
    static const int ncu = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            return (int)prop.multiProcessorCount;
        return 256;
    }();
    long long gx = (long long)ncu * 4 / gy;   // two workgroups per CU, two trips' worth of slack (pw_grid of the 1x1 kernel)
    if (gx > trips) gx = trips;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL((fuse_wp_b16_kernel<KST, NT, MT>), dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned short*>(d), w1, coef_w, w2, coef_p, reinterpret_cast<const unsigned short*>(img),
                       reinterpret_cast<unsigned short*>(out), n_pix, c_i);
    return rcf_launch_status();
}

extern "C" int rcf_fuse_wp_infer_supported(int c_d, int c_i) {
    return (c_d == 16 || c_d == 32 || c_d == 64 || c_d == 128) && c_i >= 2 && c_i % 2 == 0 ? 1 : 0;
}

extern "C" int rcf_fuse_wp_infer_b16(const float* d, const float* w1, const float* coef_w, const float* w2, const float* coef_p,
                                     const float* img, float* out, long long n_pix, int c_d, int c_i, void* stream) {
    if (!d || !w1 || !coef_w || !w2 || !coef_p || !img || !out || n_pix <= 0) return RCF_EINVAL;
    if (!rcf_fuse_wp_infer_supported(c_d, c_i)) return RCF_EUNSUPPORTED;
    switch (c_d) {
        case 16: return launch_fuse_wp<1, 1, 4>(d, w1, coef_w, w2, coef_p, img, out, n_pix, c_i, stream);
        case 32: return launch_fuse_wp<2, 2, 1>(d, w1, coef_w, w2, coef_p, img, out, n_pix, c_i, stream);
        case 64: return launch_fuse_wp<4, 2, 1>(d, w1, coef_w, w2, coef_p, img, out, n_pix, c_i, stream);
        default: return launch_fuse_wp<8, 1, 1>(d, w1, coef_w, w2, coef_p, img, out, n_pix, c_i, stream);
    }
}

extern "C" int rcf_head_fwd_b16(const float* x, const float* w, float* logit, float* depth, int n, int h, int w_, int c,
                            float min_depth, float max_depth, void* stream) { return head_fwd_impl<StB16>(x, w, logit, depth, n, h, w_, c, min_depth, max_depth, stream); }
extern "C" int rcf_head_fwd_bn(const float* z, const float* coef, const float* w, float* logit, float* depth, int n, int h, int w_,
                               int c, float min_depth, float max_depth, void* stream) { return head_fwd_bn_impl<StF32>(z, coef, w, logit, depth, n, h, w_, c, min_depth, max_depth, stream); }
extern "C" int rcf_head_fwd_bn_b16(const float* z, const float* coef, const float* w, float* logit, float* depth, int n, int h, int w_,
                               int c, float min_depth, float max_depth, void* stream) { return head_fwd_bn_impl<StB16>(z, coef, w, logit, depth, n, h, w_, c, min_depth, max_depth, stream); }
extern "C" int rcf_head_bwd_dgrad(const float* dlogit, const float* w, float* dx, int n, int h, int w_, int c, void* stream) { return head_bwd_dgrad_impl<StF32>(dlogit, w, dx, n, h, w_, c, stream); }
extern "C" int rcf_head_bwd_dgrad_b16(const float* dlogit, const float* w, float* dx, int n, int h, int w_, int c, void* stream) { return head_bwd_dgrad_impl<StB16>(dlogit, w, dx, n, h, w_, c, stream); }
extern "C" int rcf_head_bwd_wgrad(const float* x, const float* dlogit, float* dw, float* workspace, int n, int h, int w_, int c,
                                  void* stream) { return head_bwd_wgrad_impl<StF32>(x, dlogit, dw, workspace, n, h, w_, c, stream); }
extern "C" int rcf_head_bwd_wgrad_b16(const float* x, const float* dlogit, float* dw, float* workspace, int n, int h, int w_, int c,
                                  void* stream) { return head_bwd_wgrad_impl<StB16>(x, dlogit, dw, workspace, n, h, w_, c, stream); }
extern "C" int rcf_head_bwd_wgrad_bn(const float* z, const float* coef, const float* dlogit, float* dw, float* workspace, int n,
                                     int h, int w_, int c, void* stream) { return head_bwd_wgrad_bn_impl<StF32>(z, coef, dlogit, dw, workspace, n, h, w_, c, stream); }
extern "C" int rcf_head_bwd_wgrad_bn_b16(const float* z, const float* coef, const float* dlogit, float* dw, float* workspace, int n,
                                     int h, int w_, int c, void* stream) { return head_bwd_wgrad_bn_impl<StB16>(z, coef, dlogit, dw, workspace, n, h, w_, c, stream); }
