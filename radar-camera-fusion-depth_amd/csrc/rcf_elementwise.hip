// BatchNorm2d (training + eval), LeakyReLU, the ResNet residual tail and the 'weight_and_project' fusion of
// FusionNet, forward and backward, for NHWC fp32 tensors on gfx950.
//
// Replaces, on the reference's hot path: torch.nn.BatchNorm2d + LeakyReLU inside net_utils.Conv2d.forward
// (src/net_utils.py:84-91), the tail of ResNetBlock.forward (src/net_utils.py:309-323), the fusion at
// src/networks.py:863-866 and their autograd backward.
//
// All kernels are HBM-bound streams.  Thread t of a 256-thread block owns channel group (t % C4) (4 channels,
// one 16-byte access) for every pixel it visits, so per-channel coefficients live in registers and the
// per-channel reductions of the backward are in-register sums -> one LDS pass per block -> one partial row per
// block (deterministic; the tiny cross-block sum runs in fp64 in the finalize kernels).
#include "rcf_common.h"

namespace {

constexpr int EW_MAX_BLOCKS = 2048;
constexpr int EW_U = 4;   // pixels per loop trip (independent loads in flight per thread)

__host__ __device__ inline bool c4_ok(int c) {
    if (c < 4 || (c & 3)) return false;
    const int c4 = c >> 2;
    return (c4 & (c4 - 1)) == 0 && c4 <= 256;
}

inline int ew_blocks(long long n_pix, int c) {
    const int ppb = 256 / (c >> 2);
    long long b = (n_pix + ppb - 1) / ppb;
    if (b > EW_MAX_BLOCKS) b = EW_MAX_BLOCKS;
    if (b < 1) b = 1;
    return (int)b;
}

struct Coef4 {
    f32x4 scale, shift, mean, invstd;
};

__device__ __forceinline__ Coef4 load_coef(const float* __restrict__ coef, int c, int cg) {
    Coef4 k;
    k.scale = *reinterpret_cast<const f32x4*>(coef + 0 * c + cg * 4);
    k.shift = *reinterpret_cast<const f32x4*>(coef + 1 * c + cg * 4);
    k.mean = *reinterpret_cast<const f32x4*>(coef + 2 * c + cg * 4);
    k.invstd = *reinterpret_cast<const f32x4*>(coef + 3 * c + cg * 4);
    return k;
}

// Activation tensors are touched only through these two (element index i, 4 channels): S = StF32 / StB16 is their storage.
// streams larger than the 256 MB MALL: non-temporal loads measured -5..8 % on the BN kernels, non-temporal stores nothing
template <class S>
__device__ __forceinline__ f32x4 ld4(const float* p, size_t i) { return rcf_ld4_nt<S>(p, i); }
template <class S>
__device__ __forceinline__ void st4(float* p, size_t i, f32x4 v) { rcf_st4<S>(p, i, v); }

// ---------------------------------------------------------------- forward
// AM: also accumulate max|value written| into *amax (rcf_common.h: the per-tensor maximum of the two-plane fp16 convolutions)
template <class S, bool AM = false>
__global__ void __launch_bounds__(256) bn_act_fwd_kernel(const float* __restrict__ z, const float* __restrict__ coef,
                                                         const float* __restrict__ res, float* __restrict__ out,
                                                         long long n_pix, int c, int act, float* __restrict__ amax = nullptr) {
    float am = 0.f;
    const int c4n = c >> 2;
    const int cg = threadIdx.x % c4n;
    const int pl = threadIdx.x / c4n;
    const int ppb = 256 / c4n;
    const Coef4 k = load_coef(coef, c, cg);
    // EW_U pixels per trip: all their loads are issued before the first use (one 16-B load in flight per thread caps the stream at
    // ~4.4 TB/s: 2048 threads/CU x 16 B against ~2 us of HBM latency)
    const long long stride = (long long)gridDim.x * ppb;
    auto one = [&](f32x4 zz, f32x4 r, size_t i) {
        f32x4 y = zz * k.scale + k.shift;
        if (act == RCF_ACT_LEAKY_RELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = rcf_lrelu(y[j]);
        }
        if (res != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = rcf_lrelu(y[j] + r[j]);
        }
        st4<S>(out, i, y);
        if (AM) am = rcf_amax4(am, y);
    };
    long long p = (long long)blockIdx.x * ppb + pl;
    for (; p + (EW_U - 1) * stride < n_pix; p += EW_U * stride) {
        f32x4 zz[EW_U], rr[EW_U];
#pragma unroll
        for (int u = 0; u < EW_U; ++u) {
            const size_t i = (size_t)(p + u * stride) * c + cg * 4;
            zz[u] = ld4<S>(z, i);
            rr[u] = res != nullptr ? ld4<S>(res, i) : zz[u];
        }
#pragma unroll
        for (int u = 0; u < EW_U; ++u) one(zz[u], rr[u], (size_t)(p + u * stride) * c + cg * 4);
    }
    for (; p < n_pix; p += stride) {
        const size_t i = (size_t)p * c + cg * 4;
        const f32x4 zz = ld4<S>(z, i);
        one(zz, res != nullptr ? ld4<S>(res, i) : zz, i);
    }
    if (AM) rcf_amax_commit(am, amax);
}

template <class S, bool AM = false>
__global__ void __launch_bounds__(256) fuse_fwd_kernel(const float* __restrict__ zw, const float* __restrict__ coef_w,
                                                       const float* __restrict__ zp, const float* __restrict__ coef_p,
                                                       const float* __restrict__ img, float* __restrict__ out,
                                                       long long n_pix, int c, float* __restrict__ amax = nullptr) {
    float am = 0.f;
    const int c4n = c >> 2;
    const int cg = threadIdx.x % c4n;
    const int pl = threadIdx.x / c4n;
    const int ppb = 256 / c4n;
    const Coef4 kw = load_coef(coef_w, c, cg);
    const Coef4 kp = load_coef(coef_p, c, cg);
    for (long long p = (long long)blockIdx.x * ppb + pl; p < n_pix; p += (long long)gridDim.x * ppb) {
        const size_t i = (size_t)p * c + cg * 4;
        const f32x4 yw = ld4<S>(zw, i) * kw.scale + kw.shift;
        const f32x4 yp = ld4<S>(zp, i) * kp.scale + kp.shift;
        const f32x4 im = ld4<S>(img, i);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (1.f / (1.f + expf(-yw[j]))) * yp[j] + im[j];
        st4<S>(out, i, o);
        if (AM) am = rcf_amax4(am, o);
    }
    if (AM) rcf_amax_commit(am, amax);
}

// ---------------------------------------------------------------- block reduction of NS per-channel sums
// Each thread holds s[NS][4] for its channel group; result row layout: partial[blk][NS][c].
template <int NS>
__device__ __forceinline__ void block_reduce_store(const double (&s)[NS][4], double* __restrict__ partial_row, int c,
                                                   int cg, int pl, double* sm) {
    const int c4n = c >> 2;
    const int ppb = 256 / c4n;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) sm[(pl * NS + q) * c + cg * 4 + j] = s[q][j];
    __syncthreads();
    for (int e = threadIdx.x; e < NS * c; e += 256) {
        double t = 0.0;
        for (int r = 0; r < ppb; ++r) t += sm[r * NS * c + e];
        partial_row[e] = t;
    }
}

template <class S>
__global__ void __launch_bounds__(256) bn_act_bwd_reduce_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                                const float* __restrict__ coef, const float* __restrict__ out,
                                                                double* __restrict__ partials, long long n_pix, int c, int act,
                                                                int has_res) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int c4n = c >> 2;
    const int cg = threadIdx.x % c4n;
    const int pl = threadIdx.x / c4n;
    const int ppb = 256 / c4n;
    const Coef4 k = load_coef(coef, c, cg);
    // fp64 accumulation: these sums cancel almost completely in-network (the upstream BatchNorm already made
    // the gradient mean-free); PyTorch's CPU BatchNorm also accumulates float tensors in double.
    double s[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    const long long stride = (long long)gridDim.x * ppb;
    auto one = [&](f32x4 g, f32x4 zz, f32x4 o) {
        if (has_res) {
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] *= rcf_lrelu_grad(o[j]);
        }
        const f32x4 y = zz * k.scale + k.shift;
        const f32x4 xh = (zz - k.mean) * k.invstd;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (act == RCF_ACT_LEAKY_RELU) g[j] *= rcf_lrelu_grad(y[j]);
            s[0][j] += (double)g[j];
            s[1][j] += (double)g[j] * (double)xh[j];
        }
    };
    long long p = (long long)blockIdx.x * ppb + pl;
    for (; p + (EW_U - 1) * stride < n_pix; p += EW_U * stride) {   // same pixel order per thread as the one-at-a-time loop
        f32x4 g[EW_U], zz[EW_U], o[EW_U];
#pragma unroll
        for (int u = 0; u < EW_U; ++u) {
            const size_t i = (size_t)(p + u * stride) * c + cg * 4;
            g[u] = ld4<S>(dout, i);
            zz[u] = ld4<S>(z, i);
            o[u] = has_res ? ld4<S>(out, i) : zz[u];
        }
#pragma unroll
        for (int u = 0; u < EW_U; ++u) one(g[u], zz[u], o[u]);
    }
    for (; p < n_pix; p += stride) {
        const size_t i = (size_t)p * c + cg * 4;
        const f32x4 zz = ld4<S>(z, i);
        one(ld4<S>(dout, i), zz, has_res ? ld4<S>(out, i) : zz);
    }
    block_reduce_store<2>(s, partials + (size_t)blockIdx.x * 2 * c, c, cg, pl, sm);
}

template <class S, bool AM = false>
__global__ void __launch_bounds__(256) bn_act_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                               const float* __restrict__ coef, const float* __restrict__ out,
                                                               const float* __restrict__ bcoef, float* __restrict__ dz,
                                                               float* __restrict__ dres, int dres_accumulate, long long n_pix,
                                                               int c, int act, int has_res, float* __restrict__ amax = nullptr) {
    float am = 0.f;
    const int c4n = c >> 2;
    const int cg = threadIdx.x % c4n;
    const int pl = threadIdx.x / c4n;
    const int ppb = 256 / c4n;
    const Coef4 k = load_coef(coef, c, cg);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bcoef + cg * 4);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(bcoef + c + cg * 4);
    const long long stride = (long long)gridDim.x * ppb;
    const bool want_dres = has_res && dres != nullptr;
    auto one = [&](f32x4 g, f32x4 zz, f32x4 o, f32x4 dold, size_t i) {
        if (has_res) {
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] *= rcf_lrelu_grad(o[j]);
            if (dres != nullptr) {
                f32x4 d = g;
                if (dres_accumulate) d += dold;
                st4<S>(dres, i, d);
            }
        }
        const f32x4 y = zz * k.scale + k.shift;
        const f32x4 xh = (zz - k.mean) * k.invstd;
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gj = g[j];
            if (act == RCF_ACT_LEAKY_RELU) gj *= rcf_lrelu_grad(y[j]);
            r[j] = k.scale[j] * (gj - b0[j] - xh[j] * b1[j]);
        }
        st4<S>(dz, i, r);
        if (AM) am = rcf_amax4(am, r);
    };
    long long p = (long long)blockIdx.x * ppb + pl;
    for (; p + (EW_U - 1) * stride < n_pix; p += EW_U * stride) {
        f32x4 g[EW_U], zz[EW_U], o[EW_U], dold[EW_U];
#pragma unroll
        for (int u = 0; u < EW_U; ++u) {
            const size_t i = (size_t)(p + u * stride) * c + cg * 4;
            g[u] = ld4<S>(dout, i);
            zz[u] = ld4<S>(z, i);
            o[u] = has_res ? ld4<S>(out, i) : zz[u];
            dold[u] = (want_dres && dres_accumulate) ? ld4<S>(dres, i) : zz[u];
        }
#pragma unroll
        for (int u = 0; u < EW_U; ++u) one(g[u], zz[u], o[u], dold[u], (size_t)(p + u * stride) * c + cg * 4);
    }
    for (; p < n_pix; p += stride) {
        const size_t i = (size_t)p * c + cg * 4;
        const f32x4 zz = ld4<S>(z, i);
        one(ld4<S>(dout, i), zz, has_res ? ld4<S>(out, i) : zz, (want_dres && dres_accumulate) ? ld4<S>(dres, i) : zz, i);
    }
    if (AM) rcf_amax_commit(am, amax);
}

// ---- BatchNorm backward of the layer that feeds the 3x3 C->1 output head, fused with the head's input gradient:
// dout[p][c] = sum_tap dlogit[p - (ky-1, kx-1)] * w_head[tap][c] is recomputed from the dlogit halo tile in LDS in both passes, so
// the full-resolution C-channel gradient tensor is never written or read (saves one write and two reads of it per step).
constexpr int HB_H = 8, HB_W = 32, HB_HX = HB_W + 2, HB_HY = HB_H + 2, HB_NP = HB_HX * HB_HY;

template <int C4N>
struct HeadBnTile {
    static constexpr int C = 4 * C4N;
    f32x4 k[9];
    int cg;
    __device__ __forceinline__ void init(const float* __restrict__ w_head, float* wl) {
        for (int i = threadIdx.x; i < 9 * C; i += 256) wl[(i % 9) * C + i / 9] = w_head[i];   // OIHW [1][c][3][3] -> [tap][c]
        __syncthreads();
        cg = threadIdx.x % C4N;
#pragma unroll
        for (int t = 0; t < 9; ++t) k[t] = *reinterpret_cast<const f32x4*>(wl + t * C + cg * 4);
    }
    static __device__ __forceinline__ void origin(int tile, int h, int w, int* img, int* oy0, int* ox0) {
        const int tiles_x = (w + HB_W - 1) / HB_W, tiles_y = (h + HB_H - 1) / HB_H;
        const int t2 = tile / tiles_x;
        *img = t2 / tiles_y;
        *oy0 = (t2 % tiles_y) * HB_H;
        *ox0 = (tile % tiles_x) * HB_W;
    }
    static __device__ __forceinline__ void load_halo(float* D, const float* __restrict__ dl, int img, int oy0, int ox0, int h, int w) {
        for (int hp = threadIdx.x; hp < HB_NP; hp += 256) {
            const int hy = hp / HB_HX, hx = hp - hy * HB_HX;
            const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
            D[hp] = (iy >= 0 && iy < h && ix >= 0 && ix < w) ? dl[((size_t)img * h + iy) * w + ix] : 0.f;
        }
    }
    __device__ __forceinline__ f32x4 dout(const float* D, int ty, int tx) const {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) s += D[(ty + 2 - ky) * HB_HX + tx + 2 - kx] * k[ky * 3 + kx];
        return s;
    }
};

template <int C4N, class S>
__global__ void __launch_bounds__(256) head_bn_bwd_reduce_kernel(const float* __restrict__ dlogit, const float* __restrict__ w_head,
                                                                 const float* __restrict__ z, const float* __restrict__ coef,
                                                                 double* __restrict__ partials, int n, int h, int w) {
    constexpr int C = 4 * C4N;
    extern __shared__ __attribute__((aligned(16))) double smd[];   // [256 / C4N][2][C] doubles for the final reduction
    __shared__ __attribute__((aligned(16))) float wl[9 * C];
    __shared__ float D[HB_NP];
    HeadBnTile<C4N> hb;
    hb.init(w_head, wl);
    const Coef4 kc = load_coef(coef, C, hb.cg);
    double s[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    const int ntiles = n * ((h + HB_H - 1) / HB_H) * ((w + HB_W - 1) / HB_W);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int img, oy0, ox0;
        HeadBnTile<C4N>::origin(tile, h, w, &img, &oy0, &ox0);
        __syncthreads();
        HeadBnTile<C4N>::load_halo(D, dlogit, img, oy0, ox0, h, w);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < C4N; ++it) {
            const int pix = (threadIdx.x + 256 * it) / C4N;
            const int ty = pix / HB_W, tx = pix % HB_W;
            const int oy = oy0 + ty, ox = ox0 + tx;
            if (oy < h && ox < w) {
                const f32x4 zz = ld4<S>(z, (((size_t)img * h + oy) * w + ox) * C + hb.cg * 4);
                const f32x4 g0 = hb.dout(D, ty, tx);
                const f32x4 y = zz * kc.scale + kc.shift;
                const f32x4 xh = (zz - kc.mean) * kc.invstd;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float g = g0[j] * rcf_lrelu_grad(y[j]);
                    s[0][j] += (double)g;
                    s[1][j] += (double)g * (double)xh[j];
                }
            }
        }
    }
    block_reduce_store<2>(s, partials + (size_t)blockIdx.x * 2 * C, C, hb.cg, threadIdx.x / C4N, smd);
}

template <int C4N, class S, bool AM = false>
__global__ void __launch_bounds__(256) head_bn_bwd_apply_kernel(const float* __restrict__ dlogit, const float* __restrict__ w_head,
                                                                const float* __restrict__ z, const float* __restrict__ coef,
                                                                const float* __restrict__ bcoef, float* __restrict__ dz, int n, int h,
                                                                int w, float* __restrict__ amax = nullptr) {
    float am = 0.f;
    constexpr int C = 4 * C4N;
    __shared__ __attribute__((aligned(16))) float wl[9 * C];
    __shared__ float D[HB_NP];
    HeadBnTile<C4N> hb;
    int img, oy0, ox0;
    HeadBnTile<C4N>::origin(blockIdx.x, h, w, &img, &oy0, &ox0);
    HeadBnTile<C4N>::load_halo(D, dlogit, img, oy0, ox0, h, w);
    hb.init(w_head, wl);   // contains the barrier that also publishes D
    const Coef4 kc = load_coef(coef, C, hb.cg);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bcoef + hb.cg * 4);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(bcoef + C + hb.cg * 4);
#pragma unroll
    for (int it = 0; it < C4N; ++it) {
        const int pix = (threadIdx.x + 256 * it) / C4N;
        const int ty = pix / HB_W, tx = pix % HB_W;
        const int oy = oy0 + ty, ox = ox0 + tx;
        if (oy < h && ox < w) {
            const size_t i = (((size_t)img * h + oy) * w + ox) * C + hb.cg * 4;
            const f32x4 zz = ld4<S>(z, i);
            const f32x4 g0 = hb.dout(D, ty, tx);
            const f32x4 y = zz * kc.scale + kc.shift;
            const f32x4 xh = (zz - kc.mean) * kc.invstd;
            f32x4 r;
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = kc.scale[j] * (g0[j] * rcf_lrelu_grad(y[j]) - b0[j] - xh[j] * b1[j]);
            st4<S>(dz, i, r);
            if (AM) am = rcf_amax4(am, r);
        }
    }
    if (AM) rcf_amax_commit(am, amax);
}

template <class S>
__global__ void __launch_bounds__(256) fuse_bwd_reduce_kernel(const float* __restrict__ dout, const float* __restrict__ zw,
                                                              const float* __restrict__ coef_w, const float* __restrict__ zp,
                                                              const float* __restrict__ coef_p, double* __restrict__ partials,
                                                              long long n_pix, int c) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int c4n = c >> 2;
    const int cg = threadIdx.x % c4n;
    const int pl = threadIdx.x / c4n;
    const int ppb = 256 / c4n;
    const Coef4 kw = load_coef(coef_w, c, cg);
    const Coef4 kp = load_coef(coef_p, c, cg);
    double s[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[q][j] = 0.0;
    for (long long p = (long long)blockIdx.x * ppb + pl; p < n_pix; p += (long long)gridDim.x * ppb) {
        const size_t i = (size_t)p * c + cg * 4;
        const f32x4 g = ld4<S>(dout, i);
        const f32x4 a = ld4<S>(zw, i);
        const f32x4 b = ld4<S>(zp, i);
        const f32x4 yw = a * kw.scale + kw.shift;
        const f32x4 yp = b * kp.scale + kp.shift;
        const f32x4 xw = (a - kw.mean) * kw.invstd;
        const f32x4 xp = (b - kp.mean) * kp.invstd;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sg = 1.f / (1.f + expf(-yw[j]));
            const float gw = g[j] * yp[j] * sg * (1.f - sg);
            const float gp = g[j] * sg;
            s[0][j] += (double)gw;
            s[1][j] += (double)gw * (double)xw[j];
            s[2][j] += (double)gp;
            s[3][j] += (double)gp * (double)xp[j];
        }
    }
    block_reduce_store<4>(s, partials + (size_t)blockIdx.x * 4 * c, c, cg, pl, sm);
}

template <class S>
__global__ void __launch_bounds__(256) fuse_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ zw,
                                                             const float* __restrict__ coef_w, const float* __restrict__ zp,
                                                             const float* __restrict__ coef_p, const float* __restrict__ bcw,
                                                             const float* __restrict__ bcp, float* __restrict__ dzw,
                                                             float* __restrict__ dzp, float* __restrict__ dimg,
                                                             int dimg_accumulate, long long n_pix, int c) {
    const int c4n = c >> 2;
    const int cg = threadIdx.x % c4n;
    const int pl = threadIdx.x / c4n;
    const int ppb = 256 / c4n;
    const Coef4 kw = load_coef(coef_w, c, cg);
    const Coef4 kp = load_coef(coef_p, c, cg);
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(bcw + cg * 4);
    const f32x4 w1 = *reinterpret_cast<const f32x4*>(bcw + c + cg * 4);
    const f32x4 p0 = *reinterpret_cast<const f32x4*>(bcp + cg * 4);
    const f32x4 p1 = *reinterpret_cast<const f32x4*>(bcp + c + cg * 4);
    for (long long p = (long long)blockIdx.x * ppb + pl; p < n_pix; p += (long long)gridDim.x * ppb) {
        const size_t i = (size_t)p * c + cg * 4;
        const f32x4 g = ld4<S>(dout, i);
        const f32x4 a = ld4<S>(zw, i);
        const f32x4 b = ld4<S>(zp, i);
        const f32x4 yw = a * kw.scale + kw.shift;
        const f32x4 yp = b * kp.scale + kp.shift;
        const f32x4 xw = (a - kw.mean) * kw.invstd;
        const f32x4 xp = (b - kp.mean) * kp.invstd;
        f32x4 rw, rp;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sg = 1.f / (1.f + expf(-yw[j]));
            const float gw = g[j] * yp[j] * sg * (1.f - sg);
            const float gp = g[j] * sg;
            rw[j] = kw.scale[j] * (gw - w0[j] - xw[j] * w1[j]);
            rp[j] = kp.scale[j] * (gp - p0[j] - xp[j] * p1[j]);
        }
        st4<S>(dzw, i, rw);
        st4<S>(dzp, i, rp);
        if (dimg != nullptr) {
            f32x4 d = g;
            if (dimg_accumulate) d += ld4<S>(dimg, i);
            st4<S>(dimg, i, d);
        }
    }
}

// ---------------------------------------------------------------- finalize kernels (one block per channel)
__device__ __forceinline__ double block_sum_double(double v, double* smd) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smd[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += smd[w];
    return t;
}

__global__ void __launch_bounds__(256) bn_finalize_kernel(const double* __restrict__ partials, int n_partials, int c,
                                                          double count, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float momentum, float eps,
                                                          int training, float* __restrict__ coef) {
    __shared__ double smd[8];
    const int ch = blockIdx.x;
    double mean, var;
    if (training) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
        for (int r = threadIdx.x; r < n_partials; r += blockDim.x) {   // (a launch of <= 768 rows: latency, not bandwidth)
            s1 += partials[((size_t)r * 2 + 0) * c + ch];
            s2 += partials[((size_t)r * 2 + 1) * c + ch];
        }
        s1 = block_sum_double(s1, smd);
        s2 = block_sum_double(s2, smd);
        mean = s1 / count;
        var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
    } else {
        mean = (double)running_mean[ch];
        var = (double)running_var[ch];
    }
    if (threadIdx.x == 0) {
        const double invstd = 1.0 / sqrt(var + (double)eps);
        const double scale = (double)gamma[ch] * invstd;
        coef[0 * c + ch] = (float)scale;
        coef[1 * c + ch] = (float)((double)beta[ch] - mean * scale);
        coef[2 * c + ch] = (float)mean;
        coef[3 * c + ch] = (float)invstd;
        if (training) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[ch] = (float)((1.0 - (double)momentum) * (double)running_mean[ch] + (double)momentum * mean);
            running_var[ch] = (float)((1.0 - (double)momentum) * (double)running_var[ch] + (double)momentum * unbiased);
        }
    }
}

__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(const double* __restrict__ partials, int n_blocks, int stride,
                                                              int c, double count, float* __restrict__ bcoef,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double smd[8];
    const int ch = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    // column reads (one channel of every partial row): 4 rows per trip so that 8 loads are in flight per thread
    int r = threadIdx.x;
    for (; r + 768 < n_blocks; r += 1024) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = partials[(size_t)(r + 256 * u) * stride + ch];
            b[u] = partials[(size_t)(r + 256 * u) * stride + c + ch];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s1 += a[u]; s2 += b[u]; }
    }
    for (; r < n_blocks; r += 256) {
        s1 += partials[(size_t)r * stride + ch];
        s2 += partials[(size_t)r * stride + c + ch];
    }
    s1 = block_sum_double(s1, smd);
    s2 = block_sum_double(s2, smd);
    if (threadIdx.x == 0) {
        bcoef[ch] = (float)(s1 / count);
        bcoef[c + ch] = (float)(s2 / count);
        dbeta[ch] = (float)s1;
        dgamma[ch] = (float)s2;
    }
}

}   // namespace

extern "C" int rcf_ew_blocks(long long n_pix, int c) {
    if (n_pix <= 0 || !c4_ok(c)) return RCF_EINVAL;
    return ew_blocks(n_pix, c);
}

extern "C" int rcf_bn_finalize(const double* partials, int n_partials, int c, double count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               int training, float* coef, void* stream) {
    if (c <= 0 || !gamma || !beta || !running_mean || !running_var || !coef) return RCF_EINVAL;
    if (training && (!partials || n_partials <= 0 || count <= 0.0)) return RCF_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(c), dim3(256), 0, (hipStream_t)stream, partials, n_partials, c, count, gamma,
                       beta, running_mean, running_var, momentum, eps, training, coef);
    return rcf_launch_status();
}

template <class S, bool AM = false>
static int bn_act_fwd_impl(const float* z, const float* coef, const float* res, float* out, long long n_pix, int c,
                              int act, void* stream, float* amax = nullptr) {
    if (!z || !coef || !out || n_pix <= 0 || (AM && !amax)) return RCF_EINVAL;
    if (!c4_ok(c)) return RCF_EUNSUPPORTED;
    hipLaunchKernelGGL((bn_act_fwd_kernel<S, AM>), dim3(ew_blocks(n_pix, c)), dim3(256), 0, (hipStream_t)stream, z, coef, res, out,
                       n_pix, c, act, amax);
    return rcf_launch_status();
}

template <class S, bool AM = false>
static int fuse_fwd_impl(const float* zw, const float* coef_w, const float* zp, const float* coef_p, const float* img,
                            float* out, long long n_pix, int c, void* stream, float* amax = nullptr) {
    if (!zw || !coef_w || !zp || !coef_p || !img || !out || n_pix <= 0 || (AM && !amax)) return RCF_EINVAL;
    if (!c4_ok(c)) return RCF_EUNSUPPORTED;
    hipLaunchKernelGGL((fuse_fwd_kernel<S, AM>), dim3(ew_blocks(n_pix, c)), dim3(256), 0, (hipStream_t)stream, zw, coef_w, zp, coef_p,
                       img, out, n_pix, c, amax);
    return rcf_launch_status();
}

template <class S>
static int bn_act_bwd_reduce_impl(const float* dout, const float* z, const float* coef, const float* out, double* partials,
                                     long long n_pix, int c, int act, int has_res, void* stream) {
    if (!dout || !z || !coef || !partials || n_pix <= 0 || (has_res && !out)) return RCF_EINVAL;
    if (!c4_ok(c)) return RCF_EUNSUPPORTED;
    const int ppb = 256 / (c >> 2);
    hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<S>), dim3(ew_blocks(n_pix, c)), dim3(256), (size_t)ppb * 2 * c * sizeof(double),
                       (hipStream_t)stream, dout, z, coef, out, partials, n_pix, c, act, has_res);
    return rcf_launch_status();
}

extern "C" int rcf_bn_bwd_finalize(const double* partials, int n_blocks, int partial_stride, int c, double count, float* bcoef,
                                   float* dgamma, float* dbeta, void* stream) {
    if (!partials || n_blocks <= 0 || c <= 0 || partial_stride < 2 * c || count <= 0.0 || !bcoef || !dgamma || !dbeta)
        return RCF_EINVAL;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(c), dim3(256), 0, (hipStream_t)stream, partials, n_blocks, partial_stride, c,
                       count, bcoef, dgamma, dbeta);
    return rcf_launch_status();
}

template <class S, bool AM = false>
static int bn_act_bwd_apply_impl(const float* dout, const float* z, const float* coef, const float* out, const float* bcoef,
                                    float* dz, float* dres, int dres_accumulate, long long n_pix, int c, int act, int has_res,
                                    void* stream, float* amax = nullptr) {
    if (!dout || !z || !coef || !bcoef || !dz || n_pix <= 0 || (has_res && !out) || (AM && !amax)) return RCF_EINVAL;
    if (!c4_ok(c)) return RCF_EUNSUPPORTED;
    hipLaunchKernelGGL((bn_act_bwd_apply_kernel<S, AM>), dim3(ew_blocks(n_pix, c)), dim3(256), 0, (hipStream_t)stream, dout, z, coef, out,
                       bcoef, dz, dres, dres_accumulate, n_pix, c, act, has_res, amax);
    return rcf_launch_status();
}

template <class S>
static int fuse_bwd_reduce_impl(const float* dout, const float* zw, const float* coef_w, const float* zp,
                                   const float* coef_p, double* partials, long long n_pix, int c, void* stream) {
    if (!dout || !zw || !coef_w || !zp || !coef_p || !partials || n_pix <= 0) return RCF_EINVAL;
    if (!c4_ok(c)) return RCF_EUNSUPPORTED;
    const int ppb = 256 / (c >> 2);
    hipLaunchKernelGGL((fuse_bwd_reduce_kernel<S>), dim3(ew_blocks(n_pix, c)), dim3(256), (size_t)ppb * 4 * c * sizeof(double),
                       (hipStream_t)stream, dout, zw, coef_w, zp, coef_p, partials, n_pix, c);
    return rcf_launch_status();
}

template <class S>
static int fuse_bwd_apply_impl(const float* dout, const float* zw, const float* coef_w, const float* zp, const float* coef_p,
                                  const float* bcoef_w, const float* bcoef_p, float* dzw, float* dzp, float* dimg,
                                  int dimg_accumulate, long long n_pix, int c, void* stream) {
    if (!dout || !zw || !coef_w || !zp || !coef_p || !bcoef_w || !bcoef_p || !dzw || !dzp || n_pix <= 0) return RCF_EINVAL;
    if (!c4_ok(c)) return RCF_EUNSUPPORTED;
    hipLaunchKernelGGL((fuse_bwd_apply_kernel<S>), dim3(ew_blocks(n_pix, c)), dim3(256), 0, (hipStream_t)stream, dout, zw, coef_w, zp,
                       coef_p, bcoef_w, bcoef_p, dzw, dzp, dimg, dimg_accumulate, n_pix, c);
    return rcf_launch_status();
}

extern "C" int rcf_head_bn_blocks(int n, int h, int w, int c) {
    if (n <= 0 || h <= 0 || w <= 0 || !c4_ok(c) || (c >> 2) > 16) return RCF_EINVAL;
    const long long nt = (long long)n * ((h + HB_H - 1) / HB_H) * ((w + HB_W - 1) / HB_W);
    return (int)(nt < EW_MAX_BLOCKS ? nt : EW_MAX_BLOCKS);
}

template <class S>
static int head_bn_bwd_reduce_impl(const float* dlogit, const float* w_head, const float* z, const float* coef, double* partials,
                                      int n, int h, int w, int c, void* stream) {
    if (!dlogit || !w_head || !z || !coef || !partials) return RCF_EINVAL;
    const int nb = rcf_head_bn_blocks(n, h, w, c);
    if (nb <= 0) return RCF_EUNSUPPORTED;
    const int c4n = c >> 2;
    const size_t lds = (size_t)(256 / c4n) * 2 * c * sizeof(double);
    hipStream_t st = (hipStream_t)stream;
#define RCF_HBR(N) hipLaunchKernelGGL((head_bn_bwd_reduce_kernel<N, S>), dim3(nb), dim3(256), lds, st, dlogit, w_head, z, coef, partials, n, h, w)
    switch (c4n) {
        case 1: RCF_HBR(1); break;
        case 2: RCF_HBR(2); break;
        case 4: RCF_HBR(4); break;
        case 8: RCF_HBR(8); break;
        default: RCF_HBR(16); break;
    }
#undef RCF_HBR
    return rcf_launch_status();
}

template <class S, bool AM = false>
static int head_bn_bwd_apply_impl(const float* dlogit, const float* w_head, const float* z, const float* coef, const float* bcoef,
                                     float* dz, int n, int h, int w, int c, void* stream, float* amax = nullptr) {
    if (!dlogit || !w_head || !z || !coef || !bcoef || !dz || (AM && !amax)) return RCF_EINVAL;
    if (rcf_head_bn_blocks(n, h, w, c) <= 0) return RCF_EUNSUPPORTED;
    const unsigned nt = (unsigned)n * ((h + HB_H - 1) / HB_H) * ((w + HB_W - 1) / HB_W);
    hipStream_t st = (hipStream_t)stream;
#define RCF_HBA(N) hipLaunchKernelGGL((head_bn_bwd_apply_kernel<N, S, AM>), dim3(nt), dim3(256), 0, st, dlogit, w_head, z, coef, bcoef, dz, n, h, w, amax)
    switch (c >> 2) {
        case 1: RCF_HBA(1); break;
        case 2: RCF_HBA(2); break;
        case 4: RCF_HBA(4); break;
        case 8: RCF_HBA(8); break;
        default: RCF_HBA(16); break;
    }
#undef RCF_HBA
    return rcf_launch_status();
}

// ---- exported instances: NAME for fp32 NHWC tensors, NAME_b16 for bf16 NHWC tensors (same argument lists; see include/rcf_hip.h)
extern "C" int rcf_bn_act_fwd(const float* z, const float* coef, const float* res, float* out, long long n_pix, int c,
                              int act, void* stream) { return bn_act_fwd_impl<StF32>(z, coef, res, out, n_pix, c, act, stream); }
extern "C" int rcf_bn_act_fwd_b16(const float* z, const float* coef, const float* res, float* out, long long n_pix, int c,
                              int act, void* stream) { return bn_act_fwd_impl<StB16>(z, coef, res, out, n_pix, c, act, stream); }
extern "C" int rcf_fuse_fwd(const float* zw, const float* coef_w, const float* zp, const float* coef_p, const float* img,
                            float* out, long long n_pix, int c, void* stream) { return fuse_fwd_impl<StF32>(zw, coef_w, zp, coef_p, img, out, n_pix, c, stream); }
extern "C" int rcf_fuse_fwd_b16(const float* zw, const float* coef_w, const float* zp, const float* coef_p, const float* img,
                            float* out, long long n_pix, int c, void* stream) { return fuse_fwd_impl<StB16>(zw, coef_w, zp, coef_p, img, out, n_pix, c, stream); }
extern "C" int rcf_bn_act_bwd_reduce(const float* dout, const float* z, const float* coef, const float* out, double* partials,
                                     long long n_pix, int c, int act, int has_res, void* stream) { return bn_act_bwd_reduce_impl<StF32>(dout, z, coef, out, partials, n_pix, c, act, has_res, stream); }
extern "C" int rcf_bn_act_bwd_reduce_b16(const float* dout, const float* z, const float* coef, const float* out, double* partials,
                                     long long n_pix, int c, int act, int has_res, void* stream) { return bn_act_bwd_reduce_impl<StB16>(dout, z, coef, out, partials, n_pix, c, act, has_res, stream); }
extern "C" int rcf_bn_act_bwd_apply(const float* dout, const float* z, const float* coef, const float* out, const float* bcoef,
                                    float* dz, float* dres, int dres_accumulate, long long n_pix, int c, int act, int has_res,
                                    void* stream) { return bn_act_bwd_apply_impl<StF32>(dout, z, coef, out, bcoef, dz, dres, dres_accumulate, n_pix, c, act, has_res, stream); }
extern "C" int rcf_bn_act_bwd_apply_b16(const float* dout, const float* z, const float* coef, const float* out, const float* bcoef,
                                    float* dz, float* dres, int dres_accumulate, long long n_pix, int c, int act, int has_res,
                                    void* stream) { return bn_act_bwd_apply_impl<StB16>(dout, z, coef, out, bcoef, dz, dres, dres_accumulate, n_pix, c, act, has_res, stream); }
extern "C" int rcf_fuse_bwd_reduce(const float* dout, const float* zw, const float* coef_w, const float* zp,
                                   const float* coef_p, double* partials, long long n_pix, int c, void* stream) { return fuse_bwd_reduce_impl<StF32>(dout, zw, coef_w, zp, coef_p, partials, n_pix, c, stream); }
extern "C" int rcf_fuse_bwd_reduce_b16(const float* dout, const float* zw, const float* coef_w, const float* zp,
                                   const float* coef_p, double* partials, long long n_pix, int c, void* stream) { return fuse_bwd_reduce_impl<StB16>(dout, zw, coef_w, zp, coef_p, partials, n_pix, c, stream); }
extern "C" int rcf_fuse_bwd_apply(const float* dout, const float* zw, const float* coef_w, const float* zp, const float* coef_p,
                                  const float* bcoef_w, const float* bcoef_p, float* dzw, float* dzp, float* dimg,
                                  int dimg_accumulate, long long n_pix, int c, void* stream) { return fuse_bwd_apply_impl<StF32>(dout, zw, coef_w, zp, coef_p, bcoef_w, bcoef_p, dzw, dzp, dimg, dimg_accumulate, n_pix, c, stream); }
extern "C" int rcf_fuse_bwd_apply_b16(const float* dout, const float* zw, const float* coef_w, const float* zp, const float* coef_p,
                                  const float* bcoef_w, const float* bcoef_p, float* dzw, float* dzp, float* dimg,
                                  int dimg_accumulate, long long n_pix, int c, void* stream) { return fuse_bwd_apply_impl<StB16>(dout, zw, coef_w, zp, coef_p, bcoef_w, bcoef_p, dzw, dzp, dimg, dimg_accumulate, n_pix, c, stream); }
extern "C" int rcf_head_bn_bwd_reduce(const float* dlogit, const float* w_head, const float* z, const float* coef, double* partials,
                                      int n, int h, int w, int c, void* stream) { return head_bn_bwd_reduce_impl<StF32>(dlogit, w_head, z, coef, partials, n, h, w, c, stream); }
extern "C" int rcf_head_bn_bwd_reduce_b16(const float* dlogit, const float* w_head, const float* z, const float* coef, double* partials,
                                      int n, int h, int w, int c, void* stream) { return head_bn_bwd_reduce_impl<StB16>(dlogit, w_head, z, coef, partials, n, h, w, c, stream); }
extern "C" int rcf_head_bn_bwd_apply(const float* dlogit, const float* w_head, const float* z, const float* coef, const float* bcoef,
                                     float* dz, int n, int h, int w, int c, void* stream) { return head_bn_bwd_apply_impl<StF32>(dlogit, w_head, z, coef, bcoef, dz, n, h, w, c, stream); }
extern "C" int rcf_head_bn_bwd_apply_b16(const float* dlogit, const float* w_head, const float* z, const float* coef, const float* bcoef,
                                     float* dz, int n, int h, int w, int c, void* stream) { return head_bn_bwd_apply_impl<StB16>(dlogit, w_head, z, coef, bcoef, dz, n, h, w, c, stream); }

// ---- the same kernels also accumulating max|value written| (fp32 tensors; include/rcf_hip.h: rcf_*_amax)
extern "C" int rcf_bn_act_fwd_amax(const float* z, const float* coef, const float* res, float* out, long long n_pix, int c, int act,
                                   float* amax, void* stream) { return bn_act_fwd_impl<StF32, true>(z, coef, res, out, n_pix, c, act, stream, amax); }
extern "C" int rcf_fuse_fwd_amax(const float* zw, const float* coef_w, const float* zp, const float* coef_p, const float* img, float* out,
                                 long long n_pix, int c, float* amax, void* stream) { return fuse_fwd_impl<StF32, true>(zw, coef_w, zp, coef_p, img, out, n_pix, c, stream, amax); }
extern "C" int rcf_bn_act_bwd_apply_amax(const float* dout, const float* z, const float* coef, const float* out, const float* bcoef,
                                         float* dz, float* dres, int dres_accumulate, long long n_pix, int c, int act, int has_res,
                                         float* amax, void* stream) { return bn_act_bwd_apply_impl<StF32, true>(dout, z, coef, out, bcoef, dz, dres, dres_accumulate, n_pix, c, act, has_res, stream, amax); }
extern "C" int rcf_head_bn_bwd_apply_amax(const float* dlogit, const float* w_head, const float* z, const float* coef, const float* bcoef,
                                          float* dz, int n, int h, int w, int c, float* amax, void* stream) { return head_bn_bwd_apply_impl<StF32, true>(dlogit, w_head, z, coef, bcoef, dz, n, h, w, c, stream, amax); }
