// Implicit-GEMM convolution for FusionNet on MI355X (gfx950): forward / input-gradient / weight-gradient.
//
// Replaces torch.nn.Conv2d (+ the F.interpolate and torch.cat feeding it) on the reference's hot path
// (src/net_utils.py:29-91, :156-198, :473-569) and autograd's conv backward behind loss.backward()
// (src/fusionnet_main.py:398).
//
// Design:
//  * fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 -- bit-for-bit an fmaf chain, so results differ
//    from the CPU reference only by summation order.
//  * A workgroup (4 waves, one per SIMD) owns a PX x TH tile of output pixels and BN output channels.  Per
//    chunk of CK input channels it stages the input HALO tile once into LDS ([halo pixel][CK+4], NHWC so each
//    pixel's channels are one contiguous 16-B-aligned run) and all taps of the weight panel ([tap][BN][CK+4]);
//    the 3x3 taps are then LDS address offsets, so every input element is fetched from HBM/L2 once per
//    chunk instead of once per tap.  The nearest-upsample / zero-insert / channel-concat of the decoder are
//    folded into that staging gather: the upsampled or concatenated tensor is never materialised.
//  * MFMA operand fetch: lane (i = l&31, h = l>>5) reads ONE ds_read_b128 = 4 consecutive k of row i starting
//    at k = 4h and feeds 4 consecutive MFMAs with it.  A and B use the same (h, q) -> k map, so the reduction
//    is merely re-ordered.  The +4 float row pad makes those reads bank-conflict free for stride-1 tiles.
//  * Epilogue: NHWC store (128 B contiguous per pixel row per 32 channels) and, fused, the per-channel sum and
//    sum of squares BatchNorm needs (in-lane adds -> one cross-half shuffle -> LDS across the 4 waves ->
//    one partial row per workgroup; no atomics, deterministic).
#include "rcf_common.h"

namespace {

struct ConvArgs {
    const float* in1;
    const float* in2;
    const float* wp;
    float* out;
    double* stats;
    const float* dz;   // wgrad only
    float* ws;         // wgrad only
    int n, h_in, w_in, c1, c2, h1, w1, gather1;
    int h_out, w_out, c_out, pad, stride, gstep, accumulate;
    float sy, sx;
    int tiles_x, tiles_y, ntiles;
    int nchunk1, nchunk2;
    int ktot, cop;     // wgrad workspace extents
};

template <int KSY_, int KSX_, int XEXTRA_, int LSTEP_, int CK_, int CST_, int STRP_, int NT_, int PX_, int MINW_>
struct FwdCfg {
    static constexpr int KSY = KSY_, KSX = KSX_, T = KSY_ * KSX_;
    static constexpr int LSTEP = LSTEP_;   // LDS pixels between neighbouring output pixels
    static constexpr int CK = CK_;         // reduction run per tap (floats)
    static constexpr int CST = CST_;       // channels staged per halo pixel and chunk
    static constexpr int STRP = STRP_;     // LDS floats per halo pixel
    static constexpr int STRB = CK_ + 4;   // LDS floats per weight row
    static constexpr int NT = NT_, BN = 32 * NT_;
    static constexpr int PX = PX_, PY = 32 / PX_, MT = 2, TH = PY * MT * 4;
    static constexpr int HXP = (PX - 1) * LSTEP + KSX + XEXTRA_;
    static constexpr int HYP = (TH - 1) * LSTEP + KSY;
    static constexpr int A_FLOATS = ((HXP * HYP * STRP + 3) / 4) * 4;
    static constexpr int B_FLOATS = T * BN * STRB;
    static constexpr int LDS_BYTES = (A_FLOATS + B_FLOATS) * 4;
    static constexpr int MINW = MINW_;
};

// Stage one chunk of the input halo tile into LDS.  Halo pixel (hy,hx) <-> logical input pixel
// (iy0 + hy*gstep, ix0 + hx*gstep); out-of-image pixels and channels >= csrc are zero (the conv's zero padding).
template <int CST, int STRP, int HXP, int HYP>
__device__ __forceinline__ void stage_halo(float* __restrict__ As, const float* __restrict__ src, int csrc, int cb,
                                           int hs, int ws, int gmode, int img, int iy0, int ix0, int gstep,
                                           int h_in, int w_in, float sy, float sx, int tid) {
    constexpr int C4 = CST / 4;
    constexpr int NV = HXP * HYP * C4;
    const bool vec = (csrc & 3) == 0;
    for (int idx = tid; idx < NV; idx += 256) {
        const int p = idx / C4;
        const int c4 = idx - p * C4;
        const int hy = p / HXP;
        const int hx = p - hy * HXP;
        const int ly = iy0 + hy * gstep;
        const int lx = ix0 + hx * gstep;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int c = cb + c4 * 4;
        if (ly >= 0 && ly < h_in && lx >= 0 && lx < w_in && c < csrc) {
            int py = ly, px = lx;
            bool ok = true;
            if (gmode == RCF_GATHER_NEAREST) {
                // PyTorch nearest: src = min(floor(dst * (float)in/out), in-1)  (UpSampleKernel nearest_idx)
                py = min((int)floorf((float)ly * sy), hs - 1);
                px = min((int)floorf((float)lx * sx), ws - 1);
            } else if (gmode == RCF_GATHER_ZERO_INSERT) {
                ok = ((ly | lx) & 1) == 0;
                py = ly >> 1;
                px = lx >> 1;
                ok = ok && py < hs && px < ws;
            }
            if (ok) {
                const float* g = src + (((size_t)img * hs + py) * ws + px) * csrc + c;
                if (vec) {
                    v = *reinterpret_cast<const f32x4*>(g);
                } else {
                    v[0] = g[0];
                    if (c + 1 < csrc) v[1] = g[1];
                    if (c + 2 < csrc) v[2] = g[2];
                    if (c + 3 < csrc) v[3] = g[3];
                }
            }
        }
        *reinterpret_cast<f32x4*>(As + p * STRP + c4 * 4) = v;
    }
}

template <class C>
__global__ void __launch_bounds__(256, C::MINW) conv_fwd_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + C::A_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int lh = lane >> 5;

    int t = blockIdx.x;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int img = t / a.tiles_y;
    const int oy0 = ty * C::TH;
    const int ox0 = tx * C::PX;
    const int iy0 = oy0 * a.stride - a.pad;
    const int ix0 = ox0 * a.stride - a.pad;

    int abase[C::MT];
#pragma unroll
    for (int mi = 0; mi < C::MT; ++mi) {
        const int tr = (wave * C::MT + mi) * C::PY + li / C::PX;
        const int tc = li % C::PX;
        abase[mi] = (tr * C::LSTEP * C::HXP + tc * C::LSTEP) * C::STRP + 4 * lh;
    }
    const int bbase = li * C::STRB + 4 * lh;

    f32x16 acc[C::MT][C::NT];
#pragma unroll
    for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int nchunk = a.nchunk1 + a.nchunk2;
    constexpr int WCHUNK = C::T * C::BN * C::CK;
    const float* wp = a.wp + (size_t)blockIdx.y * nchunk * WCHUNK;

    for (int q = 0; q < nchunk; ++q) {
        const bool first = q < a.nchunk1;
        const float* src = first ? a.in1 : a.in2;
        const int csrc = first ? a.c1 : a.c2;
        const int cb = (first ? q : q - a.nchunk1) * C::CST;
        const int hs = first ? a.h1 : a.h_in;
        const int ws = first ? a.w1 : a.w_in;
        const int gmode = first ? a.gather1 : RCF_GATHER_DIRECT;

        __syncthreads();   // everyone is done reading the previous chunk
        stage_halo<C::CST, C::STRP, C::HXP, C::HYP>(As, src, csrc, cb, hs, ws, gmode, img, iy0, ix0, a.gstep,
                                                    a.h_in, a.w_in, a.sy, a.sx, tid);
        {
            constexpr int K4 = C::CK / 4;
            constexpr int NVB = C::T * C::BN * K4;
            const f32x4* wsrc = reinterpret_cast<const f32x4*>(wp + (size_t)q * WCHUNK);
            for (int idx = tid; idx < NVB; idx += 256) {
                const int row = idx / K4;
                const int k4 = idx - row * K4;
                *reinterpret_cast<f32x4*>(Bs + row * C::STRB + k4 * 4) = wsrc[idx];
            }
        }
        __syncthreads();

#pragma unroll
        for (int tap = 0; tap < C::T; ++tap) {
            const int toff = ((tap / C::KSX) * C::HXP + (tap % C::KSX)) * C::STRP;
#pragma unroll
            for (int s = 0; s < C::CK / 8; ++s) {
                f32x4 av[C::MT], bv[C::NT];
#pragma unroll
                for (int mi = 0; mi < C::MT; ++mi)
                    av[mi] = *reinterpret_cast<const f32x4*>(As + abase[mi] + toff + s * 8);
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni)
                    bv[ni] = *reinterpret_cast<const f32x4*>(Bs + bbase + (tap * C::BN + ni * 32) * C::STRB + s * 8);
#pragma unroll
                for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                    for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
                        for (int ni = 0; ni < C::NT; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][kq], bv[ni][kq], acc[mi][ni], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: store (+accumulate) and BN statistics ----
    const int n0 = blockIdx.y * C::BN;
    float s1[C::NT], s2[C::NT];
#pragma unroll
    for (int ni = 0; ni < C::NT; ++ni) { s1[ni] = 0.f; s2[ni] = 0.f; }
#pragma unroll
    for (int mi = 0; mi < C::MT; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rcf_mfma_row(r, lh);
            const int oy = oy0 + (wave * C::MT + mi) * C::PY + row / C::PX;
            const int ox = ox0 + row % C::PX;
            const bool pix_ok = oy < a.h_out && ox < a.w_out;
            const size_t pbase = (((size_t)img * a.h_out + oy) * a.w_out + ox) * a.c_out;
#pragma unroll
            for (int ni = 0; ni < C::NT; ++ni) {
                const int co = n0 + ni * 32 + li;
                if (pix_ok && co < a.c_out) {
                    float v = acc[mi][ni][r];
                    if (a.accumulate) v += a.out[pbase + co];
                    a.out[pbase + co] = v;
                    s1[ni] += v;
                    s2[ni] += v * v;
                }
            }
        }
    }
    if (a.stats != nullptr) {
        __syncthreads();   // LDS is free again
        // in-lane fp32 sums cover <= 32 values; everything across lanes / waves / workgroups is fp64
        // (PyTorch's CPU BatchNorm accumulates float tensors in double as well)
        double* red = reinterpret_cast<double*>(smem); // [4 waves][BN][2]
#pragma unroll
        for (int ni = 0; ni < C::NT; ++ni) {
            const double d1 = (double)s1[ni], d2 = (double)s2[ni];
            const double t1 = d1 + __shfl_xor(d1, 32);
            const double t2 = d2 + __shfl_xor(d2, 32);
            if (lh == 0) {
                red[(wave * C::BN + ni * 32 + li) * 2 + 0] = t1;
                red[(wave * C::BN + ni * 32 + li) * 2 + 1] = t2;
            }
        }
        __syncthreads();
        if (tid < C::BN) {
            const int co = n0 + tid;
            if (co < a.c_out) {
                double t1 = 0.0, t2 = 0.0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    t1 += red[(w * C::BN + tid) * 2 + 0];
                    t2 += red[(w * C::BN + tid) * 2 + 1];
                }
                a.stats[((size_t)blockIdx.x * 2 + 0) * a.c_out + co] = t1;
                a.stats[((size_t)blockIdx.x * 2 + 1) * a.c_out + co] = t2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient.  GEMM view: dW[k][co] = sum_pixels A[pixel][k] * dZ[pixel][co]; MFMA rows i = 32
// consecutive k of one tap (32 input channels of one halo pixel, contiguous in the LDS halo tile),
// columns j = 32 output channels, the MFMA's reduction index = pixel (lane half h takes pixel 2t+h).
// A workgroup walks many spatial tiles (persistent over gridDim.x splits) for one (k-chunk, 32-co group);
// wave w reduces the tile rows of slice w, the 4 slices are summed through LDS at the end, and one partial
// [T*32][32] per workgroup goes to the workspace (reduced deterministically by wgrad_reduce_kernel).
template <int KSY_, int KSX_, int XEXTRA_, int LSTEP_, int CST_, int STRP_, int PX_, int TH_, int MINW_>
struct WgCfg {
    static constexpr int KSY = KSY_, KSX = KSX_, T = KSY_ * KSX_;
    static constexpr int LSTEP = LSTEP_;
    static constexpr int CST = CST_, STRP = STRP_;
    static constexpr int PX = PX_, TH = TH_, TP = PX_ * TH_;
    static constexpr int RS = TH_ / 4;   // tile rows per wave
    static constexpr int HXP = (PX - 1) * LSTEP + KSX + XEXTRA_;
    static constexpr int HYP = (TH - 1) * LSTEP + KSY;
    static constexpr int A_FLOATS = ((HXP * HYP * STRP + 3) / 4) * 4;
    static constexpr int D_FLOATS = TP * 32;
    static constexpr int RED_FLOATS = T * 16 * 64;   // one wave's accumulators
    static constexpr int LDS_FLOATS = (A_FLOATS + D_FLOATS) > RED_FLOATS ? (A_FLOATS + D_FLOATS) : RED_FLOATS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int MINW = MINW_;
};

template <class C>
__global__ void __launch_bounds__(256, C::MINW) conv_wgrad_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Ds = smem + C::A_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int lh = lane >> 5;

    const int q = blockIdx.y;
    const int co0 = blockIdx.z * 32;
    const bool first = q < a.nchunk1;
    const float* src = first ? a.in1 : a.in2;
    const int csrc = first ? a.c1 : a.c2;
    const int cb = (first ? q : q - a.nchunk1) * C::CST;
    const int hs = first ? a.h1 : a.h_in;
    const int ws = first ? a.w1 : a.w_in;
    const int gmode = first ? a.gather1 : RCF_GATHER_DIRECT;

    f32x16 acc[C::T];
#pragma unroll
    for (int tap = 0; tap < C::T; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tap][r] = 0.f;

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % a.tiles_x;
        t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        const int img = t / a.tiles_y;
        const int oy0 = ty * C::TH;
        const int ox0 = tx * C::PX;

        __syncthreads();
        stage_halo<C::CST, C::STRP, C::HXP, C::HYP>(As, src, csrc, cb, hs, ws, gmode, img, oy0 * a.stride - a.pad,
                                                    ox0 * a.stride - a.pad, a.gstep, a.h_in, a.w_in, a.sy, a.sx, tid);
        for (int idx = tid; idx < C::TP * 8; idx += 256) {
            const int p = idx >> 3;
            const int c4 = idx & 7;
            const int oy = oy0 + p / C::PX;
            const int ox = ox0 + p % C::PX;
            const int co = co0 + c4 * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (oy < a.h_out && ox < a.w_out && co < a.c_out)
                v = *reinterpret_cast<const f32x4*>(a.dz + (((size_t)img * a.h_out + oy) * a.w_out + ox) * a.c_out + co);
            *reinterpret_cast<f32x4*>(Ds + p * 32 + c4 * 4) = v;
        }
        __syncthreads();

#pragma unroll
        for (int r = 0; r < C::RS; ++r) {
            const int trow = wave * C::RS + r;
#pragma unroll 2
            for (int tc = 0; tc < C::PX; tc += 2) {
                const int pc = tc + lh;
                const float b = Ds[(trow * C::PX + pc) * 32 + li];
                const int hb = (trow * C::LSTEP * C::HXP + pc * C::LSTEP) * C::STRP + li;
#pragma unroll
                for (int tap = 0; tap < C::T; ++tap) {
                    const float av = As[hb + ((tap / C::KSX) * C::HXP + (tap % C::KSX)) * C::STRP];
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[tap], 0, 0, 0);
                }
            }
        }
    }

    // sum the 4 waves' accumulators through LDS (wave 3 -> 2 -> 1 -> 0 chain keeps it deterministic)
    float* red = smem;
    for (int s = 3; s >= 1; --s) {
        __syncthreads();
        if (wave == s) {
#pragma unroll
            for (int tap = 0; tap < C::T; ++tap)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(tap * 16 + r) * 64 + lane] = acc[tap][r];
        }
        __syncthreads();
        if (wave == s - 1) {
#pragma unroll
            for (int tap = 0; tap < C::T; ++tap)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tap][r] += red[(tap * 16 + r) * 64 + lane];
        }
    }
    if (wave == 0) {
        float* wsp = a.ws + (size_t)blockIdx.x * a.ktot * a.cop;
#pragma unroll
        for (int tap = 0; tap < C::T; ++tap)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = (q * C::T + tap) * 32 + rcf_mfma_row(r, lh);
                wsp[(size_t)k * a.cop + co0 + li] = acc[tap][r];
            }
    }
}

// workspace [nslot][ktot][cop] -> dW in OIHW.  One thread per (k, co), co fastest (coalesced reads).
// kind: 0 generic (k = (q*T+tap)*32 + channel-in-chunk), 1 stem (k = tap(ky)*32 + kx*4 + c).
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nslot, int ktot, int cop,
                                    int c_out, int c1, int c2, int nchunk1, int T, int ksx, int kind) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ktot * cop) return;
    const int co = idx % cop;
    const int k = idx / cop;
    if (co >= c_out) return;
    const int row = k & 31;
    const int qt = k >> 5;
    const int tap = qt % T;
    const int q = qt / T;
    int ci, ky, kx;
    const int ksy_total = (kind == 1) ? T : T / ksx;
    if (kind == 1) {
        kx = row >> 2;
        ci = row & 3;
        ky = tap;
        if (kx >= 7 || ci >= c1) return;
    } else {
        ky = tap / ksx;
        kx = tap % ksx;
        if (q < nchunk1) {
            ci = q * 32 + row;
            if (ci >= c1) return;
        } else {
            ci = (q - nchunk1) * 32 + row;
            if (ci >= c2) return;
            ci += c1;
        }
    }
    double s = 0.0;   // the per-workgroup partials cancel heavily for BN-followed convs: sum them in fp64
    const size_t stride = (size_t)ktot * cop;
    for (int sl = 0; sl < nslot; ++sl) s += (double)ws[sl * stride + idx];
    const int kw = (kind == 1) ? 7 : ksx;
    dw[(((size_t)co * (c1 + c2) + ci) * ksy_total + ky) * kw + kx] = (float)s;
}

// OIHW -> [n-tile][chunk][tap][BN][CK].  kind 0 generic, 1 stem (k = kx*4 + c, tap = ky).
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ dst, size_t total, int w_o, int w_i,
                                    int ks, int mode, int i_off, int c_out, int c1, int c2, int nchunk1, int nchunk,
                                    int T, int ksx, int BN, int CK, int kind) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    size_t t = idx;
    const int k = t % CK;
    t /= CK;
    const int j = t % BN;
    t /= BN;
    const int tap = t % T;
    t /= T;
    const int q = t % nchunk;
    const int nt = t / nchunk;
    const int co = nt * BN + j;
    float v = 0.f;
    int cin = -1, ky, kx;
    if (kind == 1) {
        kx = k >> 2;
        const int c = k & 3;
        ky = tap;
        if (kx < 7 && c < c1) cin = c;
    } else {
        ky = tap / ksx;
        kx = tap % ksx;
        if (q < nchunk1) {
            const int c = q * CK + k;
            if (c < c1) cin = c;
        } else {
            const int c = (q - nchunk1) * CK + k;
            if (c < c2) cin = c1 + c;
        }
    }
    if (cin >= 0 && co < c_out) {
        if (mode == RCF_W_FORWARD) {
            v = w[(((size_t)co * w_i + cin) * ks + ky) * ks + kx];
        } else {   // dgrad: this conv's input channel cin is the forward output channel; taps flipped
            v = w[(((size_t)cin * w_i + (i_off + co)) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];
        }
    }
    dst[idx] = v;
}

// ------------------------------------------------------------------------------------------------
// configuration tables
enum Kind { K3S1 = 0, K3S2 = 1, K1 = 2, K7S2 = 3 };

struct Sel {
    int kind, ck, nt, px;
    int th, bn, t, cst;
};

template <class C>
int launch_fwd(const ConvArgs& a, int ntile_n, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fwd_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS_BYTES);
        attr_done = true;
    }
    dim3 grid(a.ntiles, ntile_n, 1);
    hipLaunchKernelGGL((conv_fwd_kernel<C>), grid, dim3(256), C::LDS_BYTES, st, a);
    return rcf_launch_status();
}

template <class C>
int launch_wgrad(const ConvArgs& a, int nsplit, int nchunk, int ncog, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS_BYTES);
        attr_done = true;
    }
    dim3 grid(nsplit, nchunk, ncog);
    hipLaunchKernelGGL((conv_wgrad_kernel<C>), grid, dim3(256), C::LDS_BYTES, st, a);
    return rcf_launch_status();
}

//                 KSY KSX XE LS  CK CST STRP NT PX MINW
using F3S1_16_1_32 = FwdCfg<3, 3, 0, 1, 16, 16, 20, 1, 32, 2>;
using F3S1_16_2_32 = FwdCfg<3, 3, 0, 1, 16, 16, 20, 2, 32, 2>;
using F3S1_16_1_16 = FwdCfg<3, 3, 0, 1, 16, 16, 20, 1, 16, 2>;
using F3S1_16_2_16 = FwdCfg<3, 3, 0, 1, 16, 16, 20, 2, 16, 2>;
using F3S1_8_1_32 = FwdCfg<3, 3, 0, 1, 8, 8, 12, 1, 32, 2>;
using F3S1_8_2_32 = FwdCfg<3, 3, 0, 1, 8, 8, 12, 2, 32, 2>;
using F3S1_8_1_16 = FwdCfg<3, 3, 0, 1, 8, 8, 12, 1, 16, 2>;
using F3S1_8_2_16 = FwdCfg<3, 3, 0, 1, 8, 8, 12, 2, 16, 2>;
using F3S2_8_1_32 = FwdCfg<3, 3, 0, 2, 8, 8, 12, 1, 32, 1>;
using F3S2_8_2_32 = FwdCfg<3, 3, 0, 2, 8, 8, 12, 2, 32, 1>;
using F3S2_8_1_16 = FwdCfg<3, 3, 0, 2, 8, 8, 12, 1, 16, 1>;
using F3S2_8_2_16 = FwdCfg<3, 3, 0, 2, 8, 8, 12, 2, 16, 1>;
using F1_32_1_32 = FwdCfg<1, 1, 0, 1, 32, 32, 36, 1, 32, 2>;
using F1_32_2_32 = FwdCfg<1, 1, 0, 1, 32, 32, 36, 2, 32, 2>;
using F1_32_1_16 = FwdCfg<1, 1, 0, 1, 32, 32, 36, 1, 16, 2>;
using F1_32_2_16 = FwdCfg<1, 1, 0, 1, 32, 32, 36, 2, 16, 2>;
using F1_16_1_32 = FwdCfg<1, 1, 0, 1, 16, 16, 20, 1, 32, 2>;
using F1_16_2_32 = FwdCfg<1, 1, 0, 1, 16, 16, 20, 2, 32, 2>;
using F1_16_1_16 = FwdCfg<1, 1, 0, 1, 16, 16, 20, 1, 16, 2>;
using F1_16_2_16 = FwdCfg<1, 1, 0, 1, 16, 16, 20, 2, 16, 2>;
using F7_32 = FwdCfg<7, 1, 7, 2, 32, 4, 4, 1, 32, 2>;
using F7_16 = FwdCfg<7, 1, 7, 2, 32, 4, 4, 1, 16, 2>;

//               KSY KSX XE LS CST STRP PX TH MINW
using W3S1_32 = WgCfg<3, 3, 0, 1, 32, 32, 32, 8, 2>;
using W3S1_16 = WgCfg<3, 3, 0, 1, 32, 32, 16, 16, 2>;
using W3S2_32 = WgCfg<3, 3, 0, 2, 32, 32, 32, 4, 1>;
using W3S2_16 = WgCfg<3, 3, 0, 2, 32, 32, 16, 8, 1>;
using W1_32 = WgCfg<1, 1, 0, 1, 32, 32, 32, 8, 2>;
using W1_16 = WgCfg<1, 1, 0, 1, 32, 32, 16, 16, 2>;
using W7_32 = WgCfg<7, 1, 7, 2, 4, 4, 32, 8, 2>;
using W7_16 = WgCfg<7, 1, 7, 2, 4, 4, 16, 16, 2>;

int ceil_div(int a, int b) { return (a + b - 1) / b; }

// tile utilisation of a PX x TH tiling of a w x h image
double tile_eff(int w, int h, int px, int th) {
    return ((double)w / (ceil_div(w, px) * px)) * ((double)h / (ceil_div(h, th) * th));
}

bool valid_desc(const rcf_conv_desc* d) {
    if (!d) return false;
    if (d->n <= 0 || d->h_in <= 0 || d->w_in <= 0 || d->c1 <= 0 || d->c2 < 0 || d->h_out <= 0 || d->w_out <= 0 ||
        d->c_out <= 0)
        return false;
    if (d->ksize != 1 && d->ksize != 3 && d->ksize != 7) return false;
    if (d->stride != 1 && d->stride != 2) return false;
    if (d->gather1 < 0 || d->gather1 > 2) return false;
    if (d->gather1 == RCF_GATHER_DIRECT && (d->h_src1 != d->h_in || d->w_src1 != d->w_in)) return false;
    if (d->h_src1 <= 0 || d->w_src1 <= 0) return false;
    if ((d->h_in + 2 * d->pad - d->ksize) / d->stride + 1 != d->h_out) return false;
    if ((d->w_in + 2 * d->pad - d->ksize) / d->stride + 1 != d->w_out) return false;
    if (d->w_mode == RCF_W_FORWARD) {
        if (d->w_o != d->c_out || d->w_i != d->c1 + d->c2) return false;
    } else if (d->w_mode == RCF_W_DGRAD) {
        if (d->w_o != d->c1 || d->c2 != 0 || d->w_i_off < 0 || d->w_i_off + d->c_out > d->w_i) return false;
    } else {
        return false;
    }
    return true;
}

int select_cfg(const rcf_conv_desc* d, Sel* s) {
    if (!valid_desc(d)) return RCF_EINVAL;
    if (d->c_out % 4 != 0) return RCF_EUNSUPPORTED;   // c_out == 1 is the head kernel's job
    const int cmax = d->c1 > d->c2 ? d->c1 : d->c2;
    s->nt = d->c_out > 32 ? 2 : 1;
    if (d->ksize == 7) {
        if (d->stride != 2 || d->c2 != 0 || d->c1 > 4 || d->gather1 != RCF_GATHER_DIRECT || d->w_mode != RCF_W_FORWARD)
            return RCF_EUNSUPPORTED;
        s->kind = K7S2; s->ck = 32; s->cst = 4; s->nt = 1; s->t = 7;
    } else if (d->ksize == 3) {
        s->t = 9;
        if (d->stride == 2) { s->kind = K3S2; s->ck = 8; }
        else { s->kind = K3S1; s->ck = cmax <= 8 ? 8 : 16; }
        s->cst = s->ck;
    } else {
        s->kind = K1; s->t = 1;
        s->ck = cmax <= 16 ? 16 : 32;
        s->cst = s->ck;
    }
    if ((d->c1 % 4 != 0 || (d->c2 % 4 != 0)) && s->kind != K7S2) {
        // scalar staging path handles it, but concat boundaries must stay 4-aligned
        if (d->c2 != 0) return RCF_EUNSUPPORTED;
    }
    // tile shape: 32x8 or 16x16 output pixels, whichever wastes less at the image edges
    const double e32 = tile_eff(d->w_out, d->h_out, 32, 8);
    const double e16 = tile_eff(d->w_out, d->h_out, 16, 16);
    s->px = e16 > e32 + 1e-9 ? 16 : 32;
    s->th = 256 / s->px;
    s->bn = 32 * s->nt;
    return RCF_OK;
}

void fill_args(const rcf_conv_desc* d, const Sel& s, ConvArgs* a) {
    a->n = d->n; a->h_in = d->h_in; a->w_in = d->w_in; a->c1 = d->c1; a->c2 = d->c2;
    a->h1 = d->h_src1; a->w1 = d->w_src1; a->gather1 = d->gather1;
    a->h_out = d->h_out; a->w_out = d->w_out; a->c_out = d->c_out; a->pad = d->pad; a->stride = d->stride;
    a->gstep = d->ksize == 1 ? d->stride : 1;
    a->accumulate = d->accumulate;
    a->sy = (float)d->h_src1 / (float)d->h_in;
    a->sx = (float)d->w_src1 / (float)d->w_in;
    a->tiles_x = ceil_div(d->w_out, s.px);
    a->tiles_y = ceil_div(d->h_out, s.th);
    a->ntiles = d->n * a->tiles_x * a->tiles_y;
    a->nchunk1 = ceil_div(d->c1, s.cst);
    a->nchunk2 = d->c2 > 0 ? ceil_div(d->c2, s.cst) : 0;
}

// wgrad tiling for the forward descriptor
struct WSel { int kind, px, th, t, cst, nchunk1, nchunk2, ncog, nsplit, ktot, cop, tiles_x, tiles_y, ntiles; };

int select_wgrad(const rcf_conv_desc* d, WSel* w) {
    if (!valid_desc(d) || d->w_mode != RCF_W_FORWARD) return RCF_EINVAL;
    if (d->c_out % 4 != 0) return RCF_EUNSUPPORTED;
    if (d->ksize == 7) {
        if (d->stride != 2 || d->c2 != 0 || d->c1 > 4 || d->gather1 != RCF_GATHER_DIRECT) return RCF_EUNSUPPORTED;
        w->kind = K7S2; w->t = 7; w->cst = 4;
    } else if (d->ksize == 3) {
        w->kind = d->stride == 2 ? K3S2 : K3S1; w->t = 9; w->cst = 32;
    } else {
        w->kind = K1; w->t = 1; w->cst = 32;
    }
    if ((d->c1 % 4 != 0 || d->c2 % 4 != 0) && w->kind != K7S2 && d->c2 != 0) return RCF_EUNSUPPORTED;
    const int th32 = (w->kind == K3S2) ? 4 : 8;
    const int th16 = (w->kind == K3S2) ? 8 : 16;
    const double e32 = tile_eff(d->w_out, d->h_out, 32, th32);
    const double e16 = tile_eff(d->w_out, d->h_out, 16, th16);
    w->px = e16 > e32 + 1e-9 ? 16 : 32;
    w->th = w->px == 32 ? th32 : th16;
    w->tiles_x = ceil_div(d->w_out, w->px);
    w->tiles_y = ceil_div(d->h_out, w->th);
    w->ntiles = d->n * w->tiles_x * w->tiles_y;
    w->nchunk1 = (w->kind == K7S2) ? 1 : ceil_div(d->c1, 32);
    w->nchunk2 = d->c2 > 0 ? ceil_div(d->c2, 32) : 0;
    w->ncog = ceil_div(d->c_out, 32);
    const int combos = (w->nchunk1 + w->nchunk2) * w->ncog;
    int ns = ceil_div(768, combos);
    if (ns > w->ntiles) ns = w->ntiles;
    if (ns < 1) ns = 1;
    w->nsplit = ns;
    w->ktot = (w->nchunk1 + w->nchunk2) * w->t * 32;
    w->cop = w->ncog * 32;
    return RCF_OK;
}

}   // namespace

extern "C" int rcf_conv2d_query(const rcf_conv_desc* d, rcf_conv_info* info) {
    if (!info) return RCF_EINVAL;
    Sel s;
    int rc = select_cfg(d, &s);
    if (rc != RCF_OK) return rc;
    ConvArgs a;
    fill_args(d, s, &a);
    const int ntile_n = ceil_div(d->c_out, s.bn);
    info->packed_weight_floats = (size_t)ntile_n * (a.nchunk1 + a.nchunk2) * s.t * s.bn * s.ck;
    info->n_partials = a.ntiles;
    info->kernel_id = s.kind * 1000 + s.ck * 10 + s.nt + (s.px == 16 ? 100 : 0);
    info->wgrad_workspace_floats = 0;
    info->wgrad_kernel_id = 0;
    if (d->w_mode == RCF_W_FORWARD) {
        WSel w;
        if (select_wgrad(d, &w) == RCF_OK) {
            info->wgrad_workspace_floats = (size_t)w.nsplit * w.ktot * w.cop;
            info->wgrad_kernel_id = 10000 + w.kind * 1000 + (w.px == 16 ? 100 : 0);
        }
    }
    return RCF_OK;
}

extern "C" int rcf_conv2d_pack_weights(const rcf_conv_desc* d, const float* w_oihw, float* packed, void* stream) {
    if (!w_oihw || !packed) return RCF_EINVAL;
    Sel s;
    int rc = select_cfg(d, &s);
    if (rc != RCF_OK) return rc;
    ConvArgs a;
    fill_args(d, s, &a);
    const int ntile_n = ceil_div(d->c_out, s.bn);
    const int nchunk = a.nchunk1 + a.nchunk2;
    const size_t total = (size_t)ntile_n * nchunk * s.t * s.bn * s.ck;
    const int ksx = s.kind == K7S2 ? 1 : d->ksize;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, packed, total, d->w_o,
                       d->w_i, d->ksize, d->w_mode, d->w_i_off, d->c_out, d->c1, d->c2, a.nchunk1, nchunk, s.t, ksx, s.bn,
                       s.ck, s.kind == K7S2 ? 1 : 0);
    return rcf_launch_status();
}

extern "C" int rcf_conv2d_fwd(const rcf_conv_desc* d, const float* in1, const float* in2, const float* packed, float* out,
                              double* stat_partials, void* stream) {
    if (!in1 || !packed || !out) return RCF_EINVAL;
    Sel s;
    int rc = select_cfg(d, &s);
    if (rc != RCF_OK) return rc;
    if (d->c2 > 0 && !in2) return RCF_EINVAL;
    ConvArgs a;
    fill_args(d, s, &a);
    a.in1 = in1; a.in2 = in2; a.wp = packed; a.out = out; a.stats = stat_partials; a.dz = nullptr; a.ws = nullptr;
    a.ktot = 0; a.cop = 0;
    const int nn = ceil_div(d->c_out, s.bn);
    hipStream_t st = (hipStream_t)stream;
    const int p16 = s.px == 16;
    switch (s.kind) {
        case K3S1:
            if (s.ck == 16) {
                if (s.nt == 1) return p16 ? launch_fwd<F3S1_16_1_16>(a, nn, st) : launch_fwd<F3S1_16_1_32>(a, nn, st);
                return p16 ? launch_fwd<F3S1_16_2_16>(a, nn, st) : launch_fwd<F3S1_16_2_32>(a, nn, st);
            }
            if (s.nt == 1) return p16 ? launch_fwd<F3S1_8_1_16>(a, nn, st) : launch_fwd<F3S1_8_1_32>(a, nn, st);
            return p16 ? launch_fwd<F3S1_8_2_16>(a, nn, st) : launch_fwd<F3S1_8_2_32>(a, nn, st);
        case K3S2:
            if (s.nt == 1) return p16 ? launch_fwd<F3S2_8_1_16>(a, nn, st) : launch_fwd<F3S2_8_1_32>(a, nn, st);
            return p16 ? launch_fwd<F3S2_8_2_16>(a, nn, st) : launch_fwd<F3S2_8_2_32>(a, nn, st);
        case K1:
            if (s.ck == 32) {
                if (s.nt == 1) return p16 ? launch_fwd<F1_32_1_16>(a, nn, st) : launch_fwd<F1_32_1_32>(a, nn, st);
                return p16 ? launch_fwd<F1_32_2_16>(a, nn, st) : launch_fwd<F1_32_2_32>(a, nn, st);
            }
            if (s.nt == 1) return p16 ? launch_fwd<F1_16_1_16>(a, nn, st) : launch_fwd<F1_16_1_32>(a, nn, st);
            return p16 ? launch_fwd<F1_16_2_16>(a, nn, st) : launch_fwd<F1_16_2_32>(a, nn, st);
        case K7S2:
            return p16 ? launch_fwd<F7_16>(a, nn, st) : launch_fwd<F7_32>(a, nn, st);
    }
    return RCF_EUNSUPPORTED;
}

extern "C" int rcf_conv2d_wgrad(const rcf_conv_desc* d, const float* in1, const float* in2, const float* dz,
                                float* dw_oihw, float* workspace, void* stream) {
    if (!in1 || !dz || !dw_oihw || !workspace) return RCF_EINVAL;
    WSel w;
    int rc = select_wgrad(d, &w);
    if (rc != RCF_OK) return rc;
    if (d->c2 > 0 && !in2) return RCF_EINVAL;
    ConvArgs a;
    a.n = d->n; a.h_in = d->h_in; a.w_in = d->w_in; a.c1 = d->c1; a.c2 = d->c2;
    a.h1 = d->h_src1; a.w1 = d->w_src1; a.gather1 = d->gather1;
    a.h_out = d->h_out; a.w_out = d->w_out; a.c_out = d->c_out; a.pad = d->pad; a.stride = d->stride;
    a.gstep = d->ksize == 1 ? d->stride : 1;
    a.accumulate = 0;
    a.sy = (float)d->h_src1 / (float)d->h_in;
    a.sx = (float)d->w_src1 / (float)d->w_in;
    a.tiles_x = w.tiles_x; a.tiles_y = w.tiles_y; a.ntiles = w.ntiles;
    a.nchunk1 = w.nchunk1; a.nchunk2 = w.nchunk2;
    a.in1 = in1; a.in2 = in2; a.wp = nullptr; a.out = nullptr; a.stats = nullptr; a.dz = dz; a.ws = workspace;
    a.ktot = w.ktot; a.cop = w.cop;
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = w.nchunk1 + w.nchunk2;
    const int p16 = w.px == 16;
    switch (w.kind) {
        case K3S1: rc = p16 ? launch_wgrad<W3S1_16>(a, w.nsplit, nchunk, w.ncog, st) : launch_wgrad<W3S1_32>(a, w.nsplit, nchunk, w.ncog, st); break;
        case K3S2: rc = p16 ? launch_wgrad<W3S2_16>(a, w.nsplit, nchunk, w.ncog, st) : launch_wgrad<W3S2_32>(a, w.nsplit, nchunk, w.ncog, st); break;
        case K1: rc = p16 ? launch_wgrad<W1_16>(a, w.nsplit, nchunk, w.ncog, st) : launch_wgrad<W1_32>(a, w.nsplit, nchunk, w.ncog, st); break;
        case K7S2: rc = p16 ? launch_wgrad<W7_16>(a, w.nsplit, nchunk, w.ncog, st) : launch_wgrad<W7_32>(a, w.nsplit, nchunk, w.ncog, st); break;
        default: return RCF_EUNSUPPORTED;
    }
    if (rc != RCF_OK) return rc;
    const int total = w.ktot * w.cop;
    const int ksx = w.kind == K7S2 ? 1 : d->ksize;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, workspace, dw_oihw, w.nsplit, w.ktot,
                       w.cop, d->c_out, d->c1, d->c2, w.nchunk1, w.t, ksx, w.kind == K7S2 ? 1 : 0);
    return rcf_launch_status();
}
