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
//
// This file is compiled twice (one translation unit each, so hipcc runs them in parallel): rcf_conv.hip with fp32 NHWC tensors
// (RCF_CONV_B16 = 0, the public extern "C" entry points) and rcf_conv_b16.hip with bf16 NHWC tensors (RCF_CONV_B16 = 1:
// rcf_conv_desc.storage == RCF_STORE_BF16, reached through the same entry points).
#include "rcf_common.h"
#include <type_traits>
#include <stdlib.h>

#ifndef RCF_CONV_B16
#define RCF_CONV_B16 0
#endif

namespace {

#if RCF_CONV_B16
using SAct = StB16;   // storage of activations and their gradients in this translation unit
#else
using SAct = StF32;
#endif
// the two stem convolutions read the fp32 network input whatever the activation storage is (3 / 2 channels staged 4 wide)
template <class C>
using SInOf = std::conditional_t<(C::CST == 4), StF32, SAct>;

// padding / out-of-image / out-of-range-channel lanes of the DMA and split weight-gradient kernels load from here instead of
// branching: a zero-initialised device global (one copy per device, nothing allocated at run time).  The kernels take its address
// from ConvArgs::zero (host: zero_page_ptr()), never from the symbol: named in device code, a `const` array sits in the constant
// address space, the select between it and a global pointer is a generic pointer and every staging load of
// conv_wgrad_split_kernel was a flat_load (which counts in lgkmcnt too: each wait for an LDS read also waited for the next tile's
// global loads); a non-const one is reached through the GOT, an s_load + s_waitcnt lgkmcnt(0) in front of every load.
__device__ __attribute__((aligned(256))) float rcf_zero_page[64] = {};

struct ConvArgs {
    const float* bias;   // fused inference epilogue (conv_split_kernel<C, true>): out = lrelu(acc + bias[co]), then lrelu(. + res) if res
    const float* res;
    const float* in1;
    const float* in2;
    const float* coef1;   // split kernels: in1 / in2 are RAW conv outputs of a BatchNorm + LeakyReLU block whose activation was
    const float* coef2;   // never written; y = lrelu(z * coef[0][c] + coef[1][c]) is applied as the operand is staged (else null)
    const float* wp;
    float* out;
    double* stats;
    const float* dz;   // wgrad only
    const float* zero; // rcf_zero_page (a kernel argument: see there)
    float* ws;         // wgrad only
    // RCF_PREC_F16X2 (two fp16 operand planes): device pointers to max|x| of the A operand's sources (forward / input gradient: in1,
    // in2; weight gradient: in1, in2) and of the B operand (the weights the packed buffer was built from; weight gradient: dz).
    // Null = unscaled (scale 1: data inside fp16's range).  The two sources of a concat share the scale of the larger maximum.
    const float* amax_a1;
    const float* amax_a2;
    const float* amax_b;
    // conv_split_kernel<C, false, SI, SO, true> (an input-gradient launch that is the only writer of dY of a BatchNorm + LeakyReLU
    // block): the block's raw conv output z (same geometry as `out`) and its coefficients [4][c_out] (scale, shift, mean, invstd);
    // `stats` then receives sum g and sum g * xhat with g = dY * lrelu'(z * scale + shift), xhat = (z - mean) * invstd
    const float* bz;
    const float* bk;
    int n, h_in, w_in, c1, c2, h1, w1, gather1;
    int h_out, w_out, c_out, pad, pad_x, stride, gstep, accumulate;
    int os, ooy, oox, ohp, owp;   // output (and wgrad dZ) phase addressing
    int ioy, iox;                 // RCF_GATHER_STRIDED2 input offsets
    int vt, hp, nimg;             // virtual tall image: rows = n*(h+1) with a zero separator row after each image
    float inv_hp;
    int phase_sum;                // sum the four input phases (up-2x dgrad) inside one launch
    int wp_phase_stride;          // floats between the phases' packed weights
    float sy, sx;
    int tiles_x, tiles_y, ntiles;
    int nchunk1, nchunk2;
    int ktot, cop;     // wgrad workspace extents
    int xcd_band;      // deal the tiles to the workgroups in eight contiguous bands, one per XCD (TileWalk)
};

// The order in which a persistent workgroup walks the tiles.  Workgroup b runs on XCD b % 8 (observed; MI355X_MICROARCH.md "Workgroup
// dispatch"), and every XCD has its own L2: dealt round-robin, the two workgroups that share a tile border (3x3: a quarter of a tile's
// input is halo) sit on different XCDs and both fetch it from HBM / the Infinity Cache.  With xcd_band the tiles are cut into eight
// contiguous bands -- whole horizontal strips of the batch -- and XCD k walks band k, so neighbours in a strip are in flight on the SAME
// L2 at about the same time.  Which workgroup computes a tile changes, the tile's result does not.
struct TileWalk {
    int step, end;
    __device__ __forceinline__ int first(const ConvArgs& a) {
        if (a.xcd_band && (gridDim.x & 7) == 0) {
            const int band = (a.ntiles + 7) >> 3, x = blockIdx.x & 7;
            step = gridDim.x >> 3;
            end = (x + 1) * band < a.ntiles ? (x + 1) * band : a.ntiles;
            const int t = x * band + (int)(blockIdx.x >> 3);
            return t < end ? t : a.ntiles;
        }
        step = gridDim.x;
        end = a.ntiles;
        return blockIdx.x;
    }
    __device__ __forceinline__ int next(int tile, const ConvArgs& a) const {
        const int t = tile + step;
        return t < end ? t : a.ntiles;
    }
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

// Halo staging, split in three so the HBM/L2 latency hides under the MFMAs of the previous chunk:
//   halo_setup  once per (tile, source): each thread's NA halo elements -> global pixel index (or -1 for zero padding /
//               zero-insert holes / outside the tile); all the gather math (nearest upsample, zero dilation) lives here
//   halo_load   per chunk: NA independent 16-B loads into registers (no LDS, no waits)
//   halo_store  per chunk: registers -> LDS
// Halo pixel (hy,hx) <-> logical input pixel (iy0 + hy*gstep, ix0 + hx*gstep).  Thread t owns channel group t % C4 of
// pixels t / C4 + i * (256 / C4).
template <int CST, int STRP, int HXP, int HYP>
struct Halo {
    static constexpr int C4 = CST / 4;
    static constexpr int PPI = 256 / C4;                       // halo pixels covered per iteration
    static constexpr int NA = (HXP * HYP + PPI - 1) / PPI;
    int pix[NA];                                               // element index of the pixel in the source, / csrc; -1: zero
    unsigned okm;                                              // bit i: the load of pix[i] is real data (NA <= 32)

    // vt != 0: rows are VIRTUAL rows of the batch stacked vertically with one zero row after each image (hp = h + 1);
    // that separator is exactly the zero padding of a stride-1 3x3 conv, so tiles may straddle images.
    __device__ __forceinline__ void setup(int hs, int ws, int gmode, int img, int iy0, int ix0, int gstep, int h_in, int w_in,
                                          float sy, float sx, int tid, int ioy = 0, int iox = 0, int vt = 0, int hp = 1,
                                          float inv_hp = 1.f, int nimg = 1) {
        const int p0 = tid / C4;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int p = p0 + i * PPI;
            const int hy = p / HXP;
            const int hx = p - hy * HXP;
            const int ly = iy0 + hy * gstep;
            const int lx = ix0 + hx * gstep;
            int v = -1;
            if (vt) {
                if (p < HXP * HYP && ly >= 0 && lx >= 0 && lx < w_in) {
                    const int im = (int)(((float)ly + 0.5f) * inv_hp);   // exact: ly < 2^20, |frac - k| >= 0.5/hp
                    const int y = ly - im * hp;
                    if (im < nimg && y < h_in) v = (im * hs + y) * ws + lx;
                }
            } else if (p < HXP * HYP && ly >= 0 && ly < h_in && lx >= 0 && lx < w_in) {
                int py = ly, px = lx;
                bool ok = true;
                if (gmode == RCF_GATHER_NEAREST) {
                    // PyTorch nearest: src = min(floor(dst * (float)in/out), in-1)
                    py = min((int)floorf((float)ly * sy), hs - 1);
                    px = min((int)floorf((float)lx * sx), ws - 1);
                } else if (gmode == RCF_GATHER_ZERO_INSERT) {
                    ok = ((ly | lx) & 1) == 0;
                    py = ly >> 1;
                    px = lx >> 1;
                    ok = ok && py < hs && px < ws;
                } else if (gmode == RCF_GATHER_STRIDED2) {
                    py = 2 * ly + ioy;
                    px = 2 * lx + iox;
                    ok = py < hs && px < ws;
                }
                if (ok) v = (img * hs + py) * ws + px;
            }
            pix[i] = v;
        }
    }

    // Loads are branch-free: a clamped (always valid) address, and the value is replaced by zero only when it is written to
    // LDS -- a branch or a select right behind each load makes the compiler wait for it before issuing the next.
    template <class S>
    __device__ __forceinline__ void load(f32x4 (&r)[NA], const float* __restrict__ src, int csrc, int cb, int tid) {
        const int c = cb + (tid % C4) * 4;
        const bool cok = c < csrc;
        okm = 0xffffffffu;
        if ((csrc & 3) == 0) {
            const int cld = cok ? c : 0;
            okm = 0u;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (cok && pix[i] >= 0) okm |= 1u << i;
                r[i] = rcf_ld4<S>(src, (size_t)(pix[i] < 0 ? 0 : pix[i]) * csrc + cld);
            }
        } else {   // stems (3 / 2 input channels): scalar loads
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (cok && pix[i] >= 0) {
                    const size_t g = (size_t)pix[i] * csrc + c;
                    v[0] = rcf_ld1<S>(src, g);
                    if (c + 1 < csrc) v[1] = rcf_ld1<S>(src, g + 1);
                    if (c + 2 < csrc) v[2] = rcf_ld1<S>(src, g + 2);
                    if (c + 3 < csrc) v[3] = rcf_ld1<S>(src, g + 3);
                }
                r[i] = v;
            }
        }
    }

    __device__ __forceinline__ void store(const f32x4 (&r)[NA], float* __restrict__ As, int tid) const {
        const int p0 = tid / C4;
        const int c4 = tid % C4;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int p = p0 + i * PPI;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (p < HXP * HYP) *reinterpret_cast<f32x4*>(As + p * STRP + c4 * 4) = ((okm >> i) & 1u) ? r[i] : z;
        }
    }
};

template <class C, class SI = SInOf<C>, class SO = SAct>
__global__ void __launch_bounds__(256, C::MINW) conv_fwd_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + C::A_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int lh = lane >> 5;

    int abase[C::MT];
#pragma unroll
    for (int mi = 0; mi < C::MT; ++mi) {
        const int tr = (wave * C::MT + mi) * C::PY + li / C::PX;
        const int tc = li % C::PX;
        abase[mi] = (tr * C::LSTEP * C::HXP + tc * C::LSTEP) * C::STRP + 4 * lh;
    }
    const int bbase = li * C::STRB + 4 * lh;

    f32x16 acc[C::MT][C::NT];

    const int nchunk = a.nchunk1 + a.nchunk2;
    constexpr int WCHUNK = C::T * C::BN * C::CK;
    const float* wp = a.wp + (size_t)blockIdx.y * nchunk * WCHUNK;
    const int n0 = blockIdx.y * C::BN;

    using H = Halo<C::CST, C::STRP, C::HXP, C::HYP>;
    constexpr int K4 = C::CK / 4;
    constexpr int NVB = C::T * C::BN * K4;
    constexpr int NB = (NVB + 255) / 256;
    H halo;
    f32x4 ra[H::NA];
    f32x4 rb[NB];

    // Work items are (tile, chunk) pairs; the workgroup is persistent over tiles (grid-stride) and the loads of item i+1
    // are issued before the MFMAs of item i, also across a tile boundary.
    const int nitem = a.phase_sum ? 4 * nchunk : nchunk;   // work items per tile
    auto load_item = [&](int tile, int item) {
        const int ph = a.phase_sum ? item / nchunk : 0;
        const int q = a.phase_sum ? item - ph * nchunk : item;
        const bool first = q < a.nchunk1;
        if (q == 0 || q == a.nchunk1) {
            int t = tile;
            const int tx = t % a.tiles_x;
            t /= a.tiles_x;
            const int ty = t % a.tiles_y;
            const int img = t / a.tiles_y;   // 0 in virtual-tall mode (tiles_y covers all images)
            const int pa = a.phase_sum ? (ph >> 1) : a.pad, pb = a.phase_sum ? (ph & 1) : a.pad_x;
            const int iy0 = ty * C::TH * a.stride - pa;
            const int ix0 = tx * C::PX * a.stride - pb;
            if (q == 0)
                halo.setup(a.h1, a.w1, a.gather1, img, iy0, ix0, a.gstep, a.h_in, a.w_in, a.sy, a.sx, tid,
                           a.phase_sum ? (ph >> 1) : a.ioy, a.phase_sum ? (ph & 1) : a.iox, a.vt, a.hp, a.inv_hp, a.nimg);
            else
                halo.setup(a.h_in, a.w_in, RCF_GATHER_DIRECT, img, iy0, ix0, a.gstep, a.h_in, a.w_in, 1.f, 1.f, tid, 0, 0, a.vt, a.hp,
                           a.inv_hp, a.nimg);
        }
        halo.template load<SI>(ra, first ? a.in1 : a.in2, first ? a.c1 : a.c2, (first ? q : q - a.nchunk1) * C::CST, tid);
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(wp + (size_t)ph * a.wp_phase_stride + (size_t)q * WCHUNK);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + i * 256;
            if (NVB % 256 == 0 || idx < NVB) rb[i] = wsrc[idx];
        }
    };

    // BatchNorm statistics of this workgroup's output channels, accumulated in fp64 over all its tiles (PyTorch's CPU
    // BatchNorm accumulates float tensors in double too) and written as ONE partial row per workgroup at the end.
    double st1[C::NT], st2[C::NT];
#pragma unroll
    for (int ni = 0; ni < C::NT; ++ni) { st1[ni] = 0.0; st2[ni] = 0.0; }

    int tile = blockIdx.x;
    int q = 0;
    if (tile < a.ntiles) load_item(tile, 0);
    while (tile < a.ntiles) {
        __syncthreads();   // everyone is done reading the previous item from LDS
        halo.store(ra, As, tid);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + i * 256;
            if (NVB % 256 == 0 || idx < NVB)
                *reinterpret_cast<f32x4*>(Bs + (idx / K4) * C::STRB + (idx % K4) * 4) = rb[i];
        }
        __syncthreads();
        int ntile = tile, nq = q + 1;
        if (nq == nitem) { nq = 0; ntile = tile + gridDim.x; }
        if (ntile < a.ntiles) load_item(ntile, nq);   // in flight while the MFMAs below run

        if (q == 0) {
#pragma unroll
            for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        }
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

        if (q == nitem - 1) {
            // ---- epilogue of this tile: store (+accumulate) and BN statistics ----
            int t = tile;
            const int tx = t % a.tiles_x;
            t /= a.tiles_x;
            const int ty = t % a.tiles_y;
            const int img = t / a.tiles_y;
            const int oy0 = ty * C::TH;
            const int ox0 = tx * C::PX;
            const bool want_stats = a.stats != nullptr;
#pragma unroll
            for (int mi = 0; mi < C::MT; ++mi) {
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 4) {
                    size_t pbase[4];
                    bool pok[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = rcf_mfma_row(r0 + j, lh);
                        int oy = oy0 + (wave * C::MT + mi) * C::PY + row / C::PX;
                        const int ox = ox0 + row % C::PX;
                        int im = img;
                        if (a.vt) {   // virtual row -> (image, row); separator rows produce no output
                            im = (int)(((float)oy + 0.5f) * a.inv_hp);
                            oy -= im * a.hp;
                            if (im >= a.nimg) oy = a.h_out;
                        }
                        const int py = oy * a.os + a.ooy, px = ox * a.os + a.oox;
                        pok[j] = oy < a.h_out && ox < a.w_out && py < a.ohp && px < a.owp;
                        pbase[j] = (((size_t)im * a.ohp + py) * a.owp + px) * a.c_out;
                    }
                    float old[4][C::NT];
                    if (a.accumulate) {   // all old values of the group in flight together (clamped address, used only where valid)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int ni = 0; ni < C::NT; ++ni) {
                                const int co = n0 + ni * 32 + li;
                                old[j][ni] = rcf_ld1<SO>(a.out, (pok[j] && co < a.c_out) ? pbase[j] + co : 0);
                            }
                        // land the loads inside this branch, or every store below waits on vmcnt(0) (see conv_split_kernel's epilogue)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int ni = 0; ni < C::NT; ++ni) asm volatile("" : "+v"(old[j][ni]));
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int ni = 0; ni < C::NT; ++ni) old[j][ni] = 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int ni = 0; ni < C::NT; ++ni) {
                            const int co = n0 + ni * 32 + li;
                            if (pok[j] && co < a.c_out) {
                                const float v = rcf_round_st<SO>(acc[mi][ni][r0 + j] + old[j][ni]);   // statistics of what is stored
                                rcf_st1<SO>(a.out, pbase[j] + co, v);
                                if (want_stats) {   // fp64 per value: E[x^2]-mean^2 must not depend on how tiles group the sum
                                    const double dv = (double)v;
                                    st1[ni] += dv;
                                    st2[ni] += dv * dv;
                                }
                            }
                        }
                }
            }
        }
        tile = ntile;
        q = nq;
    }

    if (a.stats != nullptr) {
        __syncthreads();   // LDS is free
        double* red = reinterpret_cast<double*>(smem);   // [4 waves][BN][2]
#pragma unroll
        for (int ni = 0; ni < C::NT; ++ni) {
            const double t1 = st1[ni] + __shfl_xor(st1[ni], 32);
            const double t2 = st2[ni] + __shfl_xor(st2[ni], 32);
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
// fp32 convolution on the bf16 matrix pipe ("split" kernel, 3x3 stride 1, forward and input gradient).
// Every fp32 operand x is split EXACTLY into three bf16 planes by truncation (x = p0 + p1 + p2: 3 x 8 significant bits =
// fp32's 24), and a*b is accumulated as the six partial products with i + j <= 2 on v_mfma_f32_32x32x16_bf16 (fp32
// accumulate).  The dropped terms are <= 2^-24 |ab|, below the accumulator's own rounding: measured error equals the exact
// f32 MFMA's (tools/probe/split_bf16_probe.hip: 1.0e-7 * sum|ab| at K = 2304 for both, 9 products == 6 products bitwise).
// 6 MFMAs x 32 cycles per 16-deep step against 8 x 64 cycles on the f32 MFMA: 2.7x the matrix rate.
//  * 8 waves (2 per SIMD), tile PX x TH (32x16 or 16x32) x BN; LDS holds the three planes of the halo tile
//    ([plane][pixel][16 bf16 + 8 pad]: conflict-free 16-B reads) and of the weight panel ([plane][tap][BN][16 bf16],
//    halves XOR-swizzled by (co >> 3) & 1), 143 KB for BN = 64.
//  * activations are split while staged (and/sub, ~5.5 VALU per element); weights are pre-split by the pack kernel.
//  * persistent + register-prefetched like conv_fwd_kernel; same epilogue (store, +=, fp64 BN statistics).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// The matrix instruction of the split kernels.  NPL = 3 / 1: bf16 planes.  NPL = 2 (RCF_PREC_F16X2): two fp16 planes of the operand
// scaled by its tensor's power-of-two scale (rcf_common.h: rcf_scale_of_amax) -- 22-23 significant bits in two planes where bf16 needs
// three, so three products a1*b0 + a0*b1 + a0*b0 instead of six.
typedef _Float16 rcf_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 rcf_f16x2 __attribute__((ext_vector_type(2)));
struct rcf_f16_pair { unsigned p0, p1; };
// two (already scaled) fp32 values -> (fp16 plane 0, fp16 plane 1) dwords, value `a` in the low half; both planes round to nearest
// even (v_cvt_pk_f16_f32), the residual a - p0 is exact in fp32
__device__ __forceinline__ rcf_f16_pair rcf_f16_planes(float a, float b) {
#ifdef RCF_DIAG_NO_CONVERT
    // diagnostics build (tools/probe): the split replaced by two byte permutes -- WRONG values, the same data movement: what the
    // kernels would cost if their operands arrived as planes (the conversion's share of their energy / time)
    rcf_f16_pair d;
    d.p0 = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
    d.p1 = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x05040100u);
    return d;
#endif
    const rcf_f32x2 v = {a, b};
    const rcf_f16x2 h = __builtin_convertvector(v, rcf_f16x2);
    const rcf_f32x2 r = {a - (float)h[0], b - (float)h[1]};
    rcf_f16_pair q;
    q.p0 = __builtin_bit_cast(unsigned, h);
    q.p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, rcf_f16x2));
    return q;
}
// scale and 1/scale of the A operand (common to both sources) and of the B operand, from the kernel's amax pointers (wave-uniform)
struct SplitScales { float sa, ia, sb, ib; };
__device__ __forceinline__ SplitScales rcf_split_scales(const float* amax_a1, const float* amax_a2, const float* amax_b) {
    float ma = amax_a1 ? *amax_a1 : 0.f;
    if (amax_a2) ma = fmaxf(ma, *amax_a2);
    const RcfScale a = (amax_a1 || amax_a2) ? rcf_scale_of_amax(ma) : RcfScale{1.f, 1.f};
    const RcfScale b = amax_b ? rcf_scale_of_amax(*amax_b) : RcfScale{1.f, 1.f};
    return SplitScales{a.s, a.inv, b.s, b.inv};
}
template <int NPL>
__device__ __forceinline__ f32x16 rcf_mfma_split(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (NPL == 2)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rcf_f16x8, a), __builtin_bit_cast(rcf_f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Work decomposition: a 256-thread workgroup owns a 32x8 / 16x16 output tile x BN output channels; two workgroups are resident per
// CU (<= 80 KB LDS, <= 256 VGPRs each), so one converts/stages while the other feeds the matrix pipe.  Per 16-channel chunk the
// fp32 halo tile is loaded once (registers), split, and written to LDS as three bf16 planes [pixel][16 ch] whose 16-B halves are
// XOR-swizzled by bit 3 of the pixel index (conflict-free ds_read_b128 without padding).  The pre-split packed weights arrive one
// KERNEL ROW (KSX taps) at a time into a double-buffered LDS piece, so only the A tile needs the two-barrier hand-over.
// NPL_ = 3: fp32 results (exact split, six partial products).  NPL_ = 1: rcf_conv_desc.precision == RCF_PREC_BF16 -- operands
// rounded to bf16 (nearest even), ONE product, fp32 accumulate: the "bf16" configurations of BASELINE.json.
// NPL_ = 2: RCF_PREC_F16X2 -- two fp16 planes of the operand scaled by its tensor's power-of-two scale (rcf_common.h: 22-23
// significant bits per operand, error <= 2^-23 |x| + 2^-39 amax), the THREE products a1b0 + a0b1 + a0b0, fp32 accumulate: measured
// against fp64 at or below the exact f32-MFMA kernel's error (tests/test_hip_f16x2.py), half the matrix work of the three-plane split.
// LSTEP_ = 2: stride-2 convolution -- the halo tile holds EVERY input pixel under the tile (a 3-tap kernel at stride 2 touches them
// all) and neighbouring output pixels read halo pixels two apart; everything else (weights, epilogue) is the stride-1 kernel.
// P4_ (KS_ = 2, two planes): the four OUTPUT phases of an up-2x forward (rcf_conv_desc.phase_sum == 2) from ONE staged tile.  Phase
// (a, b) is a 2x2 convolution of x with pad (1 - a, 1 - b): the four phases of an output tile read the same 3x3-halo tile of x, so it
// is loaded, split and written to LDS ONCE per channel chunk (halo geometry GK = 3) and the chunk's eight (phase, kernel row) weight
// pieces -- the per-phase packed layout as it is -- run over it into four accumulator sets, in the per-phase order (chunk, row, tap):
// bitwise the four per-phase launches, a quarter of their loads and conversions.
// PM_ = 2: the four OUTPUT phases of a 3x3 stride-2 convolution's input gradient (phase_sum == 3).  Phase (a, b) is a 2x2 convolution
// of dz with pad 0 whose taps (ty, tx) exist only for ty <= a, tx <= b (9 of 16; the per-phase launches multiply the other seven by
// zero weights): the dz tile is staged once per chunk and only the six real (phase, kernel row) pieces run, one or two taps each.
template <int KS_, int NT_, int PX_, int MT_ = 0, int NPL_ = 3, int LSTEP_ = 1, int PM_ = 0>
struct SplitCfg {
    static constexpr int PM = PM_;           // 0: one convolution; 1: up-2x forward phases; 2: stride-2 input-gradient phases
    static constexpr bool P4 = PM_ != 0;
    static_assert(!P4 || (KS_ == 2 && LSTEP_ == 1 && NPL_ == 2), "phase merging runs on two planes: 2x2 taps, stride 1");
    static constexpr int NPL = NPL_, NP = NPL_ == 3 ? 6 : (NPL_ == 2 ? 3 : 1);   // operand planes, partial products per MAC
    static constexpr int KSY = KS_, KSX = KS_, T = KS_ * KS_, LSTEP = LSTEP_, CK = 16, CST = 16;
    static constexpr int GK = PM_ == 1 ? 3 : KS_;      // halo geometry
    static constexpr int NPH = P4 ? 4 : 1;             // accumulator sets
    static constexpr int NROW = P4 ? 8 : KS_;          // (phase, kernel row) weight pieces per chunk (PM_ = 2 runs six of them)
    static constexpr int NT = NT_, BN = 32 * NT_;
    static constexpr int PX = PX_, PY = 32 / PX_, MT = MT_ ? MT_ : ((NT_ == 1 && KS_ == 3) ? 4 : 2), NW = 4, TH = PY * MT * NW;   // 3x3 32-co layers: 512-pixel tiles (2x2 phases: measured 10 % slower with them)
    static constexpr int HXP = (PX - 1) * LSTEP + GK, HYP = (TH - 1) * LSTEP + GK, NPIX = HXP * HYP;
    // A-tile layout in LDS (one plane).  HP (stride 1; three planes: 32-pixel rows only, the padded 16-pixel form does not fit the LDS):
    // [halo row][channel half][halo x][16 B].  The 32 lanes of an MFMA row block read consecutive pixels of one half of one or two
    // halo rows, 16 B apart -- conflict-free (for 16-pixel rows the row pitch is padded to a multiple of 16 slots so that the second
    // row's lanes fall on the banks the first row's leave free) -- and the address of tap (ky, kx) is the lane's base + a COMPILE-TIME
    // immediate: one address register per MFMA row block instead of ~5 VALU instructions per ds_read_b128 (the pixel-major layout
    // XOR-swizzles the halves by bit 3 of the pixel index, which the tap offset carries into).  These kernels run at the board's power
    // limit: instructions not issued are energy not spent.  Otherwise (stride 2): [pixel][2 x 16 B], halves swizzled.
    static constexpr bool HP = (LSTEP_ == 1) && (PX_ == 32 || NPL_ <= 2);
    static constexpr int HXH = (PX_ == 32) ? HXP : ((HXP + 7) / 8) * 8;   // 16-B slots per half row (16-pixel rows: row pitch 2 * HXH = 0 mod 16)
    static constexpr int ROWS16 = 2 * HXH;                                // 16-B slots per halo row
    static constexpr int A_PLANE_BYTES = HP ? HYP * ROWS16 * 16 : NPIX * 32;        // 16 bf16 per halo pixel
    static constexpr int A_BYTES = NPL * A_PLANE_BYTES;
    static constexpr int B_PLANE_BYTES = KSX * BN * 32;    // one kernel row: 16 bf16 per (tap, co)
    static constexpr int B_PIECE_BYTES = NPL * B_PLANE_BYTES;
    static constexpr int COEF_MAX_C = 512;                 // input channels (both sources) of the BN-on-load table
    static constexpr int COEF_BYTES = 2 * COEF_MAX_C * 4;
    static constexpr int LDS_BYTES = A_BYTES + 2 * B_PIECE_BYTES + COEF_BYTES;
    static constexpr int WCHUNK_BYTES = KSY * B_PIECE_BYTES;   // one chunk of pre-split packed weights
    // (three resident workgroups per CU fit the LDS of the two-plane 256-pixel tiles, but not the registers: compiled for 168 VGPRs the
    // 64-co kernels spill 54-136 of them -- measured in round 3, not shipped)
    static_assert(2 * LDS_BYTES <= 160 * 1024 || (LSTEP == 2 && LDS_BYTES <= 160 * 1024), "two workgroups per CU (stride 2, three planes: one)");
};

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
// fp32 -> bf16 (round to nearest even), result in the HIGH 16 bits (low bits garbage)
__device__ __forceinline__ unsigned rcf_bf16_rne(float x) {
    const unsigned u = __float_as_uint(x);
    return u + 0x7fffu + ((u >> 16) & 1u);
}
// s_waitcnt vmcnt(0) (expcnt/lgkmcnt untouched): LDS-DMA completion is tracked by vmcnt only, and the compiler does not know
// that a later ds_read depends on it -- the wait before the publishing barrier has to be explicit.
__device__ __forceinline__ void rcf_wait_dma() { __builtin_amdgcn_s_waitcnt(0x0F70); }

#ifdef RCF_PHASE_TIMING   // diagnostics build only (tools/phase_timing.py): where a wave of conv_split_kernel spends its cycles
__device__ unsigned long long rcf_phase_cycles[8];
#define RCF_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define RCF_TACC(slot, t1, t0) tacc[slot] += (t1) - (t0)
#else
#define RCF_T(var)
#define RCF_TACC(slot, t1, t0)
#endif

// EPI: inference epilogue -- BatchNorm folded into the weights (scale) and a per-channel bias, LeakyReLU, and the residual tail of
// ResNetBlock (lrelu(y + res)) applied to the accumulators before the only store; no statistics.
template <class C, bool EPI = false, class SI = SAct, class SO = SAct, bool BST = false>
__global__ void __launch_bounds__(256, 2) conv_split_kernel(ConvArgs a) {
    static_assert(!(EPI && BST), "the inference epilogue and the BatchNorm-backward sums exclude each other");
    static_assert(!SI::B16 || C::NPL == 1, "bf16 tensors go with bf16 operands");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* As = smem_b;
    unsigned char* Bs = smem_b + C::A_BYTES;
    float* coef_lds = reinterpret_cast<float*>(smem_b + C::A_BYTES + 2 * C::B_PIECE_BYTES);   // [2][c1 + c2]: scale, shift
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int lh = lane >> 5;
    if (a.coef1 != nullptr || a.coef2 != nullptr) {   // BN-on-load table: identity (1, 0) for a source without coefficients
        const int ctot = a.c1 + a.c2;
        for (int i = tid; i < ctot; i += 256) {
            const bool s1 = i < a.c1;
            const float* cf = s1 ? a.coef1 : a.coef2;
            const int cs = s1 ? a.c1 : a.c2, ci = s1 ? i : i - a.c1;
            coef_lds[i] = cf ? cf[ci] : 1.f;
            coef_lds[ctot + i] = cf ? cf[cs + ci] : 0.f;
        }
        // published by the first __syncthreads() of the kernel, which precedes every store_a that reads it... the prologue's
        // store_a comes first: make it explicit
        __syncthreads();
    }

    SplitScales sc = {1.f, 1.f, 1.f, 1.f};   // NPL == 2: operand scales (the packed weights already carry theirs)
    if constexpr (C::NPL == 2) sc = rcf_split_scales(a.amax_a1, a.amax_a2, a.amax_b);

    int apix[C::MT];   // halo pixel of this lane's output pixel at tap (0, 0)
#pragma unroll
    for (int mi = 0; mi < C::MT; ++mi) {
        const int tr = (wave * C::MT + mi) * C::PY + li / C::PX;
        const int tc = li % C::PX;
        apix[mi] = tr * C::LSTEP * C::HXP + tc * C::LSTEP;
        if (C::HP) apix[mi] = (tr * C::ROWS16 + lh * C::HXH + tc) * 16;   // HP layout: byte offset of tap (0, 0); taps add immediates
    }
    // byte offset inside an A plane of this lane's operand for MFMA row block mi at tap (ky, kx)
    auto a_off = [&](int mi, int ky, int kx) __attribute__((always_inline)) -> int {
        if constexpr (C::HP) return apix[mi] + (ky * C::ROWS16 + kx) * 16;
        else {
            int ap = apix[mi];
            if (C::MT > 2) asm volatile("" : "+v"(ap));   // recompute the swizzled address per tap: hoisted, the 2 x T addresses cost 18 VGPRs
            const int p = ap + ky * C::HXP + kx;
            return p * 32 + ((lh ^ ((p >> 3) & 1)) * 16);
        }
    };
    const int bbase = li * 32 + ((lh ^ ((li >> 3) & 1)) * 16);   // row co = ni*32 + li; (ni*32) keeps (co>>3)&1 == (li>>3)&1

    f32x16 acc[C::NPH][C::MT][C::NT];
    const int nchunk = a.nchunk1 + a.nchunk2;
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(a.wp) + (size_t)blockIdx.y * nchunk * C::WCHUNK_BYTES;
    const int n0 = blockIdx.y * C::BN;
    const int psum = C::P4 ? 0 : a.phase_sum;   // P4: one item per chunk, the phases live in the row loop

    // A staging: thread t owns channel quad t % 4 of halo pixels t / 4 + i * 64
    constexpr int NA = (C::NPIX + 63) / 64;
    // fp32 tensors: byte offset of the thread's 16 B (its channel quad of halo pixel i) at channel chunk 0, from the start of the tile's
    // first image in the source -- a buffer descriptor is built per tile and source, so 32 bits reach -- or 0xffffffff for padding /
    // outside pixels: the buffer unit answers those with zeros, so the plain path needs neither a clamped address nor a select, and
    // the per-chunk address of a load is this register + an SGPR.  bf16 tensors (the non-DMA fallback): pixel index, -1 = zero.
    int pix[NA];
    int fimg = 0;        // that first image (wave-uniform)
    f32x4 ra[NA];
    unsigned okm = 0u;   // bit i: ra[i] holds real data (BatchNorm-on-load / bf16 tensors: padding is re-zeroed at store time)
    int tch = 0;         // first channel (in c1 + c2 numbering) of the staged quad, and whether its source is BN-on-load
    bool ttf = false;

    auto setup = [&](int tile, bool first, int ph) {
        int t = tile;
        const int tx = t % a.tiles_x;
        t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        const int img = t / a.tiles_y;
        // phase_sum 1: the four INPUT phases of an up-2x input gradient summed; 2: the four OUTPUT phases of an up-2x forward, one after
        // the other on the same tile (pad 1 - a, 1 - b; the outputs go to (2y + a, 2x + b))
        const int pa = C::P4 ? (C::PM == 1 ? 1 : 0) : (psum == 2 ? 1 - (ph >> 1) : (psum ? (ph >> 1) : a.pad)),
                  pb = C::P4 ? (C::PM == 1 ? 1 : 0) : (psum == 2 ? 1 - (ph & 1) : (psum ? (ph & 1) : a.pad_x));
        const int ioy = psum ? (ph >> 1) : a.ioy, iox = psum ? (ph & 1) : a.iox;
        const int iy0 = ty * C::TH * C::LSTEP - pa;
        const int ix0 = tx * C::PX * C::LSTEP - pb;
        const int hs = first ? a.h1 : a.h_in, ws = first ? a.w1 : a.w_in;
        const int gmode = first ? a.gather1 : RCF_GATHER_DIRECT;
        const int pixb = (first ? a.c1 : a.c2) * 4;   // bytes per source pixel (fp32 tensors)
        fimg = img;
        if (a.vt) {   // the image of the tile's first real halo row
            fimg = (int)(((float)(iy0 < 0 ? 0 : iy0) + 0.5f) * a.inv_hp);
            fimg = fimg < a.nimg ? fimg : a.nimg - 1;
        }
        fimg = __builtin_amdgcn_readfirstlane(fimg);
        const int ibase = SI::B16 ? 0 : fimg;   // pixel indices count from this image
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int p = (tid >> 2) + i * 64;
            // 512-pixel tiles sit at the 256-VGPR limit: keep the compiler from hoisting the 2 x NA tile-invariant (hy, hx) out of the
            // tile loop (20 registers for a few integer instructions per tile)
            if (C::MT > 2) asm volatile("" : "+v"(p));
            const int hy = p / C::HXP;
            const int hx = p - hy * C::HXP;
            const int ly = iy0 + hy, lx = ix0 + hx;
            int v = -1;
            if (p < C::NPIX && ly >= 0 && lx >= 0 && lx < a.w_in) {
                if (a.vt) {
                    const int im = (int)(((float)ly + 0.5f) * a.inv_hp);
                    const int y = ly - im * a.hp;
                    if (im < a.nimg && y < a.h_in) v = ((im - ibase) * hs + y) * ws + lx;
                } else if (ly < a.h_in) {
                    int py = ly, px = lx;
                    bool ok = true;
                    if (gmode == RCF_GATHER_NEAREST) {
                        py = min((int)floorf((float)ly * a.sy), hs - 1);
                        px = min((int)floorf((float)lx * a.sx), ws - 1);
                    } else if (gmode == RCF_GATHER_STRIDED2) {
                        py = 2 * ly + ioy;
                        px = 2 * lx + iox;
                        ok = py < hs && px < ws;
                    } else if (gmode == RCF_GATHER_ZERO_INSERT) {
                        ok = ((ly | lx) & 1) == 0;
                        py = ly >> 1;
                        px = lx >> 1;
                        ok = ok && py < hs && px < ws;
                    }
                    if (ok) v = ((img - ibase) * hs + py) * ws + px;
                }
            }
            if constexpr (SI::B16) pix[i] = v;
            else pix[i] = v >= 0 ? v * pixb + (tid & 3) * 16 : -1;   // (-1 = 0xffffffff: out of the descriptor's range)
        }
    };
    const int nitem = psum ? 4 * nchunk : nchunk;   // A tiles per output tile
    // fp32 halo tile of one item -> registers.  Loads are unconditional from a clamped (always valid) address and zero-selected at
    // store time: a branch or a select behind each load makes the compiler wait for it before issuing the next.
    auto load_a = [&](int tile, int item) {
        const int ph = psum ? item / nchunk : 0;
        const int q = psum ? item - ph * nchunk : item;
        const bool first = q < a.nchunk1;
        if (q == 0 || q == a.nchunk1) setup(tile, first, ph);
        const float* src = first ? a.in1 : a.in2;
        const int csrc = first ? a.c1 : a.c2;
        const int cch = (first ? q : q - a.nchunk1) * 16 + (tid & 3) * 4;
        const bool cok = cch < csrc;
        const int cld = cok ? cch : 0;
        tch = (first ? 0 : a.c1) + cld;
        ttf = (first ? a.coef1 : a.coef2) != nullptr;
        okm = 0u;
        if constexpr (SI::B16) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (cok && pix[i] >= 0) okm |= 1u << i;
                ra[i] = rcf_ld4<SI>(src, (size_t)(pix[i] < 0 ? 0 : pix[i]) * csrc + cld);
            }
        } else {
            const int hs = first ? a.h1 : a.h_in, ws = first ? a.w1 : a.w_in;
            const __amdgpu_buffer_rsrc_t rsa = rcf_rsrc(src + (size_t)fimg * hs * ws * csrc);
            const unsigned cbb = (unsigned)((first ? q : q - a.nchunk1) * 64);   // this chunk's 16 channels: byte offset inside a pixel
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (cok && pix[i] != -1) okm |= 1u << i;
                // (a quad beyond the source's channels -- the last chunk of an 8- or 24-channel source -- is out of range like padding)
                ra[i] = rcf_buffer_load_f32x4(rsa, cok ? (unsigned)pix[i] : 0xffffffffu, cbb);
            }
        }
    };
    // BN selects the BatchNorm-on-load variant at COMPILE time: as one code path the compiler if-converts the (wave-uniform) test and
    // every plain conversion pays the multiply-add, compare and selects of the other variant (~100 of 330 VALU instructions per chunk;
    // a VALU wave-instruction costs about a fifth of an MFMA in energy, and these kernels are power-limited)
    auto store_a_impl = [&](auto bn_tag) __attribute__((always_inline)) {
        constexpr bool BN = decltype(bn_tag)::value;
        f32x4 tsc = {1.f, 1.f, 1.f, 1.f}, tsh = {0.f, 0.f, 0.f, 0.f};
        if (BN) {
            tsc = *reinterpret_cast<const f32x4*>(coef_lds + tch);
            tsh = *reinterpret_cast<const f32x4*>(coef_lds + a.c1 + a.c2 + tch);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int p = (tid >> 2) + i * 64;
            if (p < C::NPIX) {
                // exact 3-way truncation split: plane k keeps the next 8 significant bits
                unsigned x0[4], x1[4], x2[4];
                float xin[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float xv = ra[i][e];
                    if (BN) xv = rcf_lrelu(xv * tsc[e] + tsh[e]);   // the producer's BatchNorm + LeakyReLU, applied on load
                    // (fp32 tensors, plain path: out-of-range loads came back as zeros already)
                    const float x = (BN || SI::B16) ? (((okm >> i) & 1u) ? xv : 0.f) : xv;
                    xin[e] = x;
                    x0[e] = __float_as_uint(x) & 0xffff0000u;
                    const float r1 = x - __uint_as_float(x0[e]);
                    x1[e] = __float_as_uint(r1) & 0xffff0000u;
                    const float r2 = r1 - __uint_as_float(x1[e]);
                    x2[e] = __float_as_uint(r2);
                }
                const int cq = tid & 3;
                unsigned char* dst;
                if constexpr (C::HP) {
                    int po = p;
                    asm volatile("" : "+v"(po));   // recomputed per chunk: hoisted, the NA destinations stay live across the whole tile loop (spills)
                    const int hy = po / C::HXP, hx = po - hy * C::HXP;
                    dst = As + (hy * C::ROWS16 + (cq >> 1) * C::HXH + hx) * 16 + (cq & 1) * 8;
                } else {
                    dst = As + p * 32 + (((cq >> 1) ^ ((p >> 3) & 1)) * 16) + (cq & 1) * 8;
                }
                if (C::NPL == 1) {   // bf16 operands: round to nearest even instead of splitting
                    unsigned r[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        r[e] = rcf_bf16_rne(xin[e]);
                    }
                    u32x2 w0 = {(r[0] >> 16) | (r[1] & 0xffff0000u), (r[2] >> 16) | (r[3] & 0xffff0000u)};
                    *reinterpret_cast<u32x2*>(dst) = w0;
                } else {
                    u32x2 w0 = {(x0[0] >> 16) | x0[1], (x0[2] >> 16) | x0[3]};
                    u32x2 w1 = {(x1[0] >> 16) | x1[1], (x1[2] >> 16) | x1[3]};
                    if constexpr (C::NPL == 2) {   // two fp16 planes of the scaled value
                        const rcf_f16_pair q0 = rcf_f16_planes(xin[0] * sc.sa, xin[1] * sc.sa), q1 = rcf_f16_planes(xin[2] * sc.sa, xin[3] * sc.sa);
                        w0[0] = q0.p0; w1[0] = q0.p1;
                        w0[1] = q1.p0; w1[1] = q1.p1;
                    }
                    u32x2 w2 = {(x2[0] >> 16) | (x2[1] & 0xffff0000u), (x2[2] >> 16) | (x2[3] & 0xffff0000u)};
                    *reinterpret_cast<u32x2*>(dst) = w0;
                    *reinterpret_cast<u32x2*>(dst + C::A_PLANE_BYTES) = w1;
                    if (C::NPL == 3) *reinterpret_cast<u32x2*>(dst + 2 * C::A_PLANE_BYTES) = w2;
                }
            }
        }
    };
    auto store_a = [&]() __attribute__((always_inline)) {
        if (ttf) store_a_impl(std::true_type{});
        else store_a_impl(std::false_type{});
    };
    // one kernel row of pre-split weights: straight copy global -> LDS piece `buf` by LDS-DMA (each wave instruction moves 1 KiB to a
    // wave-uniform LDS base + 16 B x lane; no staging registers, no ds_write).  hipcc drains vmcnt before the next barrier.
    auto chunk_base = [&](int item) -> const unsigned char* {
        const int ph = psum ? item / nchunk : 0;
        const int q = psum ? item - ph * nchunk : item;
        return wp + (size_t)ph * a.wp_phase_stride * 4 + (size_t)q * C::WCHUNK_BYTES;
    };
    // (P4: row R of a chunk is kernel row R & 1 of phase R >> 1, each phase's packed weights where the per-phase launches read them)
    auto copy_b = [&](const unsigned char* cbase, int ky, int buf) {
        constexpr int NKB = C::B_PIECE_BYTES / 1024;
        static_assert(C::B_PIECE_BYTES % 1024 == 0, "weight piece must be whole KiB");
        const int w = __builtin_amdgcn_readfirstlane(wave);
        const unsigned char* wsrc = (C::P4 ? cbase + (size_t)(ky >> 1) * a.wp_phase_stride * 4 + (size_t)(ky & 1) * C::B_PIECE_BYTES
                                           : cbase + (size_t)ky * C::B_PIECE_BYTES) + lane * 16;
#pragma unroll
        for (int i = 0; i < (NKB + 3) / 4; ++i) {
            int kb = i * 4 + w;
            if ((i + 1) * 4 > NKB) kb = kb < NKB ? kb : NKB - 1;   // ragged tail: a duplicate copy of the last KiB is harmless
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + kb * 1024),
                                             (__attribute__((address_space(3))) void*)(Bs + buf * C::B_PIECE_BYTES + kb * 1024), 16, 0, 0);
        }
    };

    double st1[C::NT], st2[C::NT];
#pragma unroll
    for (int ni = 0; ni < C::NT; ++ni) { st1[ni] = 0.0; st2[ni] = 0.0; }

    // Phases: one kernel row (KSX taps) per barrier interval.  During a row's MFMAs the next row's weight piece arrives by DMA in
    // the other LDS slot and (one row before the chunk ends) the next A tile is loaded into registers; the A tile itself is handed
    // over between two barriers at the end of the chunk.  The second resident workgroup of the CU fills the matrix pipe meanwhile.
    TileWalk walk;
    int tile = walk.first(a);
    int q = 0;
    int pb = 0;   // LDS slot of the weight piece the current kernel row reads
    bf16x8 av[2][C::NPL][C::MT], bv[2][C::NPL][C::NT];
    auto fetch_a = [&](int ky, int kx, int slot) {
#pragma unroll
        for (int mi = 0; mi < C::MT; ++mi) {
            const int ao = a_off(mi, ky, kx);
#pragma unroll
            for (int pl = 0; pl < C::NPL; ++pl) av[slot][pl][mi] = as_bf16x8(*reinterpret_cast<const u32x4*>(As + pl * C::A_PLANE_BYTES + ao));
        }
    };
    auto fetch_b = [&](int kx, int slot, int bslot) {
        const unsigned char* Bp = Bs + bslot * C::B_PIECE_BYTES;
#pragma unroll
        for (int pl = 0; pl < C::NPL; ++pl)
#pragma unroll
            for (int ni = 0; ni < C::NT; ++ni)
                bv[slot][pl][ni] = as_bf16x8(*reinterpret_cast<const u32x4*>(Bp + pl * C::B_PLANE_BYTES + (kx * C::BN + ni * 32) * 32 + bbase));
    };
    const unsigned char* cb_cur = wp;   // packed weights of the current chunk
#ifdef RCF_PHASE_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    if (tile < a.ntiles) {
        load_a(tile, 0);
        copy_b(cb_cur, 0, 0);
        store_a();
    }
    rcf_wait_dma();
    __syncthreads();
    if (tile < a.ntiles) fetch_a(0, 0, 0);
    while (tile < a.ntiles) {
        int ntile = tile, nq = q + 1;
        if (nq == nitem) { nq = 0; ntile = walk.next(tile, a); }
        const bool more = ntile < a.ntiles;
        const unsigned char* cb_next = more ? chunk_base(nq) : wp;
        // phase_sum == 2 (the four output phases of an up-2x forward in one launch): every phase is a convolution of its own --
        // accumulators start at its first chunk, the epilogue runs at its last one and writes output pixels (2y + a, 2x + b)
        const bool phase_out = psum == 2;
        const int oph = phase_out ? q / nchunk : 0;
        if (phase_out ? (q - oph * nchunk == 0) : (q == 0)) {
#pragma unroll
            for (int pi = 0; pi < C::NPH; ++pi)
#pragma unroll
                for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[pi][mi][ni][r] = 0.f;
        }
        // phase merging: row ky = (phase ky >> 1, kernel row ky & 1) -- the halo row and first halo column its taps read, and how
        // many taps it has (stride-2 input gradient: phase (a, b) has the taps ty <= a, tx <= b only)
        auto halo_row = [](int ky) { return C::PM == 1 ? (ky >> 2) + (ky & 1) : (C::PM == 2 ? (ky & 1) : ky); };
        auto halo_col0 = [](int ky) { return C::PM == 1 ? ((ky >> 1) & 1) : 0; };
        // ky: this row; kyn: the row after it (-1: none, this is the chunk's last); load_row: the next item's A tile is loaded here
        auto row_body = [&](const int ky, const int kyn, const bool load_row) __attribute__((always_inline)) {
            const bool last_row = kyn < 0;
            const int pi = C::P4 ? (ky >> 1) : 0;
            const int hky = halo_row(ky), hkx0 = halo_col0(ky);
            const int ntap = (C::PM == 2 && ((ky >> 1) & 1) == 0) ? 1 : C::KSX;
            // every row starts in register set 0: its A operands were fetched before the barrier that published its weights
            RCF_T(t_row0);
            fetch_b(0, 0, pb);
            if (!last_row) copy_b(cb_cur, kyn, pb ^ 1);   // nobody reads that slot during this row
            else if (more) copy_b(cb_next, 0, pb ^ 1);
            if (load_row && more) load_a(ntile, nq);
            __builtin_amdgcn_sched_barrier(0);
            RCF_T(t_row1);
            RCF_TACC(0, t_row1, t_row0);   // 0: row prologue (first B reads, DMA / global-load issue)
#pragma unroll
            for (int kx = 0; kx < C::KSX; ++kx) {
                if (kx >= ntap) break;
                const int cur = kx & 1;
                // The 6 x MT x NT MFMAs of this tap in product-major, accumulator-round-robin order (dependent MFMAs stay MT x NT
                // apart), with the next tap's LDS reads issued ONE AT A TIME between them: issued as a block, the reads stall the
                // wave's MFMA issue for as long as the LDS queue takes them, and the partner wave on the SIMD tends to be doing
                // the same.  sched_barrier pins the hand-written order.
                constexpr int MN = C::MT * C::NT, NMF = C::NP * MN, NRD = C::NPL * (C::MT + C::NT);
                const bool has_next = kx + 1 < ntap;
                int nr = 0;
                int ao_next[C::MT];
#pragma unroll
                for (int j = 0; j < NMF; ++j) {
                    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // smallest partial products first
                    constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};                    // two planes: a1b0, a0b1, a0b0
                    const int pj = j / MN, mi = (j % MN) / C::NT, ni = j % C::NT;
                    const int pa = C::NPL == 3 ? PA[pj] : (C::NPL == 2 ? PA2[pj % 3] : 0), pbl = C::NPL == 3 ? PB[pj] : (C::NPL == 2 ? PB2[pj % 3] : 0);
                    acc[pi][mi][ni] = rcf_mfma_split<C::NPL>(av[cur][pa][mi], bv[cur][pbl][ni], acc[pi][mi][ni]);
                    if (has_next) {
#pragma unroll
                        for (int rep = 0; rep < 3; ++rep) {
                            if (nr < NRD && (nr + 1) * NMF <= (j + 1) * NRD) {
                                __builtin_amdgcn_sched_barrier(0);
                                if (nr < C::NPL * C::MT) {
                                    const int rmi = nr / C::NPL, pl = nr % C::NPL;
                                    if (pl == 0) ao_next[rmi] = a_off(rmi, hky, hkx0 + kx + 1);
                                    av[cur ^ 1][pl][rmi] = as_bf16x8(*reinterpret_cast<const u32x4*>(As + pl * C::A_PLANE_BYTES + ao_next[rmi]));
                                } else {
                                    const int rb = nr - C::NPL * C::MT, pl = rb / C::NT, rni = rb % C::NT;
                                    bv[cur ^ 1][pl][rni] = as_bf16x8(*reinterpret_cast<const u32x4*>(
                                        Bs + pb * C::B_PIECE_BYTES + pl * C::B_PLANE_BYTES + ((kx + 1) * C::BN + rni * 32) * 32 + bbase));
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                ++nr;
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            RCF_T(t_row2);
            RCF_TACC(1, t_row2, t_row1);   // 1: the row's MFMAs + interleaved LDS reads
            if (last_row) {
                if (phase_out ? (q - oph * nchunk == nchunk - 1) : (q == nitem - 1)) {
                  // one output tile of accumulator set PI at output offset (e_ooy, e_oox); P4 runs it once per phase (with 32 fp32
                  // output channels a pixel is one 128-byte line: the strided phase stores are whole lines already)
                  auto tile_epilogue = [&](auto ph_tag, const int e_ooy, const int e_oox) __attribute__((always_inline)) {
                    constexpr int PI = decltype(ph_tag)::value;
                    int t = tile;
                    const int tx = t % a.tiles_x;
                    t /= a.tiles_x;
                    const int ty = t % a.tiles_y;
                    const int img = t / a.tiles_y;
                    const int oy0 = ty * C::TH;
                    const int ox0 = tx * C::PX;
#ifdef RCF_DIAG_NO_STATS
                    const bool want_stats = false;
#else
                    const bool want_stats = !EPI && a.stats != nullptr;
#endif
                    float ebias[C::NT];
                    if (EPI) {
#pragma unroll
                        for (int ni = 0; ni < C::NT; ++ni) {
                            const int co = n0 + ni * 32 + li;
                            ebias[ni] = a.bias[co < a.c_out ? co : 0];
                        }
                        // land the bias loads here, not inside every masked store block below (see the note on `old`)
#pragma unroll
                        for (int ni = 0; ni < C::NT; ++ni) asm volatile("" : "+v"(ebias[ni]));
                    }
                    const bool add_old = EPI ? a.res != nullptr : a.accumulate != 0;
                    float bk[BST ? 2 : 1][C::NT];   // BST: scale, shift of this lane's channels (mean, invstd enter once, at the end)
                    if constexpr (BST) {
#pragma unroll
                        for (int ni = 0; ni < C::NT; ++ni) {
                            const int co = n0 + ni * 32 + li;
#pragma unroll
                            for (int e = 0; e < 2; ++e) bk[e][ni] = a.bk[e * a.c_out + (co < a.c_out ? co : 0)];
                        }
#pragma unroll
                        for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
                            for (int e = 0; e < 2; ++e) asm volatile("" : "+v"(bk[e][ni]));   // land them here (see `old`)
                    }
                    if constexpr (C::NPL == 2) {   // undo the operand scales: two exact multiplications by powers of two
#pragma unroll
                        for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
                            for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
                                for (int r = 0; r < 16; ++r) acc[PI][mi][ni][r] = acc[PI][mi][ni][r] * sc.ia * sc.ib;
                    }
                    if constexpr (SO::B16) {   // bf16 tensors that cannot take the DMA kernel (rcf_conv_b16_dma.h): 2-byte stores, bounds per store
                    // (BST: taking a group's sums one group later -- z loads in flight behind the next group's stores -- was built and
                    // measured: the extra live registers spill 11-31 VGPRs in the 3x3 configurations and the step is no faster)
#pragma unroll
                    for (int mi = 0; mi < C::MT; ++mi) {
#pragma unroll
                        for (int r0 = 0; r0 < 16; r0 += 4) {
                            // the four accumulator rows r0 .. r0 + 3 of a lane are four consecutive pixels of ONE tile row (MFMA row =
                            // j + 8 * (r0 / 4) + 4 * lh, PX a multiple of 4): one address computation per group, then + j pixels
                            size_t pbase[4];
                            bool pok[4];
                            if (C::MT != 4) {
                                const int row = rcf_mfma_row(r0, lh);
                                int oy = oy0 + (wave * C::MT + mi) * C::PY + row / C::PX;
                                const int ox = ox0 + row % C::PX;
                                int im = img;
                                if (a.vt) {   // virtual row -> (image, row); separator rows produce no output
                                    im = (int)(((float)oy + 0.5f) * a.inv_hp);
                                    oy -= im * a.hp;
                                    if (im >= a.nimg) oy = a.h_out;
                                }
                                const int py = oy * a.os + e_ooy, px = ox * a.os + e_oox;
                                const bool rowvalid = oy < a.h_out && py < a.ohp;
                                const size_t base0 = (((size_t)im * a.ohp + py) * a.owp + px) * a.c_out;
                                const int pstep = a.os * a.c_out;
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    pok[j] = rowvalid && ox + j < a.w_out && px + j * a.os < a.owp;
                                    pbase[j] = base0 + (size_t)(j * pstep);
                                }
                            }
                            if (C::MT == 4) {   // the 512-pixel configuration sits at 256 VGPRs: the per-row form allocates better there
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const int row = rcf_mfma_row(r0 + j, lh);
                                    int oy = oy0 + (wave * C::MT + mi) * C::PY + row / C::PX;
                                    const int ox = ox0 + row % C::PX;
                                    int im = img;
                                    if (a.vt) {
                                        im = (int)(((float)oy + 0.5f) * a.inv_hp);
                                        oy -= im * a.hp;
                                        if (im >= a.nimg) oy = a.h_out;
                                    }
                                    const int py = oy * a.os + e_ooy, px = ox * a.os + e_oox;
                                    pok[j] = oy < a.h_out && ox < a.w_out && py < a.ohp && px < a.owp;
                                    pbase[j] = (((size_t)im * a.ohp + py) * a.owp + px) * a.c_out;
                                }
                            }
                            float zv[BST ? 4 : 1][C::NT];
                            if constexpr (BST) {   // z of the BatchNorm block at the group's outputs, in flight together with `old`
#pragma unroll
                                for (int j = 0; j < 4; ++j)
#pragma unroll
                                    for (int ni = 0; ni < C::NT; ++ni) {
                                        const int co = n0 + ni * 32 + li;
                                        zv[j][ni] = rcf_ld1<SO>(a.bz, (pok[j] && co < a.c_out) ? pbase[j] + co : 0);
                                    }
                            }
                            float old[4][C::NT];
                            if (add_old) {   // all old values of the group in flight together (clamped address)
                                const auto* addsrc = EPI ? a.res : a.out;
#pragma unroll
                                for (int j = 0; j < 4; ++j)
#pragma unroll
                                    for (int ni = 0; ni < C::NT; ++ni) {
                                        const int co = n0 + ni * 32 + li;
                                        old[j][ni] = rcf_ld1<SO>(addsrc, (pok[j] && co < a.c_out) ? pbase[j] + co : 0);
                                    }
                                if constexpr (!EPI) {   // gradient accumulation: straight into the accumulators, inside this branch
#pragma unroll
                                    for (int j = 0; j < 4; ++j)
#pragma unroll
                                        for (int ni = 0; ni < C::NT; ++ni) acc[PI][mi][ni][r0 + j] += old[j][ni];
                                }
                                // Land the loads INSIDE this branch (training: by adding them into the accumulators here; inference: the
                                // empty asm below uses the registers -- either way the s_waitcnt goes here).
                                // On gfx9 stores count in vmcnt like loads: with the old values still pending at the join, hipcc put an
                                // s_waitcnt vmcnt(0) in front of EVERY store -- of the plain path too -- and the 64 stores of a tile
                                // ran one HBM round trip after the other (20-40 % of a wave's time).  (Wrapping the body in a lambda with
                                // a compile-time ADD instead costs the 512-pixel configuration ~60 spilled VGPRs: measured, slower.)
                                if constexpr (EPI) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j)
#pragma unroll
                                        for (int ni = 0; ni < C::NT; ++ni) asm volatile("" : "+v"(old[j][ni]));
                                }
                            } else if constexpr (EPI) {
#pragma unroll
                                for (int j = 0; j < 4; ++j)
#pragma unroll
                                    for (int ni = 0; ni < C::NT; ++ni) old[j][ni] = 0.f;
                            }
                            if constexpr (BST) {   // land the z loads in front of the stores (see the note on `old`)
#pragma unroll
                                for (int j = 0; j < 4; ++j)
#pragma unroll
                                    for (int ni = 0; ni < C::NT; ++ni) asm volatile("" : "+v"(zv[j][ni]));
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j)
#pragma unroll
                                for (int ni = 0; ni < C::NT; ++ni) {
                                    const int co = n0 + ni * 32 + li;
#ifdef RCF_DIAG_NO_STORE
                                    if (pok[j] && co < a.c_out && a.tiles_x < 0) {
#else
                                    if (pok[j] && co < a.c_out) {
#endif
                                        float v = acc[PI][mi][ni][r0 + j];
                                        if (EPI) {
                                            v = rcf_lrelu(v + ebias[ni]);
                                            if (a.res != nullptr) v = rcf_lrelu(v + old[j][ni]);
                                        }
                                        v = rcf_round_st<SO>(v);   // BatchNorm statistics of the values the tensor holds
                                        rcf_st1<SO>(a.out, pbase[j] + co, v);
                                        if constexpr (BST) {   // sum g and sum g * z of the gradient the tensor holds, in fp64
                                            const float zz = zv[j][ni];
                                            const float g = v * rcf_lrelu_grad(zz * bk[0][ni] + bk[1][ni]);
                                            st1[ni] += (double)g;
                                            st2[ni] += (double)g * (double)zz;
                                        } else
                                        if (want_stats) {   // fp64 per value: E[x^2]-mean^2 must not depend on how tiles group the sum
                                            const double dv = (double)v;
                                            st1[ni] += dv;
                                            st2[ni] += dv * dv;
                                        }
                                    }
                                }
                        }
                    }
                    } else {
                    // ---- fp32 tensors: ONE branch-free path for every tile (interior, image edge, virtual-tall separator rows, strided phase
                    // outputs, += / residual, the BatchNorm-backward sums).  Address = buffer descriptor of the tile's first image (SGPRs)
                    // + a wave-uniform row / pixel / n-tile offset (an SGPR: the instruction's soffset) + ONE per-lane register; a lane
                    // whose pixel or channel lies outside gets the offset 0xffffffff, which the buffer unit drops (stores) or answers with 0
                    // (loads).  Two VALU instructions per store.  (Round 4's path computed 64-bit addresses and four bounds per store behind
                    // an exec-mask branch each -- ~20 instructions per store, 28-31 % of a wave's time on the 64- and 32-channel layers,
                    // tools/phase_timing.py -- in kernels limited by board power: instructions not issued are energy not spent.)
                    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
                    int fim = img;
                    if (a.vt) {
                        fim = (int)(((float)oy0 + 0.5f) * a.inv_hp);
                        fim = fim < a.nimg ? fim : a.nimg - 1;
                    }
                    fim = __builtin_amdgcn_readfirstlane(fim);
                    const size_t img_el = (size_t)a.ohp * a.owp * a.c_out;
                    const unsigned rowb = (unsigned)a.owp * (unsigned)a.c_out * 4u;     // bytes per physical output row
                    const unsigned pstep = (unsigned)a.os * (unsigned)a.c_out * 4u;     // bytes between neighbouring output pixels
                    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)fim * img_el, 0, 0x7fffffff, 0x00020000);
                    const float* addsrc = EPI ? a.res : a.out;
                    const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(addsrc != nullptr ? addsrc : a.out) + (size_t)fim * img_el, 0, 0x7fffffff, 0x00020000);
                    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BST ? a.bz : a.out) + (size_t)fim * img_el, 0, 0x7fffffff, 0x00020000);
                    // columns of this tile that exist: logical (< w_out) and physical (px = (ox0 + c) * os + oox < owp)
                    int wlim = a.w_out - ox0;
                    {
                        const int wphys = (a.owp - e_oox - ox0 * a.os + a.os - 1) / a.os;
                        wlim = wlim < wphys ? wlim : wphys;
                    }
                    const unsigned xoff = (unsigned)(ox0 * a.os + e_oox) * (unsigned)a.c_out * 4u + (unsigned)n0 * 4u;
                    // per lane: its byte offset inside a pixel row block, and per n-tile the number of valid columns left of it (0 for a
                    // channel outside the tensor): accumulator row r of the lane is column (rr % PX) + 4 lh, valid iff rr % PX < cl[ni]
                    unsigned l0 = (unsigned)(4 * lh) * pstep + (unsigned)li * 4u;
                    int cl[C::NT];
#pragma unroll
                    for (int ni = 0; ni < C::NT; ++ni) cl[ni] = (n0 + ni * 32 + li < a.c_out) ? wlim - 4 * lh : 0;
                    // (opaque per tile: visible, hipcc hoists the tile-invariant pieces of all 16 x MT x NT offsets out of the tile loop and spills)
                    asm volatile("" : "+v"(l0));
                    auto epilogue = [&](auto add_tag, auto stats_tag) __attribute__((always_inline)) {
                        constexpr bool ADD = decltype(add_tag)::value, STATS = decltype(stats_tag)::value;
                        constexpr int RPB = 16 / C::PY;   // accumulator rows per tile row: 16 (32-pixel rows) or 8
                        // (the 64-co x 32-pixel-row BatchNorm-sums variants sit at the 256-register limit: four loads in flight there)
                        constexpr int EB = (BST && C::NT == 2 && C::PX == 32) ? 4 : 8;
#pragma unroll
                        for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
                            for (int rq = 0; rq < C::PY; ++rq) {
                                // the tile row of this group: wave-uniform image, row, validity and byte offset from the descriptor's base
                                int oy = oy0 + (wave_s * C::MT + mi) * C::PY + rq;
                                int im = img;
                                bool rok = true;
                                if (a.vt) {
                                    im = (int)(((float)oy + 0.5f) * a.inv_hp);
                                    oy -= im * a.hp;
                                    rok = im < a.nimg;
                                }
                                const int py = oy * a.os + e_ooy;
                                rok = rok && oy < a.h_out && py < a.ohp;
                                const unsigned rowoff = __builtin_amdgcn_readfirstlane((unsigned)((im - fim) * a.ohp + py) * rowb + xoff);
                                if (__builtin_amdgcn_readfirstlane((int)rok)) {
#pragma unroll
                                    for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
                                    for (int kb = 0; kb < RPB; kb += EB) {   // EB outputs at a time: their old values / z in flight together
                                        // (the masked offset is formed twice -- at the loads and at the store -- rather than kept: eight more live
                                        // registers put the 64-co x 32-pixel-row BatchNorm-sums variants over the 256 limit)
                                        float zv[BST ? EB : 1], old[ADD ? EB : 1];
                                        if constexpr (BST || ADD) {
#pragma unroll
                                            for (int k = 0; k < EB; ++k) {
                                                const int r = rq * RPB + kb + k;
                                                const int cc = ((r & 3) + 8 * (r >> 2)) % C::PX;   // rcf_mfma_row(r, 0) % PX: the lane part (4 lh) is in l0 / cl
                                                const unsigned vo = cc < cl[ni] ? l0 + (unsigned)(ni * 128) : 0xffffffffu;
                                                const unsigned so = rowoff + (unsigned)cc * pstep;
                                                if constexpr (BST) zv[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_z, vo, so, 0));
                                                if constexpr (ADD) old[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_add, vo, so, 0));
                                            }
                                        }
#pragma unroll
                                        for (int k = 0; k < EB; ++k) {
                                            const int r = rq * RPB + kb + k;
                                            const int cc = ((r & 3) + 8 * (r >> 2)) % C::PX;
                                            const unsigned so = rowoff + (unsigned)cc * pstep;
                                            unsigned lv = l0;
                                            if constexpr (BST || ADD) asm volatile("" : "+v"(lv));   // (or the compiler keeps the eight offsets of the loads)
                                            const unsigned vo = cc < cl[ni] ? lv + (unsigned)(ni * 128) : 0xffffffffu;
                                            float v = acc[PI][mi][ni][r];
                                            if constexpr (EPI) {
                                                v = rcf_lrelu(v + ebias[ni]);
                                                if constexpr (ADD) v = rcf_lrelu(v + old[k]);
                                            } else if constexpr (ADD) {
                                                v += old[k];
                                            }
                                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_out, vo, so, 0);
                                            if constexpr (BST) {   // sum g and sum g * z of the gradient the tensor holds, in fp64
                                                const float zz = zv[k];
                                                const float g = cc < cl[ni] ? v * rcf_lrelu_grad(zz * bk[0][ni] + bk[1][ni]) : 0.f;
                                                st1[ni] += (double)g;
                                                st2[ni] += (double)g * (double)zz;
                                            } else if constexpr (STATS) {   // fp64 per value: E[x^2] - mean^2 must not depend on how tiles group the sum
                                                const double dv = (double)(cc < cl[ni] ? v : 0.f);
                                                st1[ni] += dv;
                                                st2[ni] += dv * dv;
                                            }
                                        }
                                    }
                                }
                            }
                    };
                    if constexpr (BST) {   // the only writer of dY: nothing to add to, and the sums it takes are the block's, not its own
                        epilogue(std::false_type{}, std::false_type{});
                    } else if constexpr (EPI) {   // inference: no statistics
                        if (add_old) epilogue(std::true_type{}, std::false_type{});
                        else epilogue(std::false_type{}, std::false_type{});
                    } else if (add_old) {
                        if (want_stats) epilogue(std::true_type{}, std::true_type{});
                        else epilogue(std::true_type{}, std::false_type{});
                    } else {
                        if (want_stats) epilogue(std::false_type{}, std::true_type{});
                        else epilogue(std::false_type{}, std::false_type{});
                    }
                    }
                  };
                  if constexpr (C::P4) {   // (stride-2 input gradient: odd sizes give the phases different extents -- the bounds of each store handle it)
                      tile_epilogue(std::integral_constant<int, 0>{}, 0, 0);
                      tile_epilogue(std::integral_constant<int, 1>{}, 0, 1);
                      tile_epilogue(std::integral_constant<int, 2>{}, 1, 0);
                      tile_epilogue(std::integral_constant<int, 3>{}, 1, 1);
                  } else {
                      tile_epilogue(std::integral_constant<int, 0>{}, phase_out ? (oph >> 1) : a.ooy, phase_out ? (oph & 1) : a.oox);
                  }
                }
                RCF_T(t_e0);
                RCF_TACC(2, t_e0, t_row2);     // 2: output epilogue (last chunk of a tile only)
                __syncthreads();   // every wave is done reading the A tile
                RCF_T(t_e1);
                RCF_TACC(3, t_e1, t_e0);       // 3: barrier "A tile free"
                if (more) store_a();
                RCF_T(t_e2);
                RCF_TACC(4, t_e2, t_e1);       // 4: store_a (wait for the global loads, split, ds_write)
            } else {
                // next row's first A operands: the tile does not change inside a chunk
                fetch_a(halo_row(kyn), halo_col0(kyn), 0);
            }
            RCF_T(t_b0);
            rcf_wait_dma();   // the next weight piece has landed
            __syncthreads();
            RCF_T(t_b1);
            RCF_TACC(last_row ? 5 : 6, t_b1, t_b0);   // 5: DMA wait + publishing barrier after store_a; 6: the same between rows
            if (last_row && more) fetch_a(0, 0, 0);   // (P4: row 0 is phase (0, 0), kernel row 0: halo (0, 0) too)
            pb ^= 1;
        };
        if constexpr (C::PM == 1) {
            // eight rows x four epilogues exceed hipcc's full-unroll budget, and a rolled row loop would index the accumulator sets
            // dynamically (scratch): spelled out instead
            row_body(0, 1, false); row_body(1, 2, false); row_body(2, 3, false); row_body(3, 4, false);
            row_body(4, 5, false); row_body(5, 6, false); row_body(6, 7, true); row_body(7, -1, false);
        } else if constexpr (C::PM == 2) {   // the six real rows: phases (0,0) and (0,1) have kernel row 0 only
            row_body(0, 2, false); row_body(2, 4, false); row_body(4, 5, false); row_body(5, 6, false); row_body(6, 7, true); row_body(7, -1, false);
        } else {
#pragma unroll
            for (int ky = 0; ky < C::NROW; ++ky) row_body(ky, ky + 1 < C::NROW ? ky + 1 : -1, ky == (C::NROW >= 2 ? C::NROW - 2 : 0));
        }
        tile = ntile;
        q = nq;
        cb_cur = cb_next;
    }

#ifdef RCF_PHASE_TIMING
    tacc[7] = __builtin_amdgcn_s_memtime() - t_begin;   // 7: whole wave
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&rcf_phase_cycles[i], tacc[i]);
#endif
    if (a.stats != nullptr) {
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem_b);   // [4 waves][BN][2]
#pragma unroll
        for (int ni = 0; ni < C::NT; ++ni) {
            const double t1 = st1[ni] + __shfl_xor(st1[ni], 32);
            const double t2 = st2[ni] + __shfl_xor(st2[ni], 32);
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
                for (int w = 0; w < C::NW; ++w) {
                    t1 += red[(w * C::BN + tid) * 2 + 0];
                    t2 += red[(w * C::BN + tid) * 2 + 1];
                }
                // BST: sum g * xhat = invstd * (sum g * z - mean * sum g), formed in fp64 from this workgroup's fp64 sums (the
                // cancellation costs bits of the 53, not of a float)
                if constexpr (BST) t2 = (double)a.bk[3 * a.c_out + co] * (t2 - (double)a.bk[2 * a.c_out + co] * t1);
                a.stats[((size_t)blockIdx.x * 2 + 0) * a.c_out + co] = t1;
                a.stats[((size_t)blockIdx.x * 2 + 1) * a.c_out + co] = t2;
            }
        }
    }
}

// OIHW fp32 -> pre-split bf16 planes [n-tile][chunk][kernel row][plane][kx][BN][16], halves swapped when (co >> 3) & 1.
__device__ __forceinline__ void pack_weights_split_body(size_t idx, const float* __restrict__ w, unsigned short* __restrict__ dst,
                                                        size_t total_rows16, int w_o, int w_i, int mode, int i_off, int c_out, int c1,
                                                        int c2, int nchunk1, int nchunk, int BN, int ks, int npl, const float* __restrict__ amax_w) {
    // one thread per (n-tile, chunk, tap, co, k) element; writes its three planes
    if (idx >= total_rows16) return;
    size_t t = idx;
    const int k = t % 16; t /= 16;
    const int T = ks * ks;
    const int j = t % BN; t /= BN;
    const int tap = t % T; t /= T;
    const int q = t % nchunk;
    const int nt = t / nchunk;
    const int co = nt * BN + j;
    const int ky = tap / ks, kx = tap % ks;
    int cin = -1;
    if (q < nchunk1) { const int c = q * 16 + k; if (c < c1) cin = c; }
    else { const int c = (q - nchunk1) * 16 + k; if (c < c2) cin = c1 + c; }
    float v = 0.f;
    if (cin >= 0 && co < c_out) {
        if (mode == RCF_W_FORWARD) v = w[(((size_t)co * w_i + cin) * ks + ky) * ks + kx];
        else v = w[(((size_t)cin * w_i + (i_off + co)) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];
    }
    const unsigned x0 = __float_as_uint(v) & 0xffff0000u;
    const float r1 = v - __uint_as_float(x0);
    const unsigned x1 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(x1);
    const unsigned x2 = __float_as_uint(r2);
    const size_t chunk_elems = (size_t)npl * T * BN * 16;
    const size_t plane_elems = (size_t)ks * BN * 16;
    const size_t piece_elems = npl * plane_elems;
    const int kk = ((k >> 3) ^ ((j >> 3) & 1)) * 8 + (k & 7);   // XOR-swizzle the 16-B halves: conflict-free ds_read_b128
    const size_t base = ((size_t)nt * nchunk + q) * chunk_elems + (size_t)ky * piece_elems + ((size_t)kx * BN + j) * 16 + kk;
    if (npl == 1) {   // bf16 operands: round to nearest even
        const unsigned u = __float_as_uint(v);
        dst[base] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        return;
    }
    dst[base] = (unsigned short)(x0 >> 16);
    if (npl == 2) {   // RCF_PREC_F16X2: two fp16 planes of w * (the weight tensor's power-of-two scale)
        const float sw = amax_w ? rcf_scale_of_amax(*amax_w).s : 1.f;
        const rcf_f16_pair q = rcf_f16_planes(v * sw, 0.f);
        dst[base] = (unsigned short)q.p0;
        dst[base + plane_elems] = (unsigned short)q.p1;
        return;
    }
    dst[base + plane_elems] = (unsigned short)(x1 >> 16);
    dst[base + 2 * plane_elems] = (unsigned short)(x2 >> 16);
}

__global__ void pack_weights_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, size_t total_rows16, int w_o,
                                          int w_i, int mode, int i_off, int c_out, int c1, int c2, int nchunk1, int nchunk, int BN,
                                          int ks, int npl, const float* __restrict__ amax_w) {
    pack_weights_split_body((size_t)blockIdx.x * blockDim.x + threadIdx.x, w, dst, total_rows16, w_o, w_i, mode, i_off, c_out, c1, c2,
                            nchunk1, nchunk, BN, ks, npl, amax_w);
}

// Many weight packings in ONE launch (rcf_conv2d_pack_weights_batch): the per-item arguments travel by value in the kernel
// argument block (<= 4 KB), a workgroup finds its item by its block index.  `split` items run pack_weights_split_body, the others
// pack_weights_body.

struct PackArgs {
    const float* w;
    void* dst;
    const float* amax;   // RCF_PREC_F16X2 split items: max|w| of the weight tensor (device, nullable)
    unsigned long long total;
    int w_o, w_i, ks, mode, i_off, c_out, c1, c2, nchunk1, nchunk, T, ksx, BN, CK, kind, npl, split;
};
constexpr int PACK_BATCH = 34;
struct PackBatch {
    int n;
    unsigned blk_start[PACK_BATCH + 1];
    PackArgs it[PACK_BATCH];
};
static_assert(sizeof(PackBatch) <= 4096, "the batch travels as kernel arguments");

// ------------------------------------------------------------------------------------------------
// Weight gradient.  GEMM view: dW[k][co] = sum_pixels A[pixel][k] * dZ[pixel][co]; MFMA rows i = 32
// consecutive k of one tap (32 input channels of one halo pixel, contiguous in the LDS halo tile),
// columns j = 32 output channels, the MFMA's reduction index = pixel (lane half h takes pixel 2t+h).
// A workgroup walks many spatial tiles (persistent over gridDim.x splits) for one (k-chunk, 32-co group);
// wave w reduces the tile rows of slice w, the 4 slices are summed through LDS at the end, and one partial
// [T*32][32] per workgroup goes to the workspace (reduced deterministically by wgrad_reduce_kernel).
template <int KSY_, int KSX_, int XEXTRA_, int LSTEP_, int CST_, int STRP_, int PX_, int TH_, int MINW_, int PREFETCH_ = (MINW_ == 1)>
struct WgCfg {
    static constexpr bool PREFETCH = PREFETCH_ != 0;   // stage tile t+1 through registers during the MFMAs of tile t
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

template <class C, class SX = SInOf<C>, class SD = SAct>
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

    using H = Halo<C::CST, C::STRP, C::HXP, C::HYP>;
    constexpr int ND = (C::TP * 8) / 256;     // dZ tile: TP pixels x 8 float4
    H halo;
    f32x4 ra[H::NA];
    f32x4 rd[ND];
    const int dc = co0 + (tid & 7) * 4;

    auto load_tile = [&](int tile) {
        int t = tile;
        const int tx = t % a.tiles_x;
        t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        const int img = t / a.tiles_y;
        const int oy0 = ty * C::TH;
        const int ox0 = tx * C::PX;
        halo.setup(hs, ws, gmode, img, oy0 * a.stride - a.pad, ox0 * a.stride - a.pad_x, a.gstep, a.h_in, a.w_in, a.sy, a.sx, tid,
                   a.ioy, a.iox);
        halo.template load<SX>(ra, src, csrc, cb, tid);
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int p = (tid >> 3) + i * 32;
            const int oy = oy0 + p / C::PX;
            const int ox = ox0 + p % C::PX;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const int py = oy * a.os + a.ooy, px = ox * a.os + a.oox;
            if (oy < a.h_out && ox < a.w_out && py < a.ohp && px < a.owp && dc < a.c_out)
                v = rcf_ld4<SD>(a.dz, (((size_t)img * a.ohp + py) * a.owp + px) * a.c_out + dc);
            rd[i] = v;
        }
    };

    if (C::PREFETCH && (int)blockIdx.x < a.ntiles) load_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        if (!C::PREFETCH) load_tile(tile);   // two workgroups per CU cover each other's staging instead
        __syncthreads();
        halo.store(ra, As, tid);
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int p = (tid >> 3) + i * 32;
            *reinterpret_cast<f32x4*>(Ds + p * 32 + (tid & 7) * 4) = rd[i];
        }
        __syncthreads();
        if (C::PREFETCH && tile + (int)gridDim.x < a.ntiles) load_tile(tile + gridDim.x);   // in flight while the MFMAs below run

        // MFMA loop over this wave's RS tile rows x PX/2 pixel pairs; operands of iteration it+1 are read from LDS while
        // the T MFMAs of iteration it issue (one wave per SIMD: nothing else would cover the LDS latency)
        {
            constexpr int NIT = C::RS * (C::PX / 2);
            float av[2][C::T], bv[2];
            auto fetch = [&](int it, int slot) {
                const int trow = wave * C::RS + it / (C::PX / 2);
                const int pc = 2 * (it % (C::PX / 2)) + lh;
                bv[slot] = Ds[(trow * C::PX + pc) * 32 + li];
                const int hb = (trow * C::LSTEP * C::HXP + pc * C::LSTEP) * C::STRP + li;
#pragma unroll
                for (int tap = 0; tap < C::T; ++tap)
                    av[slot][tap] = As[hb + ((tap / C::KSX) * C::HXP + (tap % C::KSX)) * C::STRP];
            };
            fetch(0, 0);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int cur = it & 1;
                if (it + 1 < NIT) fetch(it + 1, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);   // keep the next operands' reads ahead of this iteration's MFMAs
#pragma unroll
                for (int tap = 0; tap < C::T; ++tap)
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][tap], bv[cur], acc[tap], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
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

// Weight gradient, LDS-DMA variant (all non-stem layers): the halo tile [pixel][32 ch] and the dZ tile [pixel][32 co] are
// unpadded, so each wave-level global_load_lds_dwordx4 moves 8 whole pixels (1 KiB) straight from HBM/L2 into LDS with
// per-lane source addresses (padding / out-of-image lanes read a zero page).  No staging registers and no ds_write pass:
// the kernel fits two workgroups per CU, which cover each other's DMA latency and address arithmetic.
template <class C>
__global__ void __launch_bounds__(256, 2) conv_wgrad_dma_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using H = Halo<C::CST, C::STRP, C::HXP, C::HYP>;
    static_assert(C::CST == 32 && C::STRP == 32, "DMA path needs the unpadded [pixel][32] tile");
    constexpr int A_DMA_FLOATS = H::NA * H::PPI * 32;   // every lane of every instruction lands somewhere
    constexpr int ND = (C::TP * 8) / 256;
    float* As = smem;
    float* Ds = smem + A_DMA_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31;
    const int lh = lane >> 5;

    const int q = blockIdx.y;
    const int co0 = blockIdx.z * 32;
    const bool first = q < a.nchunk1;
    const float* src = first ? a.in1 : a.in2;
    const int csrc = first ? a.c1 : a.c2;
    const int cb = (first ? q : q - a.nchunk1) * 32;
    const int hs = first ? a.h1 : a.h_in;
    const int ws = first ? a.w1 : a.w_in;
    const int gmode = first ? a.gather1 : RCF_GATHER_DIRECT;
    const int cch = cb + (tid & 7) * 4;
    const bool cok = cch < csrc;
    const int dc = co0 + (tid & 7) * 4;
    const bool dok = dc < a.c_out;

    f32x16 acc[C::T];
#pragma unroll
    for (int tap = 0; tap < C::T; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tap][r] = 0.f;

    H halo;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % a.tiles_x;
        t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        const int img = t / a.tiles_y;
        const int oy0 = ty * C::TH;
        const int ox0 = tx * C::PX;
        halo.setup(hs, ws, gmode, img, oy0 * a.stride - a.pad, ox0 * a.stride - a.pad_x, a.gstep, a.h_in, a.w_in, a.sy, a.sx, tid,
                   a.ioy, a.iox, a.vt, a.hp, a.inv_hp, a.nimg);
        __syncthreads();   // the previous tile's MFMAs are done with LDS
#pragma unroll
        for (int i = 0; i < H::NA; ++i) {
            const float* g = (cok && halo.pix[i] >= 0) ? src + (size_t)halo.pix[i] * csrc + cch : a.zero;
            float* dst = As + (i * H::PPI + wave * 8) * 32;   // wave-uniform; lane l lands 16*l bytes further
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int p = (tid >> 3) + i * 32;
            int oy = oy0 + p / C::PX;
            const int ox = ox0 + p % C::PX;
            int im = img;
            if (a.vt) {
                im = (int)(((float)oy + 0.5f) * a.inv_hp);
                oy -= im * a.hp;
                if (im >= a.nimg) oy = a.h_out;
            }
            const int py = oy * a.os + a.ooy, px = ox * a.os + a.oox;
            const bool ok = dok && oy < a.h_out && ox < a.w_out && py < a.ohp && px < a.owp;
            const float* g = ok ? a.dz + (((size_t)im * a.ohp + py) * a.owp + px) * a.c_out + dc : a.zero;
            float* dst = Ds + (i * 32 + wave * 8) * 32;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
        rcf_wait_dma();
        __syncthreads();

        {
            constexpr int NIT = C::RS * (C::PX / 2);
            float av[2][C::T], bv[2];
            auto fetch = [&](int it, int slot) {
                const int trow = (tid >> 6) * C::RS + it / (C::PX / 2);
                const int pc = 2 * (it % (C::PX / 2)) + lh;
                bv[slot] = Ds[(trow * C::PX + pc) * 32 + li];
                const int hb = (trow * C::LSTEP * C::HXP + pc * C::LSTEP) * C::STRP + li;
#pragma unroll
                for (int tap = 0; tap < C::T; ++tap)
                    av[slot][tap] = As[hb + ((tap / C::KSX) * C::HXP + (tap % C::KSX)) * C::STRP];
            };
            fetch(0, 0);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int cur = it & 1;
                if (it + 1 < NIT) fetch(it + 1, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tap = 0; tap < C::T; ++tap)
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][tap], bv[cur], acc[tap], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // sum the 4 waves' accumulators through LDS (wave 3 -> 2 -> 1 -> 0 chain keeps it deterministic)
    float* red = smem;
    const int wv = tid >> 6;
    for (int s = 3; s >= 1; --s) {
        __syncthreads();
        if (wv == s) {
#pragma unroll
            for (int tap = 0; tap < C::T; ++tap)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(tap * 16 + r) * 64 + lane] = acc[tap][r];
        }
        __syncthreads();
        if (wv == s - 1) {
#pragma unroll
            for (int tap = 0; tap < C::T; ++tap)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tap][r] += red[(tap * 16 + r) * 64 + lane];
        }
    }
    if (wv == 0) {
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

// ------------------------------------------------------------------------------------------------
// Weight gradient on the bf16 matrix pipe (3x3 stride 1; same exact 3-way split and six partial products as conv_split_kernel).
// The reduction index of dW[ci][tap][co] = sum_pixels x[p + tap][ci] * dz[p][co] is the PIXEL, and v_mfma_f32_32x32x16_bf16 wants
// 8 consecutive reduction elements per lane: both operands are therefore staged CHANNEL-MAJOR in LDS, [plane][channel][row][x]
// bf16, so that 8 consecutive pixels of one channel are one ds_read_b128.  A tile is 16 x 8 output pixels; one MFMA step is one
// tile row (lane half h takes pixels 8h..8h+7).  The kx = 1 operand is built from the aligned 16 B + the next 4 B with
// v_alignbit; kx = 2 is the same five dwords offset by one register.
// One workgroup per CU (4 waves, one per SIMD, up to 512 VGPRs): wave (wi, wj, wk) owns the 32 ci x 32 co x 9 tap accumulators
// (144 VGPRs) of ci-block wi and co-block wj and reduces the tile rows wk, wk + KSPLIT, ...  The next step's 21 LDS reads and this
// step's 36 v_alignbit are issued one at a time between the 54 MFMAs of a step (hand order, pinned with sched_barrier).
// LDS rows are permuted (row = (ch % 4) * (N / 4) + ch / 4) so that the transposing ds_write_b128 of adjacent lanes (adjacent
// channel quads) land on adjacent rows: with a row pitch of 16 B mod 128 B both the writes and the reads are conflict-free.
constexpr int ws_pitch(int bytes) { return ((bytes - 16 + 127) / 128) * 128 + 16; }   // >= bytes and == 16 (mod 128)

// PAIR_ (KS_ = 2; the up-2x weight gradient, phase_sum == 2): phases (a, 0) and (a, 1) of an up-2x convolution from ONE x tile.
// Phase (a, b) pairs x(y - (1 - a) + ty, x - (1 - b) + tx) with dz(2y + a, 2x + b): with the x tile staged at pad (1 - a, 1) the two
// phases read its columns kx = b + tx in {0, 1, 2} -- the three operand shifts the 3x3 kernel builds anyway -- against their own dz
// tile: eight accumulators (ty, {(kx 0, b 0), (kx 1, b 0), (kx 1, b 1), (kx 2, b 1)}), x loaded, split and transposed once for both.
template <int WCI_, int WCO_, int KS_ = 3, int TH_ = 8, int NPL_ = 3, bool PAIR_ = false>
struct WsCfg {
    static constexpr bool PAIR = PAIR_;
    static_assert(!PAIR_ || KS_ == 2, "PAIR is the up-2x weight gradient: 2x2 taps");
    static constexpr int KS = KS_, T = PAIR_ ? 8 : KS_ * KS_;
    static constexpr int KXN = PAIR_ ? 3 : KS_;       // column shifts of a tile row's x operand
    static constexpr int NDP = PAIR_ ? 2 : 1;         // dz phase tiles
    static constexpr int NPL = NPL_, NP = NPL_ == 3 ? 6 : (NPL_ == 2 ? 3 : 1);   // operand planes / partial products (1: RCF_PREC_BF16, 2: RCF_PREC_F16X2)
    static constexpr int WCI = WCI_, WCO = WCO_, KSPLIT = 4 / (WCI_ * WCO_);
    static constexpr int NCI = 32 * WCI_, NCO = 32 * WCO_;
    static constexpr int PX = 16, TH = TH_, HXP = PX + KXN - 1, HYP = TH + KS - 1;
    static constexpr int XROW = 48, DROW = 32;     // bytes per tile row of one channel: 24 px (18 used) / 16 px
    static constexpr int SX = ws_pitch(HYP * XROW), SD = ws_pitch(TH * DROW);   // bytes per channel and plane
    static constexpr int XPL = NCI * SX, DPL = NCO * SD;
    static constexpr int X_BYTES = NPL * XPL, D_BYTES = NDP * NPL * DPL;
    static constexpr int RED_BYTES = (KSPLIT > 1) ? WCI * WCO * T * 16 * 64 * 4 : 0;
    static constexpr int COEF_BYTES = 2 * NCI * 4;   // BN-on-load table of this workgroup's input channels: scale, shift
    static constexpr int LDS_BYTES = ((X_BYTES + D_BYTES) > RED_BYTES ? (X_BYTES + D_BYTES) : RED_BYTES) + COEF_BYTES;
    static constexpr int CQX = 8 * WCI, CQD = 8 * WCO;
    static constexpr int NXU = HYP * 3 * CQX, NDU = NDP * TH * 2 * CQD;     // 8-pixel x 4-channel staging units
    static constexpr int RX = (NXU + 255) / 256, RD = (NDU + 255) / 256;
    static constexpr int NS = TH / KSPLIT;         // MFMA steps (tile rows) per wave and tile
    static_assert(HYP * XROW <= SX && TH * DROW <= SD && LDS_BYTES <= 160 * 1024, "tile rows must fit the channel pitch / LDS");
};

template <class C, class SX = SAct, class SD = SAct>
__global__ void __launch_bounds__(256, 1) conv_wgrad_split_kernel(ConvArgs a) {
    static_assert(!SX::B16 || C::NPL == 1, "bf16 tensors go with bf16 operands");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* Xs = smem_b;
    unsigned char* Ds = smem_b + C::X_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31;
    const int lh = lane >> 5;
    const int wi = wave % C::WCI, wj = (wave / C::WCI) % C::WCO, wk = wave / (C::WCI * C::WCO);

    const int Q = blockIdx.y;
    const bool first = Q < a.nchunk1;
    const float* src = first ? a.in1 : a.in2;
    const int csrc = first ? a.c1 : a.c2;
    const int cb = (first ? Q : Q - a.nchunk1) * C::NCI;   // first channel of this chunk inside its source
    const int hs = first ? a.h1 : a.h_in;
    const int ws = first ? a.w1 : a.w_in;
    const int gmode = first ? a.gather1 : RCF_GATHER_DIRECT;
    const int co0 = blockIdx.z * C::NCO;
    // slot of this workgroup (it owns tiles slot, slot + nslot, ... and one workspace row block) and the phase geometry.
    // phase_sum == 2: the four phase weight gradients of an up-2x convolution in ONE launch -- workgroup (slot, phase) is an ordinary
    // phase launch's workgroup (pad 1 - a, 1 - b on x; dz read at (2y + a, 2x + b)), and the four phases of a slot sit on the SAME XCD
    // (blocks are dealt to the eight XCDs round-robin: block b -> XCD b % 8, phase (b / 8) % 4, slot b % 8 + 8 (b / 32)), working through
    // the same x tiles at the same pace: x comes from HBM once and from that XCD's L2 three times instead of four times from HBM
    // phase_sum == 1: the four phase weight gradients of a 3x3 stride-2 convolution the same way -- workgroup (slot, phase) gathers x
    // at (2y + a, 2x + b) against the SAME dz tile, which the four phases of a slot then share in their XCD's L2
    int slot = blockIdx.x, nslot = gridDim.x;
    int pad_y = a.pad, pad_x = a.pad_x, ooy = a.ooy, oox = a.oox, g_ioy = a.ioy, g_iox = a.iox;
    int wslot = blockIdx.x;
    if (C::PAIR) {   // workgroup (slot, a): phases (a, 0) and (a, 1); the two a of a slot share an XCD
        nslot = gridDim.x >> 1;
        int pa;
        if ((nslot & 7) == 0) { pa = (blockIdx.x >> 3) & 1; slot = (blockIdx.x & 7) + 8 * (blockIdx.x >> 4); }
        else { pa = blockIdx.x & 1; slot = blockIdx.x >> 1; }
        pad_y = 1 - pa; pad_x = 1; ooy = pa; oox = 0;
        wslot = 2 * pa * nslot + slot;   // phase (a, b): row block (2 a + b) * nslot + slot
    } else
    if (a.phase_sum != 0) {
        nslot = gridDim.x >> 2;
        int ph;
        if ((nslot & 7) == 0) { ph = (blockIdx.x >> 3) & 3; slot = (blockIdx.x & 7) + 8 * (blockIdx.x >> 5); }
        else { ph = blockIdx.x & 3; slot = blockIdx.x >> 2; }
        if (a.phase_sum == 2) { pad_y = 1 - (ph >> 1); pad_x = 1 - (ph & 1); ooy = ph >> 1; oox = ph & 1; }
        else { g_ioy = ph >> 1; g_iox = ph & 1; }
        wslot = ph * nslot + slot;
    }

    f32x16 acc[C::T];
#pragma unroll
    for (int tap = 0; tap < C::T; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tap][r] = 0.f;
    SplitScales sc = {1.f, 1.f, 1.f, 1.f};   // NPL == 2: scales of x (A operand) and dz (B operand)
    if constexpr (C::NPL == 2) sc = rcf_split_scales(a.amax_a1, a.amax_a2, a.amax_b);

    // ---- staging: fp32 [pixel][channel] in HBM -> registers (next tile, during this tile's MFMAs) -> bf16 planes in LDS
    // BN-on-load (x is a raw conv output, y = lrelu(z * scale + shift) applied while staging): coefficient table in LDS
    const float* cfx = first ? a.coef1 : a.coef2;
    float* coef_lds = reinterpret_cast<float*>(smem_b + C::LDS_BYTES - C::COEF_BYTES);
    if (cfx != nullptr) {
        for (int i = tid; i < C::NCI; i += 256) {
            const bool okc = cb + i < csrc;
            coef_lds[i] = okc ? cfx[cb + i] : 1.f;
            coef_lds[C::NCI + i] = okc ? cfx[csrc + cb + i] : 0.f;
        }
    }
    unsigned mx[C::RX];   // BN-on-load only: bit j = pixel j of the unit is real data (padding must stay 0 after the transform)
    // staging registers: fp32 tensors -> 4 channels as f32x4; bf16 tensors -> the same 4 channels RAW (two dwords: half the
    // registers, no conversion; the transpose below picks 16-bit halves with one v_perm per pixel pair)
    constexpr bool RAWX = SX::B16, RAWD = SD::B16;
    using RXT = std::conditional_t<RAWX, u32x2, f32x4>;
    using RDT = std::conditional_t<RAWD, u32x2, f32x4>;
    RXT rx[C::RX][8];
    RDT rd[C::RD][8];
    // loads are branch-free: padding / out-of-image / out-of-range-channel elements read a zero page (rcf_zero_page), so the values need
    // no masking afterwards
    // FAST (compile-time): plain layers -- source read as is, one image per tile, unit output stride -- address their pixels with an
    // add and a compare each; the general path (nearest-upsample gather, virtual tall image, strided phase outputs) costs ~3x the
    // VALU instructions per load, and a VALU wave-instruction costs about a fifth of an MFMA in energy (these kernels are power-limited)
    auto load_tile_impl = [&](int tile, auto fast_tag) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_tag)::value;
        int t = tile;
        const int tx = t % a.tiles_x;
        t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        const int img = t / a.tiles_y;
        const int oy0 = ty * C::TH, ox0 = tx * C::PX;
        const int iy0 = oy0 - pad_y, ix0 = ox0 - pad_x;
#pragma unroll
        for (int i = 0; i < C::RX; ++i) {
            const int u = tid + 256 * i;
            const int cq = u % C::CQX, g = (u / C::CQX) % 3, hy = u / (3 * C::CQX);
            const int ch = cb + cq * 4;
            const int ly = iy0 + hy;
            bool rowok = u < C::NXU && ch < csrc;
            int rowbase;
            if (!FAST && a.vt) {
                const int im = (int)(((float)ly + 0.5f) * a.inv_hp);
                const int y = ly - im * a.hp;
                rowok = rowok && ly >= 0 && im < a.nimg && y < a.h_in;
                rowbase = (im * hs + y) * ws;
            } else {
                rowok = rowok && (unsigned)ly < (unsigned)a.h_in;
                int py = (!FAST && gmode == RCF_GATHER_NEAREST) ? min((int)floorf((float)ly * a.sy), hs - 1) : ly;
                if (!FAST && gmode == RCF_GATHER_STRIDED2) {   // one phase of the source: logical row ly is physical row 2 ly + ioy
                    py = 2 * ly + a.ioy;
                    rowok = rowok && py < hs;
                }
                rowbase = (img * hs + py) * ws;
            }
            const float* rowptr = rcf_at<SX>(src, (size_t)(rowok ? rowbase : 0) * csrc + (rowok ? ch : 0));
            unsigned m = 0u;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int hx = 8 * g + j;
                const int lx = ix0 + hx;
                bool ok = rowok && hx < C::HXP && (unsigned)lx < (unsigned)a.w_in;
                int px = (!FAST && gmode == RCF_GATHER_NEAREST) ? min((int)floorf((float)lx * a.sx), ws - 1) : lx;
                if (!FAST && gmode == RCF_GATHER_STRIDED2) {
                    px = 2 * lx + a.iox;
                    ok = ok && px < ws;
                }
                if constexpr (RAWX) rx[i][j] = *reinterpret_cast<const u32x2*>(ok ? rcf_at<SX>(rowptr, px * csrc) : a.zero);
                else rx[i][j] = rcf_ld4<SX>(ok ? rcf_at<SX>(rowptr, px * csrc) : a.zero, 0);
                if (cfx != nullptr) m |= ok ? (1u << j) : 0u;
            }
            mx[i] = m;
        }
#pragma unroll
        for (int i = 0; i < C::RD; ++i) {
            const int u = tid + 256 * i;
            const int cq = u % C::CQD, g = (u / C::CQD) % 2, r = u / (2 * C::CQD);
            const int dc = co0 + cq * 4;
            int oy = oy0 + r;
            int im = img;
            if (!FAST && a.vt) {
                im = (int)(((float)oy + 0.5f) * a.inv_hp);
                oy -= im * a.hp;
                if (im >= a.nimg) oy = a.h_out;
            }
            const int py = FAST ? oy : oy * a.os + ooy;   // strided output rows/columns of the phase convolutions
            const bool rowok = u < C::NDU && dc < a.c_out && oy < a.h_out && py < a.ohp;
            const float* rowptr = rcf_at<SD>(a.dz, (size_t)(rowok ? (im * a.ohp + py) * a.owp : 0) * a.c_out + (rowok ? dc : 0));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ox = ox0 + 8 * g + j;
                const int px = FAST ? ox : ox * a.os + oox;
                if constexpr (RAWD) rd[i][j] = *reinterpret_cast<const u32x2*>((rowok && ox < a.w_out && px < a.owp) ? rcf_at<SD>(rowptr, px * a.c_out) : a.zero);
                else rd[i][j] = rcf_ld4<SD>((rowok && ox < a.w_out && px < a.owp) ? rcf_at<SD>(rowptr, px * a.c_out) : a.zero, 0);
            }
        }
    };
    // Every source but the nearest-upsample gather goes through buffer descriptors (round 5): per tile and staging unit three lane
    // registers -- the byte offset of the unit's first pixel from the start of the tile's first image, its first column, and the
    // number of valid columns (0 for a row outside the image) -- and per load an add, a compare and a select; a pixel outside gets the
    // offset 0xffffffff, which the buffer unit answers with zeros.  (The pointer form above costs ~10 VALU instructions per load:
    // 16-26 % of a wave's time went into "address arithmetic + load issue" with fp32 tensors, 23-46 % with bf16 ones,
    // tools/phase_timing_wgrad.py -- at one wave per SIMD nothing hides it.)
    auto load_tile_buf = [&](int tile) __attribute__((always_inline)) {
        int t = tile;
        const int tx = t % a.tiles_x;
        t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        const int img = t / a.tiles_y;
        const int oy0 = ty * C::TH, ox0 = tx * C::PX;
        const int iy0 = oy0 - pad_y, ix0 = ox0 - pad_x;
        const bool s2 = gmode == RCF_GATHER_STRIDED2;
        const int pm = s2 ? 2 : 1;                                      // source pixels per logical pixel
        const int ioy = s2 ? g_ioy : 0, iox = s2 ? g_iox : 0;
        // ---- x: halo tile of the input
        {
            int fimg = img;
            if (a.vt) {
                fimg = (int)(((float)(iy0 < 0 ? 0 : iy0) + 0.5f) * a.inv_hp);
                fimg = fimg < a.nimg ? fimg : a.nimg - 1;
            }
            fimg = __builtin_amdgcn_readfirstlane(fimg);
            const unsigned pixb = (unsigned)csrc * SX::BYTES;           // bytes per source pixel
            const unsigned rowb = (unsigned)ws * pixb;
            const __amdgpu_buffer_rsrc_t rsx = rcf_rsrc(reinterpret_cast<const unsigned char*>(src) + (size_t)fimg * hs * rowb);
            int wl = a.w_in;                                             // logical columns that exist in the source
            if (s2) { const int wph = (ws - iox + 1) / 2; wl = wl < wph ? wl : wph; }
            const unsigned stepb = (unsigned)pm * pixb;
            const unsigned cbb = (unsigned)cb * SX::BYTES;
#pragma unroll
            for (int i = 0; i < C::RX; ++i) {
                const int u = tid + 256 * i;
                const int cq = u % C::CQX, g = (u / C::CQX) % 3, hy = u / (3 * C::CQX);
                const int ly = iy0 + hy;
                bool rowok = u < C::NXU && cb + cq * 4 < csrc;
                int im = img, y = ly;
                if (a.vt) {
                    im = (int)(((float)ly + 0.5f) * a.inv_hp);
                    y = ly - im * a.hp;
                    rowok = rowok && ly >= 0 && im < a.nimg && y < a.h_in;
                } else {
                    rowok = rowok && (unsigned)ly < (unsigned)a.h_in;
                }
                const int py = pm * y + ioy;
                rowok = rowok && py < hs;
                const int lx0 = ix0 + 8 * g;                             // logical column of the unit's first pixel
                const unsigned lim = rowok ? (unsigned)wl : 0u;
                const unsigned v0 = (unsigned)(((im - fimg) * hs + py)) * rowb + (unsigned)(pm * lx0 + iox) * pixb + (unsigned)(cq * 4) * SX::BYTES;
                unsigned m = 0u;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool ok = (unsigned)(lx0 + j) < lim && 8 * g + j < C::HXP;
                    const unsigned vo = ok ? v0 + (unsigned)j * stepb : 0xffffffffu;
                    if constexpr (RAWX) rx[i][j] = rcf_buffer_load_u32x2(rsx, vo, cbb);
                    else rx[i][j] = rcf_buffer_load_f32x4(rsx, vo, cbb);
                    if (cfx != nullptr) m |= ok ? (1u << j) : 0u;
                }
                mx[i] = m;
            }
        }
        // ---- dz: the output-gradient tile (strided rows / columns for the phase convolutions)
        {
            int fimg = img;
            if (a.vt) {
                fimg = (int)(((float)oy0 + 0.5f) * a.inv_hp);
                fimg = fimg < a.nimg ? fimg : a.nimg - 1;
            }
            fimg = __builtin_amdgcn_readfirstlane(fimg);
            const unsigned pixb = (unsigned)a.c_out * SD::BYTES;
            const unsigned rowb = (unsigned)a.owp * pixb;
            const __amdgpu_buffer_rsrc_t rsd = rcf_rsrc(reinterpret_cast<const unsigned char*>(a.dz) + (size_t)fimg * a.ohp * rowb);
            const unsigned stepb = (unsigned)a.os * pixb;
#pragma unroll
            for (int i = 0; i < C::RD; ++i) {
                const int u = tid + 256 * i;
                const int cq = u % C::CQD, g = (u / C::CQD) % 2, r = (u / (2 * C::CQD)) % C::TH;
                const int uox = C::PAIR ? u / (2 * C::CQD * C::TH) : oox;   // PAIR: the unit's phase b = its dz tile
                int wl = a.w_out;
                { const int wph = (a.owp - uox + a.os - 1) / a.os; wl = wl < wph ? wl : wph; }
                int oy = oy0 + r, im = img;
                bool rowok = u < C::NDU && co0 + cq * 4 < a.c_out;
                if (a.vt) {
                    im = (int)(((float)oy + 0.5f) * a.inv_hp);
                    oy -= im * a.hp;
                    rowok = rowok && im < a.nimg;
                }
                const int py = oy * a.os + ooy;
                rowok = rowok && oy < a.h_out && py < a.ohp;
                const int lx0 = ox0 + 8 * g;
                const unsigned lim = rowok ? (unsigned)wl : 0u;
                const unsigned v0 = (unsigned)((im - fimg) * a.ohp + py) * rowb + (unsigned)(lx0 * a.os + uox) * pixb + (unsigned)(co0 + cq * 4) * SD::BYTES;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned vo = (unsigned)(lx0 + j) < lim ? v0 + (unsigned)j * stepb : 0xffffffffu;
                    if constexpr (RAWD) rd[i][j] = rcf_buffer_load_u32x2(rsd, vo, 0u);
                    else rd[i][j] = rcf_buffer_load_f32x4(rsd, vo, 0u);
                }
            }
        }
    };
    auto load_tile = [&](int tile) __attribute__((always_inline)) {
        if (gmode == RCF_GATHER_NEAREST) load_tile_impl(tile, std::false_type{});
        else load_tile_buf(tile);
    };
    // 8 pixels of one channel -> three 16-B bf16 vectors (exact truncation split), written to the channel's LDS row
    // bf16 tensors: 8 pixels of channel e (raw dwords e >> 1, half e & 1) -> one 16-B vector of the channel's LDS row
    auto pack8 = [&](const u32x2 (&v)[8], int e, unsigned char* dst) {
        u32x4 w0;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            w0[d] = __builtin_amdgcn_perm(v[2 * d + 1][e >> 1], v[2 * d][e >> 1], (e & 1) ? 0x07060302u : 0x05040100u);
        *reinterpret_cast<u32x4*>(dst) = w0;
    };
    auto split8 = [&](const f32x4 (&v)[8], int e, unsigned char* dst, int plane_bytes, float scale) {
        u32x4 w0, w1, w2;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            unsigned x0[2], x1[2], x2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float x = v[2 * d + h][e];
                x0[h] = __float_as_uint(x) & 0xffff0000u;
                const float r1 = x - __uint_as_float(x0[h]);
                x1[h] = __float_as_uint(r1) & 0xffff0000u;
                const float r2 = r1 - __uint_as_float(x1[h]);
                x2[h] = __float_as_uint(r2);
            }
            if (C::NPL == 1) {   // bf16 operands: round to nearest even instead of splitting
                w0[d] = __builtin_amdgcn_perm(rcf_bf16_rne(v[2 * d + 1][e]), rcf_bf16_rne(v[2 * d][e]), 0x07060302u);
            } else {
                w0[d] = __builtin_amdgcn_perm(x0[1], x0[0], 0x07060302u);   // high halves of the pixel pair
                w1[d] = __builtin_amdgcn_perm(x1[1], x1[0], 0x07060302u);
                w2[d] = __builtin_amdgcn_perm(x2[1], x2[0], 0x07060302u);
                if constexpr (C::NPL == 2) {   // two fp16 planes of the scaled values
                    const rcf_f16_pair q = rcf_f16_planes(v[2 * d][e] * scale, v[2 * d + 1][e] * scale);
                    w0[d] = q.p0;
                    w1[d] = q.p1;
                }
            }
        }
        *reinterpret_cast<u32x4*>(dst) = w0;
        if (C::NPL >= 2) *reinterpret_cast<u32x4*>(dst + plane_bytes) = w1;
        if (C::NPL == 3) *reinterpret_cast<u32x4*>(dst + 2 * plane_bytes) = w2;
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < C::RX; ++i) {
            const int u = tid + 256 * i;
            if (C::NXU % 256 == 0 || u < C::NXU) {
                const int cq = u % C::CQX, g = (u / C::CQX) % 3, hy = u / (3 * C::CQX);
                if constexpr (!RAWX)
                if (cfx != nullptr) {   // the producer's BatchNorm + LeakyReLU on the 8 x 4 values; padding stays zero
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(coef_lds + cq * 4);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(coef_lds + C::NCI + cq * 4);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            rx[i][j][e] = ((mx[i] >> j) & 1u) ? rcf_lrelu(rx[i][j][e] * sc[e] + sh[e]) : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if constexpr (RAWX) pack8(rx[i], e, Xs + (e * C::CQX + cq) * C::SX + hy * C::XROW + g * 16);
                    else split8(rx[i], e, Xs + (e * C::CQX + cq) * C::SX + hy * C::XROW + g * 16, C::XPL, sc.sa);
            }
        }
#pragma unroll
        for (int i = 0; i < C::RD; ++i) {
            const int u = tid + 256 * i;
            if (C::NDU % 256 == 0 || u < C::NDU) {
                const int cq = u % C::CQD, g = (u / C::CQD) % 2, r = (u / (2 * C::CQD)) % C::TH;
                unsigned char* Dp = Ds + (C::PAIR ? (u / (2 * C::CQD * C::TH)) * (C::NPL * C::DPL) : 0);   // [phase b][plane][channel][row]
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if constexpr (RAWD) pack8(rd[i], e, Dp + (e * C::CQD + cq) * C::SD + r * C::DROW + g * 16);
                    else split8(rd[i], e, Dp + (e * C::CQD + cq) * C::SD + r * C::DROW + g * 16, C::DPL, sc.sb);
            }
        }
    };

    // ---- MFMA operands of one step: raw x rows (3 planes x 3 kernel rows x 5 dwords) and dz (3 planes x 4 dwords)
    const unsigned char* xb = Xs + (wi * 32 + li) * C::SX + lh * 16 + wk * C::XROW;
    const unsigned char* db = Ds + (wj * 32 + li) * C::SD + lh * 16 + wk * C::DROW;
    // x rows live in a ring of register slots: with KSPLIT == 1 consecutive steps share two of their three halo rows, so only
    // ONE new row is read per step (4 slots); otherwise two sets of three rows.  (256 architectural VGPRs hold the staging
    // registers, the operands and the addresses; the 144 accumulators live in AGPRs.)
    constexpr int KS = C::KS;
    constexpr int ROLL = C::KSPLIT == 1;
    constexpr int NSLOT = ROLL ? KS + 1 : 2 * KS;
    constexpr int NEWROWS = ROLL ? 1 : KS;          // halo rows fetched per step
    constexpr int NPL = C::NPL, NP = C::NP;
    constexpr int NDP = C::NDP;
    constexpr int NRD = NEWROWS * 2 * NPL + NDP * NPL;    // LDS reads per step: (b128 + b32) x planes per row, + planes of dz (per phase tile)
    constexpr int NMF = NP * C::T;                  // MFMAs per step
    u32x4 xlo[NSLOT][NPL];     // [row slot][plane]  pixels 8h .. 8h+7
    unsigned xhi[NSLOT][NPL];  //                    pixels 8h+8, 8h+9
    u32x4 dzv[2][NDP][NPL];    // [set][phase tile][plane]
    u32x4 xs1[KS][NPL];        // kx = 1 operands of the current step

    int tile = slot;
#ifdef RCF_PHASE_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    if (tile < a.ntiles) load_tile(tile);
    while (tile < a.ntiles) {
        RCF_T(t_g0);
        __syncthreads();   // the previous tile's MFMAs are done with LDS
        RCF_T(t_g1);
        RCF_TACC(0, t_g1, t_g0);   // 0: barrier "LDS free"
        store_tile();
        RCF_T(t_g2);
        RCF_TACC(1, t_g2, t_g1);   // 1: wait for the tile's global loads + transpose into LDS
        __syncthreads();
        RCF_T(t_g3);
        RCF_TACC(2, t_g3, t_g2);   // 2: publishing barrier
        const int ntile = tile + nslot;
        // (issuing these loads in slices between the MFMA steps below was tried twice -- as flat loads and, after the zero page
        // became a kernel argument, as global loads -- and is NOT faster: +-2 % with fp32 tensors, -6...-15 % with bf16 tensors)
        if (ntile < a.ntiles) load_tile(ntile);
        RCF_T(t_g4);
        RCF_TACC(3, t_g4, t_g3);   // 3: address arithmetic + global-load issue of the next tile

        // prologue of the tile: operands of this wave's first row
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                xlo[ky][pl] = *reinterpret_cast<const u32x4*>(xb + pl * C::XPL + ky * C::XROW);
                xhi[ky][pl] = *reinterpret_cast<const unsigned*>(xb + pl * C::XPL + ky * C::XROW + 16);
            }
#pragma unroll
        for (int bq = 0; bq < NDP; ++bq)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) dzv[0][bq][pl] = *reinterpret_cast<const u32x4*>(db + (bq * NPL + pl) * C::DPL);
        __builtin_amdgcn_sched_barrier(0);

#pragma unroll
        for (int s = 0; s < C::NS; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            const bool has_next = s + 1 < C::NS;
            const int rn = (s + 1) * C::KSPLIT;   // next row of this wave (relative to wk)
#pragma unroll
            for (int j = 0; j < NMF; ++j) {
                // order: all kx = 0 taps, (kx = 2,) then kx = 1, whose operands are being built meanwhile; inside a group the six
                // partial products run smallest first and the kernel rows alternate
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
                constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};
                constexpr int KXO[3] = {0, KS == 3 ? 2 : 1, 1};
                // PAIR: four groups m = (kx, b) in the order (0, 0), (2, 1), then the two that need the shifted operand: (1, 0), (1, 1)
                constexpr int MORD[4] = {0, 3, 1, 2}, KXM[4] = {0, 1, 1, 2};
                const int grp = j / (NP * KS), pj = (j % (NP * KS)) / KS, ky = j % KS;
                const int pm = C::PAIR ? MORD[grp] : 0;
                const int kx = C::PAIR ? KXM[pm] : KXO[grp];
                const int tap = C::PAIR ? ky * 4 + pm : ky * KS + kx;
                const int bq = C::PAIR ? (pm >> 1) : 0;
                const int sl = ROLL ? (s + ky) % (KS + 1) : cur * KS + ky;
                const int pa = NPL == 3 ? PA[pj] : (NPL == 2 ? PA2[pj % 3] : 0), pbl = NPL == 3 ? PB[pj] : (NPL == 2 ? PB2[pj % 3] : 0);
                u32x4 av;
                if (kx == 0) av = xlo[sl][pa];
                else if (kx == 1) av = xs1[ky][pa];
                else {
                    const u32x4 lo = xlo[sl][pa];
                    av[0] = lo[1]; av[1] = lo[2]; av[2] = lo[3]; av[3] = xhi[sl][pa];
                }
                acc[tap] = rcf_mfma_split<NPL>(as_bf16x8(av), as_bf16x8(dzv[cur][bq][pbl]), acc[tap]);
                // one kx = 1 operand (4 v_alignbit) behind each of the first NPL * KS even (three planes) / consecutive (one plane)
                // MFMAs: all of them before the kx = 1 group starts
                constexpr int SHS = NPL == 3 ? 2 : 1;
                if (j < SHS * NPL * KS && j % SHS == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    const int sky = (j / SHS) / NPL, spl = (j / SHS) % NPL;
                    const int ssl = ROLL ? (s + sky) % (KS + 1) : cur * KS + sky;
                    const u32x4 lo = xlo[ssl][spl];
                    const unsigned hi = xhi[ssl][spl];
                    xs1[sky][spl][0] = __builtin_amdgcn_alignbit(lo[1], lo[0], 16);
                    xs1[sky][spl][1] = __builtin_amdgcn_alignbit(lo[2], lo[1], 16);
                    xs1[sky][spl][2] = __builtin_amdgcn_alignbit(lo[3], lo[2], 16);
                    xs1[sky][spl][3] = __builtin_amdgcn_alignbit(hi, lo[3], 16);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the next step's LDS reads, one at a time, spread evenly over the step.  Ring (KSPLIT == 1): only the new halo
                // row, into the slot that held row s - 1 and is free during the whole step; otherwise KS rows into the other set.
                const int n0 = (j * NRD) / NMF, n1 = ((j + 1) * NRD) / NMF;
#pragma unroll
                for (int nr = n0; nr < n1; ++nr) {
                    if (!has_next) break;
                    __builtin_amdgcn_sched_barrier(0);
                    if (nr < NEWROWS * 2 * NPL) {
                        const int rrow = (nr / 2) / NPL, rpl = (nr / 2) % NPL;
                        const int rky = ROLL ? KS - 1 : rrow;
                        const int nsl = ROLL ? (s + KS) % (KS + 1) : nxt * KS + rrow;
                        const unsigned char* g = xb + rpl * C::XPL + (rn + rky) * C::XROW;
                        if ((nr & 1) == 0) xlo[nsl][rpl] = *reinterpret_cast<const u32x4*>(g);
                        else xhi[nsl][rpl] = *reinterpret_cast<const unsigned*>(g + 16);
                    } else {
                        const int rdp = nr - NEWROWS * 2 * NPL;   // (phase tile, plane)
                        dzv[nxt][rdp / NPL][rdp % NPL] = *reinterpret_cast<const u32x4*>(db + rdp * C::DPL + rn * C::DROW);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        RCF_T(t_g5);
        RCF_TACC(4, t_g5, t_g4);   // 4: MFMA steps (+ interleaved LDS reads, operand shifts)
        tile = ntile;
    }
#ifdef RCF_PHASE_TIMING
    const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif

    if constexpr (C::NPL == 2) {   // undo the operand scales (exact: powers of two)
#pragma unroll
        for (int tap = 0; tap < C::T; ++tap)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tap][r] = acc[tap][r] * sc.ia * sc.ib;
    }
    // ---- sum the KSPLIT row slices of each (wi, wj) block through LDS (fixed order), then one partial per workgroup
    if (C::KSPLIT > 1) {
        float* red = reinterpret_cast<float*>(smem_b) + (wi + C::WCI * wj) * (C::T * 16 * 64);
        for (int s = C::KSPLIT - 1; s >= 1; --s) {
            __syncthreads();
            if (wk == s) {
#pragma unroll
                for (int tap = 0; tap < C::T; ++tap)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(tap * 16 + r) * 64 + lane] = acc[tap][r];
            }
            __syncthreads();
            if (wk == s - 1) {
#pragma unroll
                for (int tap = 0; tap < C::T; ++tap)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tap][r] += red[(tap * 16 + r) * 64 + lane];
            }
        }
    }
    if (wk == 0) {
        const int q32_0 = first ? 0 : (a.c1 + 31) / 32;   // 32-channel chunk index of this source's first chunk in k
        const int rdz = wj * 32 + li;
        const int co = co0 + (rdz % C::CQD) * 4 + rdz / C::CQD;
        constexpr int TP = KS * KS;   // taps of one phase's weight gradient (the reduction's k layout)
#pragma unroll
        for (int tap = 0; tap < C::T; ++tap) {
            // PAIR: accumulator (ty, m) belongs to phase b = m >> 1 (its own workspace row block), tap (ty, tx = m & 1)
            const int ptap = C::PAIR ? (tap >> 2) * 2 + (tap & 1) : tap;
            float* wsp = a.ws + ((size_t)wslot + (C::PAIR ? (size_t)((tap >> 1) & 1) * nslot : 0)) * a.ktot * a.cop;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rxr = wi * 32 + rcf_mfma_row(r, lh);
                const int ch = cb + (rxr % C::CQX) * 4 + rxr / C::CQX;   // channel inside its source
                if ((ch & ~31) < csrc && co < a.cop) {
                    const int k = ((q32_0 + (ch >> 5)) * TP + ptap) * 32 + (ch & 31);
                    wsp[(size_t)k * a.cop + co] = acc[tap][r];
                }
            }
        }
    }
#ifdef RCF_PHASE_TIMING
    tacc[5] = __builtin_amdgcn_s_memtime() - t_loop;   // 5: slice reduction + partial write
    tacc[7] = __builtin_amdgcn_s_memtime() - t_begin;
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&rcf_phase_cycles[i], tacc[i]);
#endif
}

#include "rcf_conv_wgrad_tr.h"

// workspace [nslot][ktot][cop] -> dW in OIHW.  One thread per (k, co), co fastest (coalesced reads).
// kind: 0 generic (k = (q*T+tap)*32 + channel-in-chunk), 1 stem (k = tap(ky)*32 + kx*4 + c).
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nslot, int ktot,
                                                           int cop, int c_out, int c1, int c2, int nchunk1, int T, int ksx, int kind) {
    __shared__ double sm[4][64];
    // blockIdx.y: phase of a four-phase weight gradient (its own workspace row blocks, its own dw[co][ci][T] slab); 0 otherwise
    ws += (size_t)blockIdx.y * nslot * ktot * cop;
    dw += (size_t)blockIdx.y * c_out * (c1 + c2) * T;
    const int lane_o = threadIdx.x & 63;     // 64 consecutive (k, co) outputs per block: coalesced 256-B reads per slot
    const int lane_s = threadIdx.x >> 6;     // 4 slot lanes
    const int idx = blockIdx.x * 64 + lane_o;
    const bool live = idx < ktot * cop;
    double s = 0.0;   // the per-workgroup partials cancel heavily for BN-followed convs: sum them in fp64
    if (live) {
        const size_t stride = (size_t)ktot * cop;
        // eight independent 256-B loads in flight per wave (the adds stay in slot order: the sum is the same)
#pragma unroll 8
        for (int sl = lane_s; sl < nslot; sl += 4) s += (double)ws[sl * stride + idx];
    }
    sm[lane_s][lane_o] = s;
    __syncthreads();
    if (lane_s != 0 || !live) return;
    s = sm[0][lane_o] + sm[1][lane_o] + sm[2][lane_o] + sm[3][lane_o];
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
    const int kw = (kind == 1) ? 7 : ksx;
    dw[(((size_t)co * (c1 + c2) + ci) * ksy_total + ky) * kw + kx] = (float)s;
}

// OIHW -> [n-tile][chunk][tap][BN][CK].  kind 0 generic, 1 stem (k = kx*4 + c, tap = ky).
__device__ __forceinline__ void pack_weights_body(size_t idx, const float* __restrict__ w, float* __restrict__ dst, size_t total, int w_o,
                                                  int w_i, int ks, int mode, int i_off, int c_out, int c1, int c2, int nchunk1, int nchunk,
                                                  int T, int ksx, int BN, int CK, int kind) {
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

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ dst, size_t total, int w_o, int w_i,
                                    int ks, int mode, int i_off, int c_out, int c1, int c2, int nchunk1, int nchunk,
                                    int T, int ksx, int BN, int CK, int kind) {
    pack_weights_body((size_t)blockIdx.x * blockDim.x + threadIdx.x, w, dst, total, w_o, w_i, ks, mode, i_off, c_out, c1, c2, nchunk1,
                      nchunk, T, ksx, BN, CK, kind);
}

__global__ void __launch_bounds__(256) pack_weights_batch_kernel(PackBatch b) {
    int i = 0;
    while (i + 1 < b.n && blockIdx.x >= b.blk_start[i + 1]) ++i;   // wave-uniform
    const PackArgs& p = b.it[i];
    const size_t idx = (size_t)(blockIdx.x - b.blk_start[i]) * 256 + threadIdx.x;
    if (p.split)
        pack_weights_split_body(idx, p.w, static_cast<unsigned short*>(p.dst), p.total, p.w_o, p.w_i, p.mode, p.i_off, p.c_out, p.c1,
                                p.c2, p.nchunk1, p.nchunk, p.BN, p.ks, p.npl, p.amax);
    else
        pack_weights_body(idx, p.w, static_cast<float*>(p.dst), p.total, p.w_o, p.w_i, p.ks, p.mode, p.i_off, p.c_out, p.c1, p.c2,
                          p.nchunk1, p.nchunk, p.T, p.ksx, p.BN, p.CK, p.kind);
}

// Phase weights (see include/rcf_hip.h).  Up-2x: output row 2y+a reads source rows {y-1,y} (a=0) / {y,y+1} (a=1); 3x3 tap ky
// lands on source tap t: a=0 -> (0,1,1), a=1 -> (0,0,1).  Stride-2 dgrad: input row 2y+a receives dZ rows y (tap 0) and y+1
// (tap 1): a=0 -> ky (1, none), a=1 -> ky (2, 0).
__device__ __forceinline__ int up2x_tap(int a, int k) { return a == 0 ? (k == 0 ? 0 : 1) : (k == 2 ? 1 : 0); }
__device__ __forceinline__ int s2_tap_k(int a, int t) { return a == 0 ? (t == 0 ? 1 : -1) : (t == 0 ? 2 : 0); }

__device__ __forceinline__ void phase_weights_body(int idx, const float* __restrict__ w, float* __restrict__ out, int O, int I, int mode) {
    const int total = 4 * O * I * 4;
    if (idx >= total) return;
    int t = idx;
    const int u = t & 1; t >>= 1;
    const int tt = t & 1; t >>= 1;
    const int OP = (mode == RCF_PHASE_UP2X_FWD) ? O : I;   // out channels of the phase conv
    const int IP = (mode == RCF_PHASE_UP2X_FWD) ? I : O;
    const int ip = t % IP; t /= IP;
    const int op = t % OP; t /= OP;
    const int b = t & 1, a = t >> 1;
    float v = 0.f;
    if (mode == RCF_PHASE_UP2X_FWD) {
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx)
                if (up2x_tap(a, ky) == tt && up2x_tap(b, kx) == u) v += w[((op * I + ip) * 3 + ky) * 3 + kx];
    } else if (mode == RCF_PHASE_UP2X_DGRAD) {   // Wd[n=i][c=o][t'][u'] = Wp[o][i][1-t'][1-u']
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx)
                if (up2x_tap(a, ky) == 1 - tt && up2x_tap(b, kx) == 1 - u) v += w[((ip * I + op) * 3 + ky) * 3 + kx];
    } else {   // RCF_PHASE_S2_DGRAD: Wd[n=i][c=o][t][u] = W[o][i][ky(a,t)][kx(b,u)]
        const int ky = s2_tap_k(a, tt), kx = s2_tap_k(b, u);
        if (ky >= 0 && kx >= 0) v = w[((ip * I + op) * 3 + ky) * 3 + kx];
    }
    out[idx] = v;
}

__global__ void phase_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int O, int I, int mode) {
    phase_weights_body(blockIdx.x * blockDim.x + threadIdx.x, w, out, O, I, mode);
}

struct PhaseArgs { const float* w; float* out; int O, I, mode, pad; };
constexpr int PHASE_BATCH = 96;
struct PhaseBatch {
    int n;
    unsigned blk_start[PHASE_BATCH + 1];
    PhaseArgs it[PHASE_BATCH];
};
static_assert(sizeof(PhaseBatch) <= 4096, "the batch travels as kernel arguments");

__global__ void __launch_bounds__(256) phase_weights_batch_kernel(PhaseBatch b) {
    int i = 0;
    while (i + 1 < b.n && blockIdx.x >= b.blk_start[i + 1]) ++i;   // wave-uniform
    const PhaseArgs& p = b.it[i];
    phase_weights_body((int)(blockIdx.x - b.blk_start[i]) * 256 + threadIdx.x, p.w, p.out, p.O, p.I, p.mode);
}

__global__ void phase_wgrad_fold_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int O, int I) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= O * I * 9) return;
    const int kx = idx % 3, ky = (idx / 3) % 3;
    const int oi = idx / 9;
    float v = 0.f;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            v += dwp[(((size_t)(a * 2 + b) * O * I + oi) * 2 + up2x_tap(a, ky)) * 2 + up2x_tap(b, kx)];
    dw[idx] = v;
}

// dW of a 3x3 stride-2 convolution from the four 2x2 weight gradients on the phase images of its input (RCF_PHASE_S2_WGRAD):
// tap k reads input row 2 oy + k - 1: k = 0 -> odd rows at oy - 1 (phase 1, tap 0), k = 1 -> even rows at oy (phase 0, tap 1),
// k = 2 -> odd rows at oy (phase 1, tap 1).  Each 3x3 tap is exactly one phase tap (9 of the 16 are real).
__global__ void phase_wgrad_gather_s2_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int O, int I) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= O * I * 9) return;
    const int kx = idx % 3, ky = (idx / 3) % 3;
    const int oi = idx / 9;
    const int a = ky == 1 ? 0 : 1, t = ky == 0 ? 0 : 1;
    const int b = kx == 1 ? 0 : 1, u = kx == 0 ? 0 : 1;
    dw[idx] = dwp[((size_t)(a * 2 + b) * O * I + oi) * 4 + t * 2 + u];
}

#ifdef RCF_CONV_KERNELS_ONLY   // probe translation units (tools/probe): the kernels above, no dispatch tables / entry points
}   // namespace
#else
// ------------------------------------------------------------------------------------------------
// configuration tables
enum Kind { K3S1 = 0, K3S2 = 1, K1 = 2, K7S2 = 3, K2S1 = 4, K4S1 = 7 };   // K4S1: the 7x7 stride-2 stem as a 4x4 conv on the space-to-depth image (bf16 tensors; fp32 tensors on two fp16 planes)

struct Sel {
    int kind, ck, nt, px;
    int th, bn, t, cst;
    int vt;   // virtual tall image tiling
    int split;   // fp32 on the bf16 matrix pipe (conv_split_kernel)
    int small;   // split 3x3 layer too small to fill the chip with 64-co workgroups: 256-pixel x 32-co workgroups instead
    int bf16;    // rcf_conv_desc.precision == RCF_PREC_BF16 and a split kernel: one bf16 plane, one product
    int npl;     // operand planes of a split kernel: 3 (exact fp32), 2 (RCF_PREC_F16X2), 1 (bf16)
    int dma;     // bf16 tensors, channel counts multiples of 16: conv_b16_kernel (operands reach LDS by DMA, rcf_conv_b16_dma.h)
    int pw;      // bf16 tensors, 1x1, <= 64 input and <= 128 output channels: conv1x1_b16_kernel (operands straight from global memory)
    int p4;      // four output phases from one staged tile (DmaCfg / SplitCfg PM): 1 = up-2x forward (phase_sum 2), 2 = stride-2 input gradient (phase_sum 3)
};

int num_cus() {
    static int n = 0;
    if (n == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

#if RCF_CONV_B16
}   // namespace
namespace {
#include "rcf_conv_b16_dma.h"
using D3_2_32 = DmaCfg<3, 2, 32, 2>;
using D3_2_16 = DmaCfg<3, 2, 16, 2>;
using D3_1_32 = DmaCfg<3, 1, 32, 4>;     // 32-co layers: 512-pixel tiles
using D3_1_16 = DmaCfg<3, 1, 16, 4>;
using D3_1_32s = DmaCfg<3, 1, 32, 2>;    // small layers: 256-pixel x 32-co workgroups
using D3_1_16s = DmaCfg<3, 1, 16, 2>;
using D2_2_32 = DmaCfg<2, 2, 32, 2>;
using D2_2_16 = DmaCfg<2, 2, 16, 2>;
using D2_1_32 = DmaCfg<2, 1, 32, 2>;
using D2_1_16 = DmaCfg<2, 1, 16, 2>;
// up-2x forward, four phases from one tile: 256 pixels x 32 co x 4 phases per workgroup, or 128 pixels x 64 co (the four accumulator
// sets are 128 registers either way).  32-pixel tile rows only: the 16-pixel form (per-read LDS addresses) spills
using D2P4_1_32 = DmaCfg<2, 1, 32, 2, 1, 1>;
using D2P4_2_32 = DmaCfg<2, 2, 32, 1, 1, 1>;
using D2S2_1_32 = DmaCfg<2, 1, 32, 2, 1, 2>;   // stride-2 input gradient, four output phases from one dz tile
using D2S2_2_32 = DmaCfg<2, 2, 32, 1, 1, 2>;
using D3S2_2_32 = DmaCfg<3, 2, 32, 1, 2>;   // stride 2: 128-pixel tiles (the 65 x 9 halo tile is 18 KB per buffer)
using D3S2_2_16 = DmaCfg<3, 2, 16, 1, 2>;
using D3S2_1_32 = DmaCfg<3, 1, 32, 1, 2>;
using D3S2_1_16 = DmaCfg<3, 1, 16, 1, 2>;
using D4_1_32 = DmaCfg<4, 1, 32, 2>;        // the stems: 4x4 on the 16-channel space-to-depth image, <= 32 output channels
using D4_1_16 = DmaCfg<4, 1, 16, 2>;
#else
}   // namespace
namespace {
#include "rcf_conv_pw_f16x2.h"
#endif

// Persistent grid: one resident wave of workgroups (occupancy API), split between the n-tiles.  Also the number of
// BN-statistics partial rows the kernel writes, so rcf_conv2d_query reports the same number.
template <class C>
int fwd_grid_x(int ntiles, int ntile_n) {
    static int resident = 0;   // workgroups that fit on the device at once
    if (resident == 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fwd_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  C::LDS_BYTES);
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_fwd_kernel<C>, 256, C::LDS_BYTES) != hipSuccess || per_cu < 1)
            per_cu = 1;
        resident = per_cu * num_cus();
    }
    int gx = resident / ntile_n;
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    return gx;
}

template <class C>
int launch_fwd(const ConvArgs& a, int ntile_n, hipStream_t st) {
    const int gx = fwd_grid_x<C>(a.ntiles, ntile_n);
    dim3 grid(gx, ntile_n, 1);
    hipLaunchKernelGGL((conv_fwd_kernel<C>), grid, dim3(256), C::LDS_BYTES, st, a);
    return rcf_launch_status();
}

template <class C>
int split_grid_x(int ntiles, int ntile_n) {
    static int resident = 0;
    if (resident == 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  C::LDS_BYTES);
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_split_kernel<C>, 256, C::LDS_BYTES) != hipSuccess || per_cu < 1)
            per_cu = 1;
        resident = per_cu * num_cus();
    }
    int gx = resident / ntile_n;
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    return gx;
}

template <class C, bool EPI = false>
int launch_split(const ConvArgs& a, int ntile_n, hipStream_t st) {
    const int gx = split_grid_x<C>(a.ntiles, ntile_n);
    if (EPI) {
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_kernel<C, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      C::LDS_BYTES);
            attr_done = true;
        }
    }
    hipLaunchKernelGGL((conv_split_kernel<C, EPI>), dim3(gx, ntile_n, 1), dim3(256), C::LDS_BYTES, st, a);
    return rcf_launch_status();
}

// the input-gradient launch that also takes the BatchNorm-backward sums of the block whose dY it writes (ConvArgs.bz / bk)
template <class C>
int launch_split_bst(const ConvArgs& a, int ntile_n, hipStream_t st) {
    const int gx = split_grid_x<C>(a.ntiles, ntile_n);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_kernel<C, false, SAct, SAct, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_split_kernel<C, false, SAct, SAct, true>), dim3(gx, ntile_n, 1), dim3(256), C::LDS_BYTES, st, a);
    return rcf_launch_status();
}


template <class C>
int launch_wgrad(const ConvArgs& a, int nsplit, int nchunk, int ncog, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  C::LDS_BYTES);
        attr_done = true;
    }
    dim3 grid(nsplit, nchunk, ncog);
    hipLaunchKernelGGL((conv_wgrad_kernel<C>), grid, dim3(256), C::LDS_BYTES, st, a);
    return rcf_launch_status();
}

template <class C>
int launch_wgrad_split(const ConvArgs& a, int nsplit, int gy, int gz, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_split_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  C::LDS_BYTES);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_wgrad_split_kernel<C>), dim3(nsplit, gy, gz), dim3(256), C::LDS_BYTES, st, a);
    return rcf_launch_status();
}

template <class C>
int launch_wgrad_tr(const ConvArgs& a, int nsplit, int gy, int gz, hipStream_t st) {
    constexpr int lds = C::template lds_bytes<SAct::B16>();
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_tr_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_wgrad_tr_kernel<C>), dim3(nsplit, gy, gz), dim3(512), lds, st, a);
    return rcf_launch_status();
}

#if RCF_CONV_B16
#define RCF_DMA(expr) RCF_EUNSUPPORTED   /* LDS-DMA copies fp32 tiles verbatim: fp32 tensors only */
#else
#define RCF_DMA(expr) (expr)
#endif
template <class C>
int launch_wgrad_dma(const ConvArgs& a, int nsplit, int nchunk, int ncog, hipStream_t st) {
    using H = Halo<C::CST, C::STRP, C::HXP, C::HYP>;
    constexpr int red_floats = C::T * 16 * 64;
    constexpr int tile_floats = H::NA * H::PPI * 32 + C::TP * 32;
    constexpr int lds_bytes = (tile_floats > red_floats ? tile_floats : red_floats) * 4;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_dma_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  lds_bytes);
        attr_done = true;
    }
    dim3 grid(nsplit, nchunk, ncog);
    hipLaunchKernelGGL((conv_wgrad_dma_kernel<C>), grid, dim3(256), lds_bytes, st, a);
    return rcf_launch_status();
}

//                 KSY KSX XE LS  CK CST STRP NT PX MINW
using F3S1_16_1_32 = FwdCfg<3, 3, 0, 1, 16, 16, 20, 1, 32, 2>;
using F3S1_16_2_32 = FwdCfg<3, 3, 0, 1, 16, 16, 20, 2, 32, 2>;
using F3S1_16_1_16 = FwdCfg<3, 3, 0, 1, 16, 16, 20, 1, 16, 2>;
using F3S1_16_2_16 = FwdCfg<3, 3, 0, 1, 16, 16, 20, 2, 16, 2>;
using F3S1_16_1_8 = FwdCfg<3, 3, 0, 1, 16, 16, 20, 1, 8, 2>;
using F3S1_16_2_8 = FwdCfg<3, 3, 0, 1, 16, 16, 20, 2, 8, 2>;
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
using F2_32_1_32 = FwdCfg<2, 2, 0, 1, 32, 32, 36, 1, 32, 2>;
using F2_32_2_32 = FwdCfg<2, 2, 0, 1, 32, 32, 36, 2, 32, 2>;
using F2_32_1_16 = FwdCfg<2, 2, 0, 1, 32, 32, 36, 1, 16, 2>;
using F2_32_2_16 = FwdCfg<2, 2, 0, 1, 32, 32, 36, 2, 16, 2>;
using F7_32 = FwdCfg<7, 1, 7, 2, 32, 4, 4, 1, 32, 2>;
using S3_1_32 = SplitCfg<3, 1, 32>;
using S3_2_32 = SplitCfg<3, 2, 32>;
using S3_1_16 = SplitCfg<3, 1, 16>;
using S3_2_16 = SplitCfg<3, 2, 16>;
using B3_1_32 = SplitCfg<3, 1, 32, 0, 1>;    // bf16-operand variants (one plane, one product)
using B3_2_32 = SplitCfg<3, 2, 32, 0, 1>;
using B3_1_16 = SplitCfg<3, 1, 16, 0, 1>;
using B3_2_16 = SplitCfg<3, 2, 16, 0, 1>;
using B2_1_32 = SplitCfg<2, 1, 32, 0, 1>;
using B2_2_32 = SplitCfg<2, 2, 32, 0, 1>;
using B2_1_16 = SplitCfg<2, 1, 16, 0, 1>;
using B2_2_16 = SplitCfg<2, 2, 16, 0, 1>;
using B3_1_32s = SplitCfg<3, 1, 32, 2, 1>;
using B3_1_16s = SplitCfg<3, 1, 16, 2, 1>;
using S3_1_32s = SplitCfg<3, 1, 32, 2>;   // 256-pixel x 32-co workgroups: more of them for the small layers
using S3_1_16s = SplitCfg<3, 1, 16, 2>;
using S3S2_1_32 = SplitCfg<3, 1, 32, 1, 3, 2>;   // 3x3 stride 2, three planes: 128-pixel tiles (the 65 x 9 halo tile in three planes is 56 KB)
using S3S2_2_32 = SplitCfg<3, 2, 32, 1, 3, 2>;
using S3S2_1_16 = SplitCfg<3, 1, 16, 1, 3, 2>;
using S3S2_2_16 = SplitCfg<3, 2, 16, 1, 3, 2>;
using B3S2_1_32 = SplitCfg<3, 1, 32, 2, 1, 2>;   // bf16 operands: one plane, 256-pixel tiles, two workgroups per CU
using B3S2_2_32 = SplitCfg<3, 2, 32, 2, 1, 2>;
using B3S2_1_16 = SplitCfg<3, 1, 16, 2, 1, 2>;
using B3S2_2_16 = SplitCfg<3, 2, 16, 2, 1, 2>;
using S2_1_32 = SplitCfg<2, 1, 32>;
using S2_2_32 = SplitCfg<2, 2, 32>;
using S2_1_16 = SplitCfg<2, 1, 16>;
using S2_2_16 = SplitCfg<2, 2, 16>;
using F7_16 = FwdCfg<7, 1, 7, 2, 32, 4, 4, 1, 16, 2>;

//               KSY KSX XE LS CST STRP PX TH MINW
using W3S1_32 = WgCfg<3, 3, 0, 1, 32, 32, 32, 8, 1>;
using W3S1_16 = WgCfg<3, 3, 0, 1, 32, 32, 16, 16, 1>;
using W3S1_8 = WgCfg<3, 3, 0, 1, 32, 32, 8, 32, 1>;
using W3S2_32 = WgCfg<3, 3, 0, 2, 32, 32, 32, 4, 1>;
using W3S2_16 = WgCfg<3, 3, 0, 2, 32, 32, 16, 8, 1>;
using W1_32 = WgCfg<1, 1, 0, 1, 32, 32, 32, 8, 2>;
using W1_16 = WgCfg<1, 1, 0, 1, 32, 32, 16, 16, 2>;
using W2_32 = WgCfg<2, 2, 0, 1, 32, 32, 32, 8, 1>;
using W2_16 = WgCfg<2, 2, 0, 1, 32, 32, 16, 16, 1>;
using W7_32 = WgCfg<7, 1, 7, 2, 4, 4, 32, 8, 2>;
using W7_16 = WgCfg<7, 1, 7, 2, 4, 4, 16, 16, 2>;

int ceil_div(int a, int b) { return (a + b - 1) / b; }

template <class T>
struct Tag { using type = T; };

// tile utilisation of a PX x TH tiling of a w x h image
double tile_eff(int w, int h, int px, int th) {
    return ((double)w / (ceil_div(w, px) * px)) * ((double)h / (ceil_div(h, th) * th));
}

// utilisation when the n images are stacked with one separator row each (rows n*(h+1))
double tile_eff_vt(int w, int h, int n, int px, int th) {
    return ((double)w / (ceil_div(w, px) * px)) * ((double)n * h / (ceil_div(n * (h + 1), th) * th));
}

// The split-bf16 kernels (fp32 arithmetic on the bf16 matrix pipe) are the default wherever they exist;
// RCF_CONV_SPLIT=0 selects the exact-f32-MFMA kernels instead (both are parity-tested).
bool split_enabled() {
    const char* e = getenv("RCF_CONV_SPLIT");
    return e == nullptr || e[0] != '0';
}

// 3x3 stride-2 forward convolutions on the bf16 matrix pipe (LSTEP = 2 split kernels).  Default: with bf16 operands only (2.5-3.5x
// the f32-MFMA kernel there).  With fp32 precision the three-plane variant is as accurate as the f32 MFMA against an fp64 reference
// (tools/diag_s2.py: 2.3e-7 vs 3.0e-7 of sum|ab|) and 1.4x faster, but it is 0.7 % of the fp32 step and its different rounding
// pattern moved the (chaotic) parameter-gradient comparison of the tiny 2x113x200 fixture (tools/diag_seeds.py), so the exact f32
// MFMA keeps these four layers in the fp32 configuration.  RCF_S2_SPLIT=0/1 overrides.
bool s2_split_enabled(const rcf_conv_desc* d) {
    const char* e = getenv("RCF_S2_SPLIT");
    if (e != nullptr) return e[0] != '0';
    return d->precision != RCF_PREC_FP32;
}

// The split / DMA kernels address their tensors through buffer descriptors based at the tile's first image, with 32-bit byte offsets
// checked against a 2 GB range; a tile (incl. its halo) touches at most two consecutive images.  An image of 1 GB or more would put
// valid offsets out of that range -- where the hardware silently returns zeros / drops stores -- so such shapes are refused loudly.
// Virtual tall image (vt): a tile of up to 34 rows (32-row tiles + halo) spans ceil(34 / (h + 1)) + 1 images when the image is lower
// than the tile, all of them addressed from the first one's descriptor: the bound per image shrinks accordingly.
bool buffer_range_ok(const rcf_conv_desc* d, bool vt = false) {
    double lim = 1073741824.0;
    const double b = SAct::BYTES;
    if (vt) lim = 2147483648.0 / (double)((34 + d->h_out) / (d->h_out + 1) + 1);
    return (double)d->h_src1 * d->w_src1 * d->c1 * b < lim && (double)d->h_in * d->w_in * d->c2 * b < lim &&
           (double)d->out_h_phys * d->out_w_phys * d->c_out * b < lim;
}

// stride-1 3x3 conv with pad 1 on directly addressed sources: the separator row is the conv's own zero padding
bool vt_allowed(const rcf_conv_desc* d) {
    return d->ksize == 3 && d->stride == 1 && d->pad == 1 && d->pad_x == 1 && d->gather1 == RCF_GATHER_DIRECT &&
           d->out_stride == 1 && d->h_in == d->h_out && d->n > 1 && getenv("RCF_NO_VT") == nullptr;
}

bool valid_desc(const rcf_conv_desc* d) {
    if (!d) return false;
    if (d->n <= 0 || d->h_in <= 0 || d->w_in <= 0 || d->c1 <= 0 || d->c2 < 0 || d->h_out <= 0 || d->w_out <= 0 ||
        d->c_out <= 0)
        return false;
    if (d->ksize != 1 && d->ksize != 2 && d->ksize != 3 && d->ksize != 4 && d->ksize != 7) return false;
    if (d->stride != 1 && d->stride != 2) return false;
    if (d->gather1 < 0 || d->gather1 > 3) return false;
    if (d->precision != RCF_PREC_FP32 && d->precision != RCF_PREC_BF16 && d->precision != RCF_PREC_F16X2) return false;
    if (d->storage != RCF_STORE_FP32 && d->storage != RCF_STORE_BF16) return false;
    if (d->storage == RCF_STORE_BF16 && d->precision != RCF_PREC_BF16) return false;   // bf16 tensors are consumed as bf16 operands
    if ((d->storage == RCF_STORE_BF16) != SAct::B16) return false;                    // (each translation unit serves one storage)
    if (d->out_stride != 1 && d->out_stride != 2) return false;
    if (d->out_h_phys <= 0 || d->out_w_phys <= 0) return false;
    if (d->gather1 == RCF_GATHER_DIRECT && (d->h_src1 != d->h_in || d->w_src1 != d->w_in)) return false;
    if (d->h_src1 <= 0 || d->w_src1 <= 0) return false;
    if (d->ksize == 4) {   // the stem on the space-to-depth image: output grid given (pad 2 on top / left, what is left at the bottom / right)
        if (d->stride != 1 || d->w_mode != RCF_W_FORWARD || d->c1 != 16 || d->c2 != 0 || d->gather1 != RCF_GATHER_DIRECT || d->pad != 2 ||
            d->pad_x != 2 || d->out_stride != 1 || d->accumulate)
            return false;
    } else
    if (d->ksize != 2) {   // 2x2 phase convs pad asymmetrically: their output grid is given, not derived
        if ((d->h_in + 2 * d->pad - d->ksize) / d->stride + 1 != d->h_out) return false;
        if ((d->w_in + 2 * d->pad_x - d->ksize) / d->stride + 1 != d->w_out) return false;
    } else if (d->stride != 1 || d->w_mode != RCF_W_FORWARD) {
        return false;
    }
    if (d->phase_sum < 0 || d->phase_sum > 3) return false;
    if (d->phase_sum == 3 && (d->ksize != 2 || d->gather1 != RCF_GATHER_DIRECT || d->c2 != 0 || d->out_stride != 2 || d->pad != 0 || d->pad_x != 0 ||
                              d->w_mode != RCF_W_FORWARD || d->h_out != (d->out_h_phys + 1) / 2 || d->w_out != (d->out_w_phys + 1) / 2))
        return false;
    if (d->phase_sum == 1 && (d->ksize != 2 || d->gather1 != RCF_GATHER_STRIDED2 || d->c2 != 0)) return false;
    if (d->phase_sum == 2 && (d->ksize != 2 || d->gather1 != RCF_GATHER_DIRECT || d->c2 != 0 || d->out_stride != 2 || d->accumulate ||
                              d->w_mode != RCF_W_FORWARD || d->h_out != d->h_in || d->w_out != d->w_in || d->out_h_phys != 2 * d->h_out ||
                              d->out_w_phys != 2 * d->w_out))
        return false;
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
    } else if (d->ksize == 4) {
        // bf16 tensors (conv_b16_kernel) or fp32 tensors on the two-plane split kernel (RCF_PREC_F16X2), one 32-co tile
        if ((!SAct::B16 && d->precision != RCF_PREC_F16X2) || d->c_out > 32 || !split_enabled()) return RCF_EUNSUPPORTED;
        s->kind = K4S1; s->t = 16; s->ck = 16; s->cst = 16; s->nt = 1;
    } else if (d->ksize == 2) {
        s->kind = K2S1; s->t = 4; s->ck = 32; s->cst = 32;
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
    // tile shape: 32x8 or 16x16 output pixels (8x32 too for the stride-1 3x3 kernels with >= 16 channels), whichever
    // wastes least at the image edges; stride-1 3x3 convs may also tile the batch as one tall virtual image
    s->vt = 0;
    s->split = 0;
    s->small = 0;
    s->bf16 = 0;
    s->npl = 3;
    s->dma = 0;
    s->pw = 0;
    s->p4 = 0;
#if !RCF_CONV_B16
    {   // fp32 tensors, two fp16 planes: the streaming 1x1 kernel (conv1x1_f16x2_kernel; RCF_F32_PW=0 keeps the f32-MFMA kernel)
        const char* e = getenv("RCF_F32_PW");
        if (s->kind == K1 && d->precision == RCF_PREC_F16X2 && d->c2 == 0 && d->c1 % 16 == 0 && d->c1 <= 64 && d->c_out <= 128 &&
            pw2_cfg_ok(d->c1 / 16, ceil_div(d->c_out, 32)) && d->gather1 == RCF_GATHER_DIRECT && d->out_stride == 1 &&
            d->out_off_y == 0 && d->out_off_x == 0 && d->out_h_phys == d->h_out && d->out_w_phys == d->w_out &&
            (d->w_mode == RCF_W_FORWARD || d->stride == 1) && split_enabled() && (e == nullptr || e[0] != '0')) {
            s->pw = 1; s->split = 1; s->bf16 = 0; s->npl = 2; s->ck = 16; s->cst = 16;
            s->nt = ceil_div(d->c_out, 32);
            s->px = 32; s->th = 8; s->bn = 32 * s->nt;
            return RCF_OK;
        }
    }
#endif
#if RCF_CONV_B16
    {
        const char* e = getenv("RCF_B16_PW");
        const bool small_k = d->c1 <= 64 && d->c_out <= 128;
        const bool big_k = (d->c1 == 128 || d->c1 == 256) && d->c_out <= 256;   // weights of one 64 / 32-channel n-tile in registers
        if (s->kind == K1 && d->precision == RCF_PREC_BF16 && d->c2 == 0 && d->c1 % 16 == 0 && (small_k || big_k) &&
            (d->gather1 == RCF_GATHER_DIRECT || (d->gather1 == RCF_GATHER_ZERO_INSERT && d->stride == 1)) && d->out_stride == 1 &&
            (d->w_mode == RCF_W_FORWARD || d->stride == 1) && (e == nullptr || e[0] != '0')) {
            s->pw = 1; s->split = 1; s->bf16 = 1; s->npl = 1; s->ck = 16; s->cst = 16;
            s->nt = pw_nt(d->c1 / 16, ceil_div(d->c_out, 32));
            s->px = 32; s->th = 8; s->bn = 32 * s->nt;
            return RCF_OK;
        }
    }
#endif
    const bool s2_split = s->kind == K3S2 && cmax >= 16 && d->w_mode == RCF_W_FORWARD && d->gather1 == RCF_GATHER_DIRECT && d->c2 == 0 &&
                          d->out_stride == 1 && s2_split_enabled(d);
    if (((s->kind == K3S1 && s->ck == 16) || (s->kind == K2S1 && cmax >= 16) || s2_split) && (d->c1 % 4 == 0) && (d->c2 % 4 == 0) &&
        split_enabled()) {
        s->split = 1;
        s->ck = 16;
        s->cst = 16;
        // fewer 256-pixel x 64-co workgroups than CUs: halve the workgroup (BN = 32) to double their number
        const long long wgs = (((long long)d->n * d->h_out * d->w_out + 255) / 256) * ceil_div(d->c_out, 64);
        if (s->kind == K3S1 && s->nt == 2 && wgs < num_cus()) { s->small = 1; s->nt = 1; }
        s->bf16 = d->precision == RCF_PREC_BF16 ? 1 : 0;
        s->npl = s->bf16 ? 1 : (d->precision == RCF_PREC_F16X2 ? 2 : 3);
#if RCF_CONV_B16
        const char* e = getenv("RCF_B16_DMA");
        s->dma = (s->bf16 && d->c1 % 16 == 0 && d->c2 % 16 == 0 && (e == nullptr || e[0] != '0')) ? 1 : 0;
#endif
        // the four output phases of an up-2x forward from one staged tile: conv_b16_kernel (bf16 tensors) and the two-plane
        // conv_split_kernel (fp32 tensors).  RCF_UP2X_MERGED=0: the four phases back to back, each staging its own tile (round 4's
        // one-launch form), for an A/B
        static const int merged = [] { const char* m = getenv("RCF_UP2X_MERGED"); return m ? atoi(m) : 1; }();
        s->p4 = (d->phase_sum >= 2 && merged && (s->dma || (!s->bf16 && s->npl == 2))) ? d->phase_sum - 1 : 0;   // 1: up-2x forward, 2: stride-2 input gradient
    }
    if (s->kind == K4S1) {
        s->split = 1; s->ck = 16; s->cst = 16;
        if (SAct::B16) { s->bf16 = 1; s->npl = 1; s->dma = 1; }
        else { s->bf16 = 0; s->npl = 2; s->dma = 0; }
    }
    double best = -1.0;
    const bool vt_ok = vt_allowed(d);
    const int pxs[3] = {32, 16, 8};
    // pixels per workgroup tile: 256; 512 for the 32-co 3x3 split layers; 128 for the three-plane stride-2 split kernel
    const int tile_px = ((s->split && s->kind == K3S2 && (!s->bf16 || s->dma)) || (s->p4 && s->nt == 2)) ? 128 :
                        ((s->split && s->nt == 1 && s->kind == K3S1 && !s->small) ? 512 : 256);
    for (int i = 0; i < 3; ++i) {
        const int px = pxs[i], th = tile_px / px;
        if (px == 8 && (s->split || !(s->kind == K3S1 && s->ck == 16))) continue;
        if (s->p4 && px != 32) continue;
        for (int vt = 0; vt <= (vt_ok ? 1 : 0); ++vt) {
            const double e = vt ? tile_eff_vt(d->w_out, d->h_out, d->n, px, th) : tile_eff(d->w_out, d->h_out, px, th);
            if (e > best + 1e-9) { best = e; s->px = px; s->vt = vt; }
        }
    }
    s->th = tile_px / s->px;
    s->bn = 32 * s->nt;
    if (d->phase_sum == 2 && !s->split) return RCF_EUNSUPPORTED;   // the four output phases in one launch: conv_split_kernel / conv_b16_kernel only
    if (d->phase_sum == 3 && s->p4 != 2) return RCF_EUNSUPPORTED;  // the stride-2 input gradient's phases exist only as the merged kernels
    if (s->split && !s->pw && !buffer_range_ok(d, s->vt != 0)) return RCF_EUNSUPPORTED;
    return RCF_OK;
}

// device address of rcf_zero_page (this translation unit's copy), looked up once
const float* zero_page_ptr() {
    static const float* p = nullptr;
    if (p == nullptr) {
        void* q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(rcf_zero_page)) == hipSuccess) p = static_cast<const float*>(q);
    }
    return p;
}

void fill_args(const rcf_conv_desc* d, const Sel& s, ConvArgs* a) {
    a->coef1 = nullptr; a->coef2 = nullptr; a->bias = nullptr; a->res = nullptr;
    a->amax_a1 = nullptr; a->amax_a2 = nullptr; a->amax_b = nullptr;
    a->bz = nullptr; a->bk = nullptr;
    a->zero = zero_page_ptr();
    a->n = d->n; a->h_in = d->h_in; a->w_in = d->w_in; a->c1 = d->c1; a->c2 = d->c2;
    a->h1 = d->h_src1; a->w1 = d->w_src1; a->gather1 = d->gather1;
    a->h_out = d->h_out; a->w_out = d->w_out; a->c_out = d->c_out; a->pad = d->pad; a->pad_x = d->pad_x; a->stride = d->stride;
    a->gstep = d->ksize == 1 ? d->stride : 1;
    a->accumulate = d->accumulate;
    a->os = d->out_stride; a->ooy = d->out_off_y; a->oox = d->out_off_x; a->ohp = d->out_h_phys; a->owp = d->out_w_phys;
    a->ioy = d->in_off_y; a->iox = d->in_off_x;
    a->phase_sum = d->phase_sum; a->wp_phase_stride = 0;
    a->sy = (float)d->h_src1 / (float)d->h_in;
    a->sx = (float)d->w_src1 / (float)d->w_in;
    a->tiles_x = ceil_div(d->w_out, s.px);
    a->vt = s.vt; a->hp = d->h_out + 1; a->nimg = d->n; a->inv_hp = 1.0f / (float)(d->h_out + 1);
    if (s.vt) {
        a->tiles_y = ceil_div(d->n * (d->h_out + 1), s.th);
        a->ntiles = a->tiles_x * a->tiles_y;
    } else {
        a->tiles_y = ceil_div(d->h_out, s.th);
        a->ntiles = d->n * a->tiles_x * a->tiles_y;
    }
    a->nchunk1 = ceil_div(d->c1, s.cst);
    a->nchunk2 = d->c2 > 0 ? ceil_div(d->c2, s.cst) : 0;
    // on by default: same step time within 0.3 %, 8 % (two-plane 3x3) to 15 % (bf16 3x3) fewer bytes fetched past the L2 (PMC, round 4:
    // profiles/r04_xcd_bands.txt); RCF_XCD_BANDS=0 restores the round-robin deal for an A/B
    static const int xcd_band = [] { const char* e = getenv("RCF_XCD_BANDS"); return e ? atoi(e) : 1; }();
    a->xcd_band = xcd_band;
}

template <class F>
int dispatch_fwd(const Sel& s, F&& f) {
    const bool p16 = s.px == 16;
    switch (s.kind) {
        case K3S1:
            if (s.ck == 16) {
                if (s.px == 8) return s.nt == 1 ? f(Tag<F3S1_16_1_8>{}) : f(Tag<F3S1_16_2_8>{});
                if (s.nt == 1) return p16 ? f(Tag<F3S1_16_1_16>{}) : f(Tag<F3S1_16_1_32>{});
                return p16 ? f(Tag<F3S1_16_2_16>{}) : f(Tag<F3S1_16_2_32>{});
            }
            if (s.nt == 1) return p16 ? f(Tag<F3S1_8_1_16>{}) : f(Tag<F3S1_8_1_32>{});
            return p16 ? f(Tag<F3S1_8_2_16>{}) : f(Tag<F3S1_8_2_32>{});
        case K3S2:
            if (s.nt == 1) return p16 ? f(Tag<F3S2_8_1_16>{}) : f(Tag<F3S2_8_1_32>{});
            return p16 ? f(Tag<F3S2_8_2_16>{}) : f(Tag<F3S2_8_2_32>{});
        case K1:
            if (s.ck == 32) {
                if (s.nt == 1) return p16 ? f(Tag<F1_32_1_16>{}) : f(Tag<F1_32_1_32>{});
                return p16 ? f(Tag<F1_32_2_16>{}) : f(Tag<F1_32_2_32>{});
            }
            if (s.nt == 1) return p16 ? f(Tag<F1_16_1_16>{}) : f(Tag<F1_16_1_32>{});
            return p16 ? f(Tag<F1_16_2_16>{}) : f(Tag<F1_16_2_32>{});
        case K7S2:
            return p16 ? f(Tag<F7_16>{}) : f(Tag<F7_32>{});
        case K2S1:
            if (s.nt == 1) return p16 ? f(Tag<F2_32_1_16>{}) : f(Tag<F2_32_1_32>{});
            return p16 ? f(Tag<F2_32_2_16>{}) : f(Tag<F2_32_2_32>{});
    }
    return RCF_EUNSUPPORTED;
}

#if RCF_CONV_B16
template <class F>
int dispatch_dma(const Sel& s, F&& f) {
    const bool p16 = s.px == 16;
    if (s.kind == K2S1 && s.p4 == 2) return s.nt == 1 ? f(Tag<D2S2_1_32>{}) : f(Tag<D2S2_2_32>{});
    if (s.kind == K2S1 && s.p4) return s.nt == 1 ? f(Tag<D2P4_1_32>{}) : f(Tag<D2P4_2_32>{});
    if (s.kind == K2S1) {
        if (s.nt == 1) return p16 ? f(Tag<D2_1_16>{}) : f(Tag<D2_1_32>{});
        return p16 ? f(Tag<D2_2_16>{}) : f(Tag<D2_2_32>{});
    }
    if (s.kind == K4S1) return p16 ? f(Tag<D4_1_16>{}) : f(Tag<D4_1_32>{});
    if (s.kind == K3S2) {
        if (s.nt == 1) return p16 ? f(Tag<D3S2_1_16>{}) : f(Tag<D3S2_1_32>{});
        return p16 ? f(Tag<D3S2_2_16>{}) : f(Tag<D3S2_2_32>{});
    }
    if (s.nt == 1 && s.small) return p16 ? f(Tag<D3_1_16s>{}) : f(Tag<D3_1_32s>{});
    if (s.nt == 1) return p16 ? f(Tag<D3_1_16>{}) : f(Tag<D3_1_32>{});
    return p16 ? f(Tag<D3_2_16>{}) : f(Tag<D3_2_32>{});
}
#endif

// fp32 tensors: NPL = 3 (the S* configurations above: exact) or NPL = 2 (RCF_PREC_F16X2), same tile shapes
template <int NPL, class F>
int dispatch_split_planes(const Sel& s, F&& f) {
    const bool p16 = s.px == 16;
    if (s.kind == K4S1) {   // the stems on the fp32 space-to-depth image: 4x4 taps, 16 channels, <= 32 output channels, two planes only
        if constexpr (NPL == 2) return p16 ? f(Tag<SplitCfg<4, 1, 16, 0, 2>>{}) : f(Tag<SplitCfg<4, 1, 32, 0, 2>>{});
        else return RCF_EUNSUPPORTED;
    }
    if (s.kind == K3S2) {
        if (s.nt == 1) return p16 ? f(Tag<SplitCfg<3, 1, 16, 1, NPL, 2>>{}) : f(Tag<SplitCfg<3, 1, 32, 1, NPL, 2>>{});
        return p16 ? f(Tag<SplitCfg<3, 2, 16, 1, NPL, 2>>{}) : f(Tag<SplitCfg<3, 2, 32, 1, NPL, 2>>{});
    }
    if (s.kind == K2S1 && s.p4) {   // (select_cfg: two planes, 32-pixel tile rows; 64 co: 128-pixel tiles -- four accumulator sets)
        if constexpr (NPL == 2) {
            if (s.p4 == 2) return s.nt == 1 ? f(Tag<SplitCfg<2, 1, 32, 0, 2, 1, 2>>{}) : f(Tag<SplitCfg<2, 2, 32, 1, 2, 1, 2>>{});
            return s.nt == 1 ? f(Tag<SplitCfg<2, 1, 32, 0, 2, 1, 1>>{}) : f(Tag<SplitCfg<2, 2, 32, 1, 2, 1, 1>>{});
        } else return RCF_EUNSUPPORTED;
    }
    if (s.kind == K2S1) {
        if (s.nt == 1) return p16 ? f(Tag<SplitCfg<2, 1, 16, 0, NPL>>{}) : f(Tag<SplitCfg<2, 1, 32, 0, NPL>>{});
        return p16 ? f(Tag<SplitCfg<2, 2, 16, 0, NPL>>{}) : f(Tag<SplitCfg<2, 2, 32, 0, NPL>>{});
    }
    if (s.nt == 1 && s.small) return p16 ? f(Tag<SplitCfg<3, 1, 16, 2, NPL>>{}) : f(Tag<SplitCfg<3, 1, 32, 2, NPL>>{});
    if (s.nt == 1) return p16 ? f(Tag<SplitCfg<3, 1, 16, 0, NPL>>{}) : f(Tag<SplitCfg<3, 1, 32, 0, NPL>>{});
    return p16 ? f(Tag<SplitCfg<3, 2, 16, 0, NPL>>{}) : f(Tag<SplitCfg<3, 2, 32, 0, NPL>>{});
}

// configurations that exist with the BatchNorm-backward sums in the epilogue (rcf_conv_info.bn_bwd_sums): fp32 tensors, two fp16
// planes, 3x3 stride 1 and 2x2 (the four-phase input gradient of an up-2x convolution in one launch)
bool split_bst_ok(const Sel& s) {
#if RCF_CONV_B16
    // conv_b16_kernel<C, false, true>: every 3x3 stride-1 and 2x2 configuration (round 5: the buffer-addressed epilogue freed ~40
    // registers; the 64-co x 32-pixel-row one had been excluded for its spills)
    return s.dma && !s.pw && (s.kind == K2S1 || s.kind == K3S1);
#else
    return s.split && !s.bf16 && s.npl == 2 && (s.kind == K3S1 || s.kind == K2S1);
#endif
}
template <class F>
int dispatch_split_bst(const Sel& s, F&& f) {
    if constexpr (!SAct::B16) {
        if (!split_bst_ok(s)) return RCF_EUNSUPPORTED;
        const bool p16 = s.px == 16;
        if (s.kind == K2S1) {
            if (s.nt == 1) return p16 ? f(Tag<SplitCfg<2, 1, 16, 0, 2>>{}) : f(Tag<SplitCfg<2, 1, 32, 0, 2>>{});
            return p16 ? f(Tag<SplitCfg<2, 2, 16, 0, 2>>{}) : f(Tag<SplitCfg<2, 2, 32, 0, 2>>{});
        }
        if (s.nt == 1 && s.small) return p16 ? f(Tag<SplitCfg<3, 1, 16, 2, 2>>{}) : f(Tag<SplitCfg<3, 1, 32, 2, 2>>{});
        if (s.nt == 1) return p16 ? f(Tag<SplitCfg<3, 1, 16, 0, 2>>{}) : f(Tag<SplitCfg<3, 1, 32, 0, 2>>{});
        return p16 ? f(Tag<SplitCfg<3, 2, 16, 0, 2>>{}) : f(Tag<SplitCfg<3, 2, 32, 0, 2>>{});
    }
    return RCF_EUNSUPPORTED;
}


template <class F>
int dispatch_split(const Sel& s, F&& f) {
    if (s.bf16) {
        if (s.kind == K2S1) {
            if (s.nt == 1) return s.px == 16 ? f(Tag<B2_1_16>{}) : f(Tag<B2_1_32>{});
            return s.px == 16 ? f(Tag<B2_2_16>{}) : f(Tag<B2_2_32>{});
        }
        if (s.kind == K3S2) {
            if (s.nt == 1) return s.px == 16 ? f(Tag<B3S2_1_16>{}) : f(Tag<B3S2_1_32>{});
            return s.px == 16 ? f(Tag<B3S2_2_16>{}) : f(Tag<B3S2_2_32>{});
        }
        if (s.nt == 1 && s.small) return s.px == 16 ? f(Tag<B3_1_16s>{}) : f(Tag<B3_1_32s>{});
        if (s.nt == 1) return s.px == 16 ? f(Tag<B3_1_16>{}) : f(Tag<B3_1_32>{});
        return s.px == 16 ? f(Tag<B3_2_16>{}) : f(Tag<B3_2_32>{});
    }
    if constexpr (!SAct::B16) {   // the multi-plane splits exist for fp32 tensors only
        return s.npl == 2 ? dispatch_split_planes<2>(s, f) : dispatch_split_planes<3>(s, f);
    }
    return RCF_EUNSUPPORTED;
}

// The split weight-gradient kernel addresses a virtual-tall tile through its general path (~3x the address arithmetic per load of
// the plain path, and that arithmetic is 20-60 % of its wave time: tools/phase_timing_wgrad.py), so the virtual-tall tiling must
// win more than a few per cent of tile utilisation to pay: its utilisation is discounted by this factor (RCF_WGRAD_VT_BIAS to tune).
double wgrad_vt_bias(int split) {
    static double bias = -1.0;
    if (bias < 0.0) {
        const char* e = getenv("RCF_WGRAD_VT_BIAS");
        bias = e ? atof(e) : 0.8;
    }
    return split ? bias : 1.0;
}

// wgrad tiling for the forward descriptor
struct WSel { int kind, px, th, t, cst, nchunk1, nchunk2, ncog, nsplit, ktot, cop, tiles_x, tiles_y, ntiles, vt, split, wci, wco, gy, gz; };

// tile slots per phase of a four-phase weight-gradient launch (phase_sum 1 / 2) given the slots of a single-phase one: a quarter,
// in whole rounds of the eight XCDs where there are that many (the four phases of a slot then share an XCD and its L2), at least one
// (the up-2x weight gradient on one or two operand planes runs phase PAIRS: a workgroup owns (a, 0) and (a, 1) -- half the slots, two
// row blocks each; the three-plane tier has no LDS for two dz tiles and keeps (slot, phase) workgroups)
bool wgrad_phase_pairs(const rcf_conv_desc* d) { return d->phase_sum == 2 && d->precision != RCF_PREC_FP32; }
int wgrad_phase_slots(int nsplit, const rcf_conv_desc* d) {
    int n = nsplit / (wgrad_phase_pairs(d) ? 2 : 4);
    if (n >= 8) n &= ~7;
    return n < 1 ? 1 : n;
}

int select_wgrad(const rcf_conv_desc* d, WSel* w) {
    if (!valid_desc(d) || d->w_mode != RCF_W_FORWARD) return RCF_EINVAL;
    if (d->phase_sum == 3) return RCF_EUNSUPPORTED;   // (an input-gradient descriptor: it has no weight gradient of its own)
    if (d->c_out % 4 != 0) return RCF_EUNSUPPORTED;
    if (d->ksize == 4) return RCF_EUNSUPPORTED;   // the stem's weight gradient is taken on the 7x7 form (fp32 NHWC input)
    if (d->ksize == 7) {
        if (d->stride != 2 || d->c2 != 0 || d->c1 > 4 || d->gather1 != RCF_GATHER_DIRECT) return RCF_EUNSUPPORTED;
        w->kind = K7S2; w->t = 7; w->cst = 4;
    } else if (d->ksize == 2) {
        w->kind = K2S1; w->t = 4; w->cst = 32;
    } else if (d->ksize == 3) {
        w->kind = d->stride == 2 ? K3S2 : K3S1; w->t = 9; w->cst = 32;
    } else {
        w->kind = K1; w->t = 1; w->cst = 32;
    }
    if ((d->c1 % 4 != 0 || d->c2 % 4 != 0) && w->kind != K7S2 && d->c2 != 0) return RCF_EUNSUPPORTED;
    const int th32 = (w->kind == K3S2) ? 4 : 8;
    const int th16 = (w->kind == K3S2) ? 8 : 16;
    const bool dma_ok = w->kind != K7S2 && (d->c1 % 4 == 0) && (d->c2 % 4 == 0);
    w->vt = 0;
    // 3x3 stride-1 layers run on the bf16 matrix pipe (conv_wgrad_split_kernel): 16x8 tiles, 32/64-channel blocks per workgroup
    w->split = (dma_ok && split_enabled() &&
                ((w->kind == K3S1 && d->out_stride == 1 && d->out_h_phys == d->h_out && d->out_w_phys == d->w_out &&
                  (d->gather1 == RCF_GATHER_DIRECT || d->gather1 == RCF_GATHER_NEAREST)) ||
                 (w->kind == K2S1 && (d->gather1 == RCF_GATHER_DIRECT || d->gather1 == RCF_GATHER_STRIDED2)) ||
                 // 1x1 (the fusion convs, the stride-1 projections) with bf16 tensors: the same kernel with one tap -- an HBM-bound
                 // layer that the register-staged f32-MFMA kernel ran at 1.3 TB/s [r4].  fp32 tensors keep the DMA-staged f32 kernel
                 // (a stride-2 projection reads x at the even positions: the kernel's STRIDED2 addressing with zero offsets)
                 (w->kind == K1 && SAct::B16 && d->gather1 == RCF_GATHER_DIRECT && d->out_stride == 1 &&
                  d->out_h_phys == d->h_out && d->out_w_phys == d->w_out))) ? 1 : 0;
    // virtual-tall tiling: the DMA / split kernels address the separator rows, the register-staged kernel (the only f32-MFMA
    // weight-gradient kernel for bf16 tensors) does not
    const bool vt_ok = dma_ok && vt_allowed(d) && (!SAct::B16 || w->split) && !(w->kind == K1 && d->stride == 2);
    w->wci = w->wco = 1; w->gy = w->gz = 0;
    if (w->split) {   // channels per workgroup: 64 x 64, 32 x 64, 64 x 32 or 32 x 32 (the last with 16-row tiles)
        w->wco = d->c_out > 32 ? 2 : 1;
        w->wci = (d->c1 % 64 == 0 && d->c2 % 64 == 0) ? 2 : 1;
    }
    // rows per tile: bf16 tensors stage raw (half the registers) and their MFMA phase is 6x shorter: twice the rows per barrier pair
    const int th_split = ((w->wci == 1 && w->wco == 1) || SAct::B16) ? 16 : 8;
    double best = -1.0;
    const int pxs[3] = {32, 16, 8}, ths[3] = {th32, w->split ? th_split : th16, 32};
    for (int i = 0; i < 3; ++i) {
        if (w->split && pxs[i] != 16) continue;
        if (pxs[i] == 8 && (SAct::B16 || !(w->kind == K3S1 && dma_ok))) continue;
        for (int vt = 0; vt <= (vt_ok ? 1 : 0); ++vt) {
            const double e = vt ? tile_eff_vt(d->w_out, d->h_out, d->n, pxs[i], ths[i]) : tile_eff(d->w_out, d->h_out, pxs[i], ths[i]);
            if (e * (vt ? wgrad_vt_bias(w->split) : 1.0) > best + 1e-9) { best = e * (vt ? wgrad_vt_bias(w->split) : 1.0); w->px = pxs[i]; w->th = ths[i]; w->vt = vt; }
        }
    }
    w->tiles_x = ceil_div(d->w_out, w->px);
    if (w->vt) {
        w->tiles_y = ceil_div(d->n * (d->h_out + 1), w->th);
        w->ntiles = w->tiles_x * w->tiles_y;
    } else {
        w->tiles_y = ceil_div(d->h_out, w->th);
        w->ntiles = d->n * w->tiles_x * w->tiles_y;
    }
    w->nchunk1 = (w->kind == K7S2) ? 1 : ceil_div(d->c1, 32);
    w->nchunk2 = d->c2 > 0 ? ceil_div(d->c2, 32) : 0;
    w->ncog = ceil_div(d->c_out, 32);
    const int combos = (w->nchunk1 + w->nchunk2) * w->ncog;
    int ns = 512 / combos;   // <= one resident wave at 2 workgroups/CU (two at 1/CU); never a nearly empty extra wave
    if (ns > w->ntiles) ns = w->ntiles;
    if (ns < 1) ns = 1;
    w->nsplit = ns;
    w->ktot = (w->nchunk1 + w->nchunk2) * w->t * 32;
    w->cop = w->ncog * 32;
    if (w->split) {
        w->gy = ceil_div(d->c1, 32 * w->wci) + (d->c2 > 0 ? ceil_div(d->c2, 32 * w->wci) : 0);
        w->gz = ceil_div(d->c_out, 32 * w->wco);
        int nsp = num_cus() / (w->gy * w->gz);   // one workgroup per CU
        if (nsp > w->ntiles) nsp = w->ntiles;
        if (nsp < 1) nsp = 1;
        w->nsplit = nsp;
        if (!buffer_range_ok(d, w->vt != 0)) return RCF_EUNSUPPORTED;
    }
    return RCF_OK;
}

// conv_wgrad_tr_kernel (rcf_conv_wgrad_tr.h: producer / consumer waves, transposing LDS reads) serves the split weight gradients
// of plain 3x3 / 2x2 (/ 1x1 with bf16 tensors) layers on one bf16 plane (bf16 tensors) or two fp16 planes (fp32 tensors); the
// merged up-2x phase pairs, BatchNorm-on-load and the three-plane tier keep conv_wgrad_split_kernel.
// RCF_WGRAD_TR=0 switches it off (same-box A/B, bitwise tests).
bool wgrad_tr_ok(const rcf_conv_desc* d, const WSel& w) {
    const char* e = getenv("RCF_WGRAD_TR");   // read per call: tests toggle it inside one process
    if ((e != nullptr && e[0] == '0') || !w.split) return false;
    // phase_sum 1: the four phases of a stride-2 weight gradient as (slot, phase) workgroups; 2 (the up-2x phase pairs from one x
    // tile, eight accumulators and two dz tiles) stays on conv_wgrad_split_kernel
    if (d->phase_sum != 0 && !(d->phase_sum == 1 && w.kind == K2S1 && d->gather1 == RCF_GATHER_STRIDED2)) return false;
    if (SAct::B16) {
        if (d->precision != RCF_PREC_BF16 || d->c1 % 8 != 0 || d->c2 % 8 != 0 || d->c_out % 8 != 0) return false;
        return w.kind == K3S1 || w.kind == K2S1 || (w.kind == K1 && d->stride == 1);
    }
    return d->precision == RCF_PREC_F16X2 && (w.kind == K3S1 || w.kind == K2S1);
}

}   // namespace

// ---- entry points.  The fp32 unit owns the public names and hands descriptors with storage == RCF_STORE_BF16 to the bf16 unit.
#if RCF_CONV_B16
#define RCF_FN(name) name##_b16impl
#define RCF_TO_B16(d, call)
#else
#define RCF_FN(name) name
#define RCF_TO_B16(d, call) \
    if ((d) != nullptr && (d)->storage == RCF_STORE_BF16) return call
extern "C" {
int rcf_conv2d_query_b16impl(const rcf_conv_desc* d, rcf_conv_info* info);
int rcf_conv2d_dgrad_bn_sums_b16impl(const rcf_conv_desc* d, const void* dz, const float* packed, void* dx, const void* bn_z,
                                     const float* bn_coef, double* sum_partials, const rcf_conv_scales* scales, void* stream);
int rcf_conv2d_pack_weights_b16impl(const rcf_conv_desc* d, const float* w_oihw, float* packed, void* stream);
int rcf_conv2d_fwd_b16impl(const rcf_conv_desc* d, const void* in1, const void* in2, const float* packed, void* out, double* stat_partials,
                           void* stream);
int rcf_conv2d_fwd_bn_b16impl(const rcf_conv_desc* d, const void* in1, const float* coef1, const void* in2, const float* coef2,
                              const float* packed, void* out, double* stat_partials, void* stream);
int rcf_conv2d_fwd_act_b16impl(const rcf_conv_desc* d, const void* in1, const void* in2, const float* packed, const float* bias,
                               const void* res, void* out, void* stream);
int rcf_conv2d_wgrad_b16impl(const rcf_conv_desc* d, const void* in1, const void* in2, const void* dz, float* dw_oihw, float* workspace,
                             void* stream);
int rcf_conv2d_pack_weights_batch_b16impl(const rcf_pack_item* items, int n, void* stream);
int rcf_conv2d_wgrad_bn_b16impl(const rcf_conv_desc* d, const void* in1, const float* coef1, const void* in2, const float* coef2,
                                const void* dz, float* dw_oihw, float* workspace, void* stream);
}
#endif

#if !RCF_CONV_B16
extern "C" int rcf_phase_weights(const float* w_oihw, float* out, int o, int i, int mode, void* stream) {
    if (!w_oihw || !out || o <= 0 || i <= 0 || mode < 0 || mode > 2) return RCF_EINVAL;
    const int total = 16 * o * i;
    hipLaunchKernelGGL(phase_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, out, o, i, mode);
    return rcf_launch_status();
}

extern "C" int rcf_phase_weights_batch(const rcf_phase_item* items, int n, void* stream) {
    if (!items || n <= 0) return RCF_EINVAL;
    static thread_local PhaseBatch b;
    int i = 0;
    while (i < n) {
        b.n = 0;
        unsigned nblk = 0;
        while (i < n && b.n < PHASE_BATCH) {
            const rcf_phase_item& it = items[i];
            if (!it.w_oihw || !it.out || it.o <= 0 || it.i <= 0 || it.mode < 0 || it.mode > 2) return RCF_EINVAL;
            b.it[b.n].w = it.w_oihw; b.it[b.n].out = it.out; b.it[b.n].O = it.o; b.it[b.n].I = it.i; b.it[b.n].mode = it.mode; b.it[b.n].pad = 0;
            b.blk_start[b.n] = nblk;
            nblk += (unsigned)((16 * it.o * it.i + 255) / 256);
            ++b.n;
            ++i;
        }
        b.blk_start[b.n] = nblk;
        for (int j = b.n + 1; j <= PHASE_BATCH; ++j) b.blk_start[j] = nblk;
        if (nblk > 0) hipLaunchKernelGGL(phase_weights_batch_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, b);
    }
    return rcf_launch_status();
}

extern "C" int rcf_phase_wgrad_fold(const float* dwp, float* dw_oihw, int o, int i, void* stream) {
    if (!dwp || !dw_oihw || o <= 0 || i <= 0) return RCF_EINVAL;
    const int total = 9 * o * i;
    hipLaunchKernelGGL(phase_wgrad_fold_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, dwp, dw_oihw, o, i);
    return rcf_launch_status();
}

#endif   // !RCF_CONV_B16

#if !RCF_CONV_B16
extern "C" int rcf_phase_wgrad_gather_s2(const float* dwp, float* dw_oihw, int o, int i, void* stream) {
    if (!dwp || !dw_oihw || o <= 0 || i <= 0) return RCF_EINVAL;
    const int total = 9 * o * i;
    hipLaunchKernelGGL(phase_wgrad_gather_s2_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, dwp, dw_oihw, o, i);
    return rcf_launch_status();
}
#endif

// may this descriptor's launch take the BatchNorm-backward sums of the block whose output gradient it writes?  It must write every
// element of a plain NHWC tensor exactly once (or add to it as the LAST writer: the caller's business), on a kernel that has the
// epilogue (split_bst_ok), with one source (an input gradient has one) and no statistics of its own
static bool bn_sums_ok(const rcf_conv_desc* d, const Sel& s) {
    // (not accumulating: the launch is the only writer of dY -- the kernels' BatchNorm-sums variants contain no += path)
    return split_bst_ok(s) && d->c2 == 0 && d->out_stride == 1 && d->out_off_y == 0 && d->out_off_x == 0 &&
           d->out_h_phys == d->h_out && d->out_w_phys == d->w_out && d->accumulate == 0;
}

extern "C" int RCF_FN(rcf_conv2d_query)(const rcf_conv_desc* d, rcf_conv_info* info) {
    RCF_TO_B16(d, rcf_conv2d_query_b16impl(d, info));
    if (!info) return RCF_EINVAL;
    Sel s;
    int rc = select_cfg(d, &s);
    if (rc != RCF_OK) return rc;
    ConvArgs a;
    fill_args(d, s, &a);
    const int ntile_n = ceil_div(d->c_out, s.bn);
    info->packed_weight_floats = (size_t)ntile_n * (a.nchunk1 + a.nchunk2) * s.t * s.bn * s.ck;
    if (s.split) info->packed_weight_floats = (size_t)ntile_n * (a.nchunk1 + a.nchunk2) * s.t * s.bn * 8 * s.npl;   // 16 bf16 x planes per row
#if !RCF_CONV_B16
    if (s.pw) info->n_partials = pw2_grid((long long)d->n * d->h_out * d->w_out, d->c1 / 16, s.nt);
    else
#endif
#if RCF_CONV_B16
    if (s.pw) info->n_partials = pw_grid((long long)d->n * d->h_out * d->w_out, d->c1 / 16, s.nt, ceil_div(d->c_out, 32 * s.nt));
    else if (s.dma) info->n_partials = dispatch_dma(s, [&](auto tag) { return dma_grid_x<typename decltype(tag)::type>(a.ntiles, ntile_n); });
    else
#endif
    info->n_partials = s.split ? dispatch_split(s, [&](auto tag) { return split_grid_x<typename decltype(tag)::type>(a.ntiles, ntile_n); })
                               : dispatch_fwd(s, [&](auto tag) { return fwd_grid_x<typename decltype(tag)::type>(a.ntiles, ntile_n); });
    if (info->n_partials <= 0) return RCF_EUNSUPPORTED;
    // (the stem on the space-to-depth image reports kind 3 like the 7x7 stem it stands for: 3000 + 5000 (split) stays below the
    // weight-gradient ids, 10000 + ...)
    info->kernel_id = (s.kind == K4S1 ? (int)K7S2 : s.kind) * 1000 + s.ck * 10 + s.nt + (s.px == 16 ? 100 : (s.px == 8 ? 200 : 0)) + (s.vt ? 400 : 0) + (s.split ? 5000 : 0) + (s.small ? 5 : 0) + (s.bf16 ? 20000 : 0) + ((s.split && s.npl == 2) ? 40000 : 0);
    info->wgrad_workspace_floats = 0;
    info->wgrad_kernel_id = 0;
    info->bn_on_load = (s.split && !s.dma && !s.pw && d->w_mode == RCF_W_FORWARD && d->c1 + d->c2 <= 512 && s.npl != 2) ? 1 : 0;   // a DMA cannot transform; fp16 planes need the maximum of the TRANSFORMED tensor
    info->wgrad_bn_on_load = 0;
    info->fwd_act = (s.split && !s.pw && d->w_mode == RCF_W_FORWARD && !d->accumulate) ? 1 : 0;
    info->bn_bwd_sums = bn_sums_ok(d, s) ? 1 : 0;
    if (d->w_mode == RCF_W_FORWARD) {
        WSel w;
        if (select_wgrad(d, &w) == RCF_OK) {
            // (the four-phase launches -- phase_sum 1 / 2 -- keep one row block per (phase, slot): wgrad_phase_slots)
            const int nrows = (d->phase_sum == 1 || d->phase_sum == 2) ? 4 * wgrad_phase_slots(w.nsplit, d) : w.nsplit;
            info->wgrad_workspace_floats = (size_t)nrows * w.ktot * w.cop + 64;   // + a zero page for the DMA path
            info->wgrad_bn_on_load = (w.split && !SAct::B16 && d->precision != RCF_PREC_F16X2) ? 1 : 0;   // bf16 tensors are staged raw: nothing to apply BatchNorm to
            info->wgrad_kernel_id = 10000 + w.kind * 1000 + (w.px == 16 ? 100 : (w.px == 8 ? 200 : 0)) + (w.vt ? 400 : 0) + (wgrad_tr_ok(d, w) ? 200 : 0) +
                                    (w.split ? 5000 + w.wci * 10 + w.wco : 0) + ((w.split && d->precision == RCF_PREC_BF16) ? 20000 : 0) +
                                    ((w.split && d->precision == RCF_PREC_F16X2) ? 40000 : 0);
        }
    }
    return RCF_OK;
}

// arguments of one weight packing (shared by the single and the batched entry point)
static int pack_args(const rcf_conv_desc* d, const float* w_oihw, float* packed, const float* amax_w, PackArgs* p, unsigned* blocks) {
    if (!d || !w_oihw || !packed) return RCF_EINVAL;
    Sel s;
    int rc = select_cfg(d, &s);
    if (rc != RCF_OK) return rc;
    ConvArgs a;
    fill_args(d, s, &a);
    const int ntile_n = ceil_div(d->c_out, s.bn);
    const int nchunk = a.nchunk1 + a.nchunk2;
    p->w = w_oihw; p->dst = packed; p->amax = amax_w;
    p->w_o = d->w_o; p->w_i = d->w_i; p->ks = d->ksize; p->mode = d->w_mode; p->i_off = d->w_i_off; p->c_out = d->c_out;
    p->c1 = d->c1; p->c2 = d->c2; p->nchunk1 = a.nchunk1; p->nchunk = nchunk; p->T = s.t; p->BN = s.bn; p->CK = s.ck;
    p->ksx = s.kind == K7S2 ? 1 : d->ksize; p->kind = s.kind == K7S2 ? 1 : 0; p->npl = s.npl; p->split = s.split ? 1 : 0;
    p->total = s.split ? (unsigned long long)ntile_n * nchunk * s.t * s.bn * 16 : (unsigned long long)ntile_n * nchunk * s.t * s.bn * s.ck;
    *blocks = (unsigned)((p->total + 255) / 256);
    return RCF_OK;
}

static int pack_weights_impl(const rcf_conv_desc* d, const float* w_oihw, float* packed, const float* amax_w, void* stream) {
    if (!w_oihw || !packed) return RCF_EINVAL;
    PackArgs p;
    unsigned blocks = 0;
    const int rc = pack_args(d, w_oihw, packed, amax_w, &p, &blocks);
    if (rc != RCF_OK) return rc;
    if (p.split)
        hipLaunchKernelGGL(pack_weights_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p.w, static_cast<unsigned short*>(p.dst),
                           (size_t)p.total, p.w_o, p.w_i, p.mode, p.i_off, p.c_out, p.c1, p.c2, p.nchunk1, p.nchunk, p.BN, p.ks, p.npl, p.amax);
    else
        hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p.w, static_cast<float*>(p.dst), (size_t)p.total,
                           p.w_o, p.w_i, p.ks, p.mode, p.i_off, p.c_out, p.c1, p.c2, p.nchunk1, p.nchunk, p.T, p.ksx, p.BN, p.CK, p.kind);
    return rcf_launch_status();
}

extern "C" int RCF_FN(rcf_conv2d_pack_weights)(const rcf_conv_desc* d, const float* w_oihw, float* packed, void* stream) {
    RCF_TO_B16(d, rcf_conv2d_pack_weights_b16impl(d, w_oihw, packed, stream));
    return pack_weights_impl(d, w_oihw, packed, nullptr, stream);
}
#if !RCF_CONV_B16
extern "C" int rcf_conv2d_pack_weights_scaled(const rcf_conv_desc* d, const float* w_oihw, float* packed, const float* amax_w, void* stream) {
    if (d != nullptr && (d->storage == RCF_STORE_BF16 || d->precision != RCF_PREC_F16X2)) return RCF_EUNSUPPORTED;
    return pack_weights_impl(d, w_oihw, packed, amax_w, stream);
}
#endif

// rcf_conv2d_pack_weights_batch: the items of THIS translation unit's storage, PACK_BATCH per launch
extern "C" int RCF_FN(rcf_conv2d_pack_weights_batch)(const rcf_pack_item* items, int n, void* stream) {
    if (!items || n <= 0) return RCF_EINVAL;
#if !RCF_CONV_B16
    {   // bf16-tensor descriptors go to the bf16 unit as one batch (a model uses one storage: all or nothing is the common case)
        int nb = 0;
        for (int i = 0; i < n; ++i) {
            if (!items[i].desc || !items[i].w_oihw || !items[i].packed) return RCF_EINVAL;
            nb += items[i].desc->storage == RCF_STORE_BF16;
        }
        if (nb == n && n > 0) return rcf_conv2d_pack_weights_batch_b16impl(items, n, stream);
        if (nb > 0) {
            for (int i = 0; i < n; ++i) {
                const int rc = items[i].desc->storage == RCF_STORE_BF16 ? rcf_conv2d_pack_weights_batch_b16impl(items + i, 1, stream)
                                                                          : rcf_conv2d_pack_weights_batch(items + i, 1, stream);
                if (rc != RCF_OK) return rc;
            }
            return RCF_OK;
        }
    }
#endif
    static thread_local PackBatch b;   // ~3 KB: not on the stack of a ctypes caller's thread; one per host thread
    int i = 0;
    while (i < n) {
        b.n = 0;
        unsigned nblk = 0;
        while (i < n && b.n < PACK_BATCH) {
            unsigned blocks = 0;
            const int rc = pack_args(items[i].desc, items[i].w_oihw, items[i].packed,
                                     items[i].desc->precision == RCF_PREC_F16X2 ? items[i].amax_w : nullptr, &b.it[b.n], &blocks);
            if (rc != RCF_OK) return rc;
            b.blk_start[b.n] = nblk;
            nblk += blocks;
            ++b.n;
            ++i;
        }
        b.blk_start[b.n] = nblk;
        for (int j = b.n + 1; j <= PACK_BATCH; ++j) b.blk_start[j] = nblk;
        if (nblk > 0) hipLaunchKernelGGL(pack_weights_batch_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, b);
    }
    return rcf_launch_status();
}

static int conv2d_fwd_impl(const rcf_conv_desc* d, const float* in1, const float* coef1, const float* in2, const float* coef2,
                           const float* packed, float* out, double* stat_partials, void* stream, const rcf_conv_scales* sc = nullptr,
                           const float* bn_z = nullptr, const float* bn_coef = nullptr);

extern "C" int RCF_FN(rcf_conv2d_fwd)(const rcf_conv_desc* d, const void* in1, const void* in2, const float* packed, void* out,
                                      double* stat_partials, void* stream) {
    RCF_TO_B16(d, rcf_conv2d_fwd_b16impl(d, in1, in2, packed, out, stat_partials, stream));
    return conv2d_fwd_impl(d, (const float*)in1, nullptr, (const float*)in2, nullptr, packed, (float*)out, stat_partials, stream);
}

extern "C" int RCF_FN(rcf_conv2d_fwd_bn)(const rcf_conv_desc* d, const void* in1, const float* coef1, const void* in2, const float* coef2,
                                         const float* packed, void* out, double* stat_partials, void* stream) {
    RCF_TO_B16(d, rcf_conv2d_fwd_bn_b16impl(d, in1, coef1, in2, coef2, packed, out, stat_partials, stream));
    return conv2d_fwd_impl(d, (const float*)in1, coef1, (const float*)in2, coef2, packed, (float*)out, stat_partials, stream);
}

#if !RCF_CONV_B16
extern "C" int rcf_conv2d_fwd_scaled(const rcf_conv_desc* d, const void* in1, const void* in2, const float* packed, void* out,
                                     double* stat_partials, const rcf_conv_scales* scales, void* stream) {
    if (!scales) return RCF_EINVAL;
    if (d != nullptr && (d->storage == RCF_STORE_BF16 || d->precision != RCF_PREC_F16X2)) return RCF_EUNSUPPORTED;
    return conv2d_fwd_impl(d, (const float*)in1, nullptr, (const float*)in2, nullptr, packed, (float*)out, stat_partials, stream, scales);
}
#endif

extern "C" int RCF_FN(rcf_conv2d_dgrad_bn_sums)(const rcf_conv_desc* d, const void* dz, const float* packed, void* dx, const void* bn_z,
                                                const float* bn_coef, double* sum_partials, const rcf_conv_scales* scales, void* stream) {
    RCF_TO_B16(d, rcf_conv2d_dgrad_bn_sums_b16impl(d, dz, packed, dx, bn_z, bn_coef, sum_partials, scales, stream));
    if (!bn_z || !bn_coef || !sum_partials) return RCF_EINVAL;
#if RCF_CONV_B16
    if (d != nullptr && d->precision != RCF_PREC_BF16) return RCF_EUNSUPPORTED;
    scales = nullptr;   // bf16 operands carry no per-tensor scale
#else
    if (d != nullptr && d->precision != RCF_PREC_F16X2) return RCF_EUNSUPPORTED;
#endif
    return conv2d_fwd_impl(d, (const float*)dz, nullptr, nullptr, nullptr, packed, (float*)dx, sum_partials, stream, scales,
                           (const float*)bn_z, bn_coef);
}

static int conv2d_fwd_impl(const rcf_conv_desc* d, const float* in1, const float* coef1, const float* in2, const float* coef2,
                           const float* packed, float* out, double* stat_partials, void* stream, const rcf_conv_scales* sc,
                           const float* bn_z, const float* bn_coef) {
    if (!in1 || !packed || !out) return RCF_EINVAL;
    Sel s;
    int rc = select_cfg(d, &s);
    if (rc != RCF_OK) return rc;
    if (d->c2 > 0 && !in2) return RCF_EINVAL;
    if ((coef1 || coef2) && (!s.split || d->w_mode != RCF_W_FORWARD || d->c1 + d->c2 > 512 || (coef2 && d->c2 == 0)))
        return RCF_EUNSUPPORTED;   // BN-on-load exists in the split kernels only (rcf_conv_info.bn_on_load)
    ConvArgs a;
    fill_args(d, s, &a);
    if (a.zero == nullptr) return (int)hipErrorInvalidSymbol;
    a.coef1 = coef1; a.coef2 = coef2;
    a.in1 = in1; a.in2 = in2; a.wp = packed; a.out = out; a.stats = stat_partials; a.dz = nullptr; a.ws = nullptr;
    a.ktot = 0; a.cop = 0;
    if (sc != nullptr) {
        if (!s.split || s.npl != 2) return RCF_EUNSUPPORTED;   // scales belong to the two-plane fp16 split kernels
        a.amax_a1 = sc->amax_in1; a.amax_a2 = d->c2 > 0 ? sc->amax_in2 : nullptr; a.amax_b = sc->amax_w;
    }
    const int nn = ceil_div(d->c_out, s.bn);
    a.wp_phase_stride = (int)((size_t)nn * (a.nchunk1 + a.nchunk2) * s.t * s.bn * (s.split ? 8 * s.npl : s.ck));
#if RCF_CONV_B16
    if (s.pw) {
        if (coef1 || coef2) return RCF_EUNSUPPORTED;
        return dispatch_pw(d->c1 / 16, s.nt, [&](auto cfg) { return launch_pw<decltype(cfg)>(a, (hipStream_t)stream); });
    }
    if (s.dma && bn_z != nullptr) {   // rcf_conv_info.bn_bwd_sums
        if (!bn_sums_ok(d, s) || coef1 || coef2) return RCF_EUNSUPPORTED;
        a.bz = bn_z; a.bk = bn_coef;
        const bool p16 = s.px == 16;
        hipStream_t st = (hipStream_t)stream;
        if (s.kind == K2S1) {
            if (s.nt == 1) return p16 ? launch_dma_bst<D2_1_16>(a, nn, st) : launch_dma_bst<D2_1_32>(a, nn, st);
            return p16 ? launch_dma_bst<D2_2_16>(a, nn, st) : launch_dma_bst<D2_2_32>(a, nn, st);
        }
        if (s.nt == 1 && s.small) return p16 ? launch_dma_bst<D3_1_16s>(a, nn, st) : launch_dma_bst<D3_1_32s>(a, nn, st);
        if (s.nt == 1) return p16 ? launch_dma_bst<D3_1_16>(a, nn, st) : launch_dma_bst<D3_1_32>(a, nn, st);
        return p16 ? launch_dma_bst<D3_2_16>(a, nn, st) : launch_dma_bst<D3_2_32>(a, nn, st);
    }
    if (s.dma && !coef1 && !coef2)
        return dispatch_dma(s, [&](auto tag) { return launch_dma<typename decltype(tag)::type, false>(a, nn, (hipStream_t)stream); });
    if (s.dma) return RCF_EUNSUPPORTED;   // rcf_conv_info.bn_on_load is 0 for these descriptors (the tile geometry differs)
#endif
    if (s.kind == K4S1 && SAct::B16) return RCF_EUNSUPPORTED;
#if !RCF_CONV_B16
    if (s.pw) {
        if (coef1 || coef2 || bn_z) return RCF_EUNSUPPORTED;
        return dispatch_pw2(d->c1 / 16, s.nt, [&](auto cfg) { return launch_pw2<decltype(cfg)>(a, (hipStream_t)stream); });
    }
#endif
    if (bn_z != nullptr) {   // rcf_conv_info.bn_bwd_sums: an input-gradient launch over a plain (unit-stride, whole) output tensor
        if (!bn_sums_ok(d, s)) return RCF_EUNSUPPORTED;
        a.bz = bn_z; a.bk = bn_coef;
        return dispatch_split_bst(s, [&](auto tag) { return launch_split_bst<typename decltype(tag)::type>(a, nn, (hipStream_t)stream); });
    }
    if (s.split) return dispatch_split(s, [&](auto tag) { return launch_split<typename decltype(tag)::type>(a, nn, (hipStream_t)stream); });
    return dispatch_fwd(s, [&](auto tag) { return launch_fwd<typename decltype(tag)::type>(a, nn, (hipStream_t)stream); });
}

extern "C" int RCF_FN(rcf_conv2d_fwd_act)(const rcf_conv_desc* d, const void* in1_, const void* in2_, const float* packed,
                                          const float* bias, const void* res_, void* out_, void* stream) {
    RCF_TO_B16(d, rcf_conv2d_fwd_act_b16impl(d, in1_, in2_, packed, bias, res_, out_, stream));
    const float* in1 = (const float*)in1_;
    const float* in2 = (const float*)in2_;
    const float* res = (const float*)res_;
    float* out = (float*)out_;
    if (!in1 || !packed || !out || !bias) return RCF_EINVAL;
    Sel s;
    int rc = select_cfg(d, &s);
    if (rc != RCF_OK) return rc;
    if (d->c2 > 0 && !in2) return RCF_EINVAL;
    if (!s.split || s.pw || d->w_mode != RCF_W_FORWARD || d->accumulate) return RCF_EUNSUPPORTED;   // rcf_conv_info.fwd_act
    ConvArgs a;
    fill_args(d, s, &a);
    if (a.zero == nullptr) return (int)hipErrorInvalidSymbol;
    a.bias = bias; a.res = res;
    a.in1 = in1; a.in2 = in2; a.wp = packed; a.out = out; a.stats = nullptr; a.dz = nullptr; a.ws = nullptr;
    a.ktot = 0; a.cop = 0;
    const int nn = ceil_div(d->c_out, s.bn);
    a.wp_phase_stride = (int)((size_t)nn * (a.nchunk1 + a.nchunk2) * s.t * s.bn * 8 * s.npl);
#if RCF_CONV_B16
    if (s.dma) return dispatch_dma(s, [&](auto tag) { return launch_dma<typename decltype(tag)::type, true>(a, nn, (hipStream_t)stream); });
#endif
    return dispatch_split(s, [&](auto tag) { return launch_split<typename decltype(tag)::type, true>(a, nn, (hipStream_t)stream); });
}

// split weight-gradient kernels on fp32 tensors: NPL = 3 (exact) or 2 (RCF_PREC_F16X2); same tilings (select_wgrad)
template <int NPL>
static int launch_wgrad_split_planes(const ConvArgs& a, const WSel& w, int cfg, hipStream_t st) {
    if constexpr (NPL == 2)
    if (w.kind == K2S1 && a.phase_sum == 2) {   // the up-2x weight gradient: phase pairs from one x tile
        if (cfg == 22) return launch_wgrad_split<WsCfg<2, 2, 2, 8, NPL, true>>(a, w.nsplit, w.gy, w.gz, st);
        if (cfg == 12) return launch_wgrad_split<WsCfg<1, 2, 2, 8, NPL, true>>(a, w.nsplit, w.gy, w.gz, st);
        if (cfg == 21) return launch_wgrad_split<WsCfg<2, 1, 2, 8, NPL, true>>(a, w.nsplit, w.gy, w.gz, st);
        return launch_wgrad_split<WsCfg<1, 1, 2, 16, NPL, true>>(a, w.nsplit, w.gy, w.gz, st);
    }
    if (w.kind == K2S1) {
        if (cfg == 22) return launch_wgrad_split<WsCfg<2, 2, 2, 8, NPL>>(a, w.nsplit, w.gy, w.gz, st);
        if (cfg == 12) return launch_wgrad_split<WsCfg<1, 2, 2, 8, NPL>>(a, w.nsplit, w.gy, w.gz, st);
        if (cfg == 21) return launch_wgrad_split<WsCfg<2, 1, 2, 8, NPL>>(a, w.nsplit, w.gy, w.gz, st);
        return launch_wgrad_split<WsCfg<1, 1, 2, 16, NPL>>(a, w.nsplit, w.gy, w.gz, st);
    }
    if (cfg == 22) return launch_wgrad_split<WsCfg<2, 2, 3, 8, NPL>>(a, w.nsplit, w.gy, w.gz, st);
    if (cfg == 12) return launch_wgrad_split<WsCfg<1, 2, 3, 8, NPL>>(a, w.nsplit, w.gy, w.gz, st);
    if (cfg == 21) return launch_wgrad_split<WsCfg<2, 1, 3, 8, NPL>>(a, w.nsplit, w.gy, w.gz, st);
    return launch_wgrad_split<WsCfg<1, 1, 3, 16, NPL>>(a, w.nsplit, w.gy, w.gz, st);
}

static int conv2d_wgrad_impl(const rcf_conv_desc* d, const float* in1, const float* coef1, const float* in2, const float* coef2,
                             const float* dz, float* dw_oihw, float* workspace, void* stream, const rcf_conv_scales* sc = nullptr);

extern "C" int RCF_FN(rcf_conv2d_wgrad)(const rcf_conv_desc* d, const void* in1, const void* in2, const void* dz,
                                        float* dw_oihw, float* workspace, void* stream) {
    RCF_TO_B16(d, rcf_conv2d_wgrad_b16impl(d, in1, in2, dz, dw_oihw, workspace, stream));
    return conv2d_wgrad_impl(d, (const float*)in1, nullptr, (const float*)in2, nullptr, (const float*)dz, dw_oihw, workspace, stream);
}

extern "C" int RCF_FN(rcf_conv2d_wgrad_bn)(const rcf_conv_desc* d, const void* in1, const float* coef1, const void* in2,
                                           const float* coef2, const void* dz, float* dw_oihw, float* workspace, void* stream) {
    RCF_TO_B16(d, rcf_conv2d_wgrad_bn_b16impl(d, in1, coef1, in2, coef2, dz, dw_oihw, workspace, stream));
    return conv2d_wgrad_impl(d, (const float*)in1, coef1, (const float*)in2, coef2, (const float*)dz, dw_oihw, workspace, stream);
}

#if !RCF_CONV_B16
extern "C" int rcf_conv2d_wgrad_scaled(const rcf_conv_desc* d, const void* in1, const void* in2, const void* dz, float* dw_oihw,
                                       float* workspace, const rcf_conv_scales* scales, void* stream) {
    if (!scales) return RCF_EINVAL;
    if (d != nullptr && (d->storage == RCF_STORE_BF16 || d->precision != RCF_PREC_F16X2)) return RCF_EUNSUPPORTED;
    return conv2d_wgrad_impl(d, (const float*)in1, nullptr, (const float*)in2, nullptr, (const float*)dz, dw_oihw, workspace, stream, scales);
}
#endif

static int conv2d_wgrad_impl(const rcf_conv_desc* d, const float* in1, const float* coef1, const float* in2, const float* coef2,
                             const float* dz, float* dw_oihw, float* workspace, void* stream, const rcf_conv_scales* sc) {
    if (!in1 || !dz || !dw_oihw || !workspace) return RCF_EINVAL;
    WSel w;
    int rc = select_wgrad(d, &w);
    if (rc != RCF_OK) return rc;
    if (d->c2 > 0 && !in2) return RCF_EINVAL;
    if ((coef1 || coef2) && (!w.split || SAct::B16 || (coef2 && d->c2 == 0))) return RCF_EUNSUPPORTED;
    ConvArgs a;
    a.xcd_band = 0;
    a.bias = nullptr; a.res = nullptr;
    a.amax_a1 = nullptr; a.amax_a2 = nullptr; a.amax_b = nullptr;
    a.bz = nullptr; a.bk = nullptr;
    if (sc != nullptr) {
        if (!w.split || d->precision != RCF_PREC_F16X2) return RCF_EUNSUPPORTED;
        a.amax_a1 = sc->amax_in1; a.amax_a2 = d->c2 > 0 ? sc->amax_in2 : nullptr; a.amax_b = sc->amax_dz;
    }
    a.zero = zero_page_ptr();
    if (a.zero == nullptr) return (int)hipErrorInvalidSymbol;
    a.coef1 = coef1; a.coef2 = coef2;
    a.n = d->n; a.h_in = d->h_in; a.w_in = d->w_in; a.c1 = d->c1; a.c2 = d->c2;
    a.h1 = d->h_src1; a.w1 = d->w_src1; a.gather1 = d->gather1;
    a.h_out = d->h_out; a.w_out = d->w_out; a.c_out = d->c_out; a.pad = d->pad; a.pad_x = d->pad_x; a.stride = d->stride;
    a.gstep = d->ksize == 1 ? d->stride : 1;
    a.accumulate = 0;
    a.os = d->out_stride; a.ooy = d->out_off_y; a.oox = d->out_off_x; a.ohp = d->out_h_phys; a.owp = d->out_w_phys;
    a.ioy = d->in_off_y; a.iox = d->in_off_x;
    a.phase_sum = 0; a.wp_phase_stride = 0;
    // phase_sum == 2 (the merged up-2x forward descriptor): the four phases' weight gradients in one launch of the split kernel, into
    // dw[4][c_out][c_in][2][2]; each phase keeps its own workspace rows and its own reduction
    // phase_sum == 1 on a RCF_GATHER_STRIDED2 descriptor (in_off_* ignored): the four phase weight gradients of a 3x3 stride-2
    // convolution (x gathered at (2y + a, 2x + b), the same dz) in one launch, into dw[4][c_out][c_in][2][2]
    int nslot4 = 0;
    if (d->phase_sum == 1 || d->phase_sum == 2) {
        if (!w.split || w.kind != K2S1 || (d->phase_sum == 1 && d->gather1 != RCF_GATHER_STRIDED2)) return RCF_EUNSUPPORTED;
        nslot4 = wgrad_phase_slots(w.nsplit, d);
        a.phase_sum = d->phase_sum;
        w.nsplit = (wgrad_phase_pairs(d) ? 2 : 4) * nslot4;   // workgroups: (slot, a) pairs / (slot, phase)
    }
    a.vt = w.vt; a.hp = d->h_out + 1; a.nimg = d->n; a.inv_hp = 1.0f / (float)(d->h_out + 1);
    a.sy = (float)d->h_src1 / (float)d->h_in;
    a.sx = (float)d->w_src1 / (float)d->w_in;
    a.tiles_x = w.tiles_x; a.tiles_y = w.tiles_y; a.ntiles = w.ntiles;
    a.nchunk1 = w.nchunk1; a.nchunk2 = w.nchunk2;
    a.in1 = in1; a.in2 = in2; a.wp = nullptr; a.out = nullptr; a.stats = nullptr; a.dz = dz; a.ws = workspace;
    a.ktot = w.ktot; a.cop = w.cop;
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = w.nchunk1 + w.nchunk2;
    const int p16 = w.px == 16;
    const bool dma = !SAct::B16 && w.kind != K7S2 && (d->c1 % 4 == 0) && (d->c2 % 4 == 0);
    if (w.split) {
        if (w.kind == K1 && d->stride == 2) {   // x at the even positions of the physical h_in x w_in tensor; logical input = output grid
            a.h_in = d->h_out; a.w_in = d->w_out; a.h1 = d->h_in; a.w1 = d->w_in;
            a.gather1 = RCF_GATHER_STRIDED2; a.ioy = 0; a.iox = 0;
        }
        a.nchunk1 = ceil_div(d->c1, 32 * w.wci);
        a.nchunk2 = d->c2 > 0 ? ceil_div(d->c2, 32 * w.wci) : 0;
        const int cfg = w.wci * 10 + w.wco;
        if (wgrad_tr_ok(d, w) && !coef1 && !coef2) {
#ifdef RCF_WGRAD_DIAG
            { const char* dg = getenv("RCF_WGRAD_DIAG"); if (dg && dg[0] >= '1' && dg[0] <= '3') a.xcd_band = 76 + (dg[0] - '0'); }
#endif
            constexpr int NPT = SAct::B16 ? 1 : 2;       // operand planes
            constexpr int THT = SAct::B16 ? 16 : 8;      // tile rows of the 64-channel configurations (select_wgrad: th_split)
            const int ks = w.kind == K3S1 ? 3 : (w.kind == K2S1 ? 2 : 1);
#define RCF_TR_KS(KSV)                                                                                                  \
            if (cfg == 22) rc = launch_wgrad_tr<WtCfg<2, 2, KSV, THT, NPT>>(a, w.nsplit, w.gy, w.gz, st);               \
            else if (cfg == 12) rc = launch_wgrad_tr<WtCfg<1, 2, KSV, THT, NPT>>(a, w.nsplit, w.gy, w.gz, st);          \
            else if (cfg == 21) rc = launch_wgrad_tr<WtCfg<2, 1, KSV, THT, NPT>>(a, w.nsplit, w.gy, w.gz, st);          \
            else rc = launch_wgrad_tr<WtCfg<1, 1, KSV, 16, NPT>>(a, w.nsplit, w.gy, w.gz, st);
            if (ks == 3) { RCF_TR_KS(3) }
            else if (ks == 2) { RCF_TR_KS(2) }
            else {
                if constexpr (SAct::B16) { RCF_TR_KS(1) } else return RCF_EUNSUPPORTED;
            }
#undef RCF_TR_KS
        } else
        if (d->precision == RCF_PREC_BF16) {
            constexpr int THB = SAct::B16 ? 16 : 8;   // tile rows (select_wgrad: th_split)
            if (w.kind == K1) {
                if constexpr (SAct::B16) {
                    if (cfg == 22) rc = launch_wgrad_split<WsCfg<2, 2, 1, 16, 1>>(a, w.nsplit, w.gy, w.gz, st);
                    else if (cfg == 12) rc = launch_wgrad_split<WsCfg<1, 2, 1, 16, 1>>(a, w.nsplit, w.gy, w.gz, st);
                    else if (cfg == 21) rc = launch_wgrad_split<WsCfg<2, 1, 1, 16, 1>>(a, w.nsplit, w.gy, w.gz, st);
                    else rc = launch_wgrad_split<WsCfg<1, 1, 1, 16, 1>>(a, w.nsplit, w.gy, w.gz, st);
                } else return RCF_EUNSUPPORTED;
            } else if (w.kind == K2S1 && a.phase_sum == 2) {
                if (cfg == 22) rc = launch_wgrad_split<WsCfg<2, 2, 2, THB, 1, true>>(a, w.nsplit, w.gy, w.gz, st);
                else if (cfg == 12) rc = launch_wgrad_split<WsCfg<1, 2, 2, THB, 1, true>>(a, w.nsplit, w.gy, w.gz, st);
                else if (cfg == 21) rc = launch_wgrad_split<WsCfg<2, 1, 2, THB, 1, true>>(a, w.nsplit, w.gy, w.gz, st);
                else rc = launch_wgrad_split<WsCfg<1, 1, 2, 16, 1, true>>(a, w.nsplit, w.gy, w.gz, st);
            } else if (w.kind == K2S1) {
                if (cfg == 22) rc = launch_wgrad_split<WsCfg<2, 2, 2, THB, 1>>(a, w.nsplit, w.gy, w.gz, st);
                else if (cfg == 12) rc = launch_wgrad_split<WsCfg<1, 2, 2, THB, 1>>(a, w.nsplit, w.gy, w.gz, st);
                else if (cfg == 21) rc = launch_wgrad_split<WsCfg<2, 1, 2, THB, 1>>(a, w.nsplit, w.gy, w.gz, st);
                else rc = launch_wgrad_split<WsCfg<1, 1, 2, 16, 1>>(a, w.nsplit, w.gy, w.gz, st);
            } else if (cfg == 22) rc = launch_wgrad_split<WsCfg<2, 2, 3, THB, 1>>(a, w.nsplit, w.gy, w.gz, st);
            else if (cfg == 12) rc = launch_wgrad_split<WsCfg<1, 2, 3, THB, 1>>(a, w.nsplit, w.gy, w.gz, st);
            else if (cfg == 21) rc = launch_wgrad_split<WsCfg<2, 1, 3, THB, 1>>(a, w.nsplit, w.gy, w.gz, st);
            else rc = launch_wgrad_split<WsCfg<1, 1, 3, 16, 1>>(a, w.nsplit, w.gy, w.gz, st);
        } else if constexpr (!SAct::B16) {
            rc = d->precision == RCF_PREC_F16X2 ? launch_wgrad_split_planes<2>(a, w, cfg, st) : launch_wgrad_split_planes<3>(a, w, cfg, st);
        } else return RCF_EUNSUPPORTED;
    } else
    switch (w.kind) {
        case K3S1:
            if (dma && w.px == 8) rc = RCF_DMA(launch_wgrad_dma<W3S1_8>(a, w.nsplit, nchunk, w.ncog, st));
            else if (dma) rc = RCF_DMA(p16 ? launch_wgrad_dma<W3S1_16>(a, w.nsplit, nchunk, w.ncog, st) : launch_wgrad_dma<W3S1_32>(a, w.nsplit, nchunk, w.ncog, st));
            else rc = p16 ? launch_wgrad<W3S1_16>(a, w.nsplit, nchunk, w.ncog, st) : launch_wgrad<W3S1_32>(a, w.nsplit, nchunk, w.ncog, st);
            break;
        case K3S2:
            if (dma) rc = RCF_DMA(p16 ? launch_wgrad_dma<W3S2_16>(a, w.nsplit, nchunk, w.ncog, st) : launch_wgrad_dma<W3S2_32>(a, w.nsplit, nchunk, w.ncog, st));
            else rc = p16 ? launch_wgrad<W3S2_16>(a, w.nsplit, nchunk, w.ncog, st) : launch_wgrad<W3S2_32>(a, w.nsplit, nchunk, w.ncog, st);
            break;
        case K1:
            if (dma) rc = RCF_DMA(p16 ? launch_wgrad_dma<W1_16>(a, w.nsplit, nchunk, w.ncog, st) : launch_wgrad_dma<W1_32>(a, w.nsplit, nchunk, w.ncog, st));
            else rc = p16 ? launch_wgrad<W1_16>(a, w.nsplit, nchunk, w.ncog, st) : launch_wgrad<W1_32>(a, w.nsplit, nchunk, w.ncog, st);
            break;
        case K2S1:
            if (dma) rc = RCF_DMA(p16 ? launch_wgrad_dma<W2_16>(a, w.nsplit, nchunk, w.ncog, st) : launch_wgrad_dma<W2_32>(a, w.nsplit, nchunk, w.ncog, st));
            else rc = p16 ? launch_wgrad<W2_16>(a, w.nsplit, nchunk, w.ncog, st) : launch_wgrad<W2_32>(a, w.nsplit, nchunk, w.ncog, st);
            break;
        case K7S2: rc = p16 ? launch_wgrad<W7_16>(a, w.nsplit, nchunk, w.ncog, st) : launch_wgrad<W7_32>(a, w.nsplit, nchunk, w.ncog, st); break;
        default: return RCF_EUNSUPPORTED;
    }
    if (rc != RCF_OK) return rc;
    const int total = w.ktot * w.cop;
    const int ksx = w.kind == K7S2 ? 1 : d->ksize;
    if (nslot4 > 0) {   // the four phases' reductions in one launch (grid y = phase)
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 63) / 64, 4), dim3(256), 0, st, workspace, dw_oihw, nslot4, w.ktot, w.cop,
                           d->c_out, d->c1, d->c2, w.nchunk1, w.t, ksx, 0);
        return rcf_launch_status();
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, st, workspace, dw_oihw, w.nsplit, w.ktot,
                       w.cop, d->c_out, d->c1, d->c2, w.nchunk1, w.t, ksx, w.kind == K7S2 ? 1 : 0);
    return rcf_launch_status();
}

#if defined(RCF_PHASE_TIMING)
// diagnostics build only: summed s_memtime cycles of all conv_split_kernel waves since the last reset (see RCF_TACC slots)
extern "C" int RCF_FN(rcf_debug_phase_cycles)(unsigned long long* out8, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return rcf_launch_status();
    if (out8 != nullptr && hipMemcpyFromSymbol(out8, HIP_SYMBOL(rcf_phase_cycles), 8 * sizeof(unsigned long long)) != hipSuccess)
        return rcf_launch_status();
    if (reset) {
        const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(rcf_phase_cycles), z, sizeof(z)) != hipSuccess) return rcf_launch_status();
    }
    return RCF_OK;
}
#endif
#endif   // RCF_CONV_KERNELS_ONLY
