// conv1x1_f16x2_kernel: 1x1 convolutions (the encoder's fusion convs W1 d, W2 d and the ResNet projections, src/networks.py:863-866,
// src/net_utils.py:300-307) and their stride-1 input gradients on fp32 NHWC tensors, two scaled fp16 operand planes (RCF_PREC_F16X2).
// Included by rcf_conv_impl.h in the fp32 translation unit only; the bf16-tensor twin is conv1x1_b16_kernel (rcf_conv_b16_dma.h).
//
// A 1x1 convolution is a [pixels x Cin] x [Cin x Cout] GEMM with a tiny K (16 ... 64): 4 (Cin + Cout) bytes per pixel against
// 2 Cin Cout flops -- HBM-bound on the 16-bit matrix pipe even with three products per multiply.  Nothing is staged: lane (pixel li,
// k half lh) loads its 8 consecutive input channels of one pixel (32 B of the NHWC row) straight from global memory, forms the two
// fp16 planes of x * s_x in registers (the same split conv_split_kernel's staging does, rcf_f16_planes) and feeds the MFMA; both
// planes of the whole weight matrix (<= 8 k-step x co-tile pieces) live in registers for the lifetime of the wave.  No LDS, no
// barrier in the main loop.  The f32-MFMA implicit-GEMM kernel ran these layers at 1.9 - 2.9 TB/s (profiles/r03_pmc_bench.json).
#pragma once

template <int KST_, int NT_>
struct Pw2Cfg {
    // k-steps of 16 input channels, 32-co tiles; two 32-pixel blocks per trip while the weight planes leave room for their accumulators
    static constexpr int KST = KST_, NT = NT_, MT = (KST_ * NT_ <= 4 && NT_ <= 2) ? 2 : 1;
    static_assert(KST_ * NT_ <= 8, "both weight planes must fit the register file next to the accumulators");
};

template <class C>
__global__ void __launch_bounds__(256, 2) conv1x1_f16x2_kernel(ConvArgs a) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int lh = lane >> 5;
    const SplitScales sc = rcf_split_scales(a.amax_a1, nullptr, a.amax_b);
    // weights: [k-step s][plane][co][16] fp16, halves swizzled by (co >> 3) & 1 (pack_weights_split_kernel with KS = 1, BN = 32 NT, NPL = 2)
    u32x4 bw[C::KST][2][C::NT];
    {
        const unsigned char* wp = reinterpret_cast<const unsigned char*>(a.wp);
#pragma unroll
        for (int s = 0; s < C::KST; ++s)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni) {
                    const int co = ni * 32 + li;
                    bw[s][pl][ni] = *reinterpret_cast<const u32x4*>(wp + ((size_t)((s * 2 + pl) * 32 * C::NT + co) * 32) + ((lh ^ ((co >> 3) & 1)) * 16));
                }
    }
    const long long npix = (long long)a.n * a.h_out * a.w_out;
    const long long nblk32 = (npix + 31) / 32;
    const int nwaves = gridDim.x * 4;
    const bool want_stats = a.stats != nullptr;
    double st1[C::NT], st2[C::NT];
#pragma unroll
    for (int ni = 0; ni < C::NT; ++ni) { st1[ni] = 0.0; st2[ni] = 0.0; }

    for (long long blk0 = ((long long)blockIdx.x * 4 + wave) * C::MT; blk0 < nblk32; blk0 += (long long)nwaves * C::MT) {
        // A operands of MT blocks: lane (li, lh) <- channels 16 s + 8 lh .. + 7 of output pixel blk * 32 + li (source pixel through the stride)
        f32x4 raw[C::MT][C::KST][2];
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
            const long long p = (blk0 + mt) * 32 + li;
            const bool ok = p < npix;
            size_t sp = 0;
            if (ok) {
                if (a.stride == 1) sp = (size_t)p;
                else {
                    const int ox = (int)(p % a.w_out);
                    const long long t = p / a.w_out;
                    const int oy = (int)(t % a.h_out);
                    const int im = (int)(t / a.h_out);
                    sp = ((size_t)im * a.h_in + (size_t)oy * a.stride) * a.w_in + (size_t)ox * a.stride;
                }
            }
            const float* row = ok ? a.in1 + sp * a.c1 + lh * 8 : a.zero;   // the zero page: 64 floats, enough for every k-step
#pragma unroll
            for (int s = 0; s < C::KST; ++s) {
                raw[mt][s][0] = *reinterpret_cast<const f32x4*>(row + (ok ? s * 16 : 0));
                raw[mt][s][1] = *reinterpret_cast<const f32x4*>(row + (ok ? s * 16 : 0) + 4);
            }
        }
        f32x16 acc[C::MT][C::NT];
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
#pragma unroll
            for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][ni][r] = 0.f;
#pragma unroll
            for (int s = 0; s < C::KST; ++s) {
                u32x4 a0, a1;   // the two fp16 planes of this lane's 8 channels
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const rcf_f16_pair q = rcf_f16_planes(raw[mt][s][d >> 1][(d & 1) * 2] * sc.sa, raw[mt][s][d >> 1][(d & 1) * 2 + 1] * sc.sa);
                    a0[d] = q.p0;
                    a1[d] = q.p1;
                }
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni) {   // smallest products first (conv_split_kernel's order)
                    acc[mt][ni] = rcf_mfma_split<2>(as_bf16x8(a1), as_bf16x8(bw[s][0][ni]), acc[mt][ni]);
                    acc[mt][ni] = rcf_mfma_split<2>(as_bf16x8(a0), as_bf16x8(bw[s][1][ni]), acc[mt][ni]);
                    acc[mt][ni] = rcf_mfma_split<2>(as_bf16x8(a0), as_bf16x8(bw[s][0][ni]), acc[mt][ni]);
                }
            }
        }
        // epilogue: lane holds channel ni * 32 + li of pixels row(r): for one r the 32 lanes of a half write 128 contiguous bytes
        auto epilogue = [&](auto add_tag) __attribute__((always_inline)) {
            constexpr bool ADD = decltype(add_tag)::value;
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt) {
                const long long pb0 = (blk0 + mt) * 32;
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 4) {
                    float old[4][C::NT];
                    if (ADD) {   // the four pixels' old values in flight together, landed before the stores (gfx9 counts stores in vmcnt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const long long p = pb0 + rcf_mfma_row(r0 + j, lh);
#pragma unroll
                            for (int ni = 0; ni < C::NT; ++ni) {
                                const int co = ni * 32 + li;
                                old[j][ni] = a.out[(p < npix && co < a.c_out) ? (size_t)p * a.c_out + co : 0];
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int ni = 0; ni < C::NT; ++ni) asm volatile("" : "+v"(old[j][ni]));
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const long long p = pb0 + rcf_mfma_row(r0 + j, lh);
#pragma unroll
                        for (int ni = 0; ni < C::NT; ++ni) {
                            const int co = ni * 32 + li;
                            float v = acc[mt][ni][r0 + j] * sc.ia * sc.ib;   // undo the operand scales (exact: powers of two)
                            if (ADD) v += old[j][ni];
                            if (p < npix && co < a.c_out) {
                                a.out[(size_t)p * a.c_out + co] = v;
                                if (want_stats) {   // fp64 per value, like conv_split_kernel
                                    const double dv = (double)v;
                                    st1[ni] += dv;
                                    st2[ni] += dv * dv;
                                }
                            }
                        }
                    }
                }
            }
        };
        if (a.accumulate) epilogue(std::true_type{});
        else epilogue(std::false_type{});
    }
    if (want_stats) {
        __shared__ double red[4 * 32 * C::NT * 2];
#pragma unroll
        for (int ni = 0; ni < C::NT; ++ni) {
            const double t1 = st1[ni] + __shfl_xor(st1[ni], 32);
            const double t2 = st2[ni] + __shfl_xor(st2[ni], 32);
            if (lh == 0) {
                red[((wave * C::NT + ni) * 32 + li) * 2 + 0] = t1;
                red[((wave * C::NT + ni) * 32 + li) * 2 + 1] = t2;
            }
        }
        __syncthreads();
        if (tid < 32 * C::NT && tid < a.c_out) {
            double t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                t1 += red[((w * C::NT + tid / 32) * 32 + (tid & 31)) * 2 + 0];
                t2 += red[((w * C::NT + tid / 32) * 32 + (tid & 31)) * 2 + 1];
            }
            a.stats[((size_t)blockIdx.x * 2 + 0) * a.c_out + tid] = t1;
            a.stats[((size_t)blockIdx.x * 2 + 1) * a.c_out + tid] = t2;
        }
    }
}

// grid of the pointwise kernel = number of BatchNorm partial rows it writes
inline int pw2_grid(long long npix, int kst, int nt) {
    const int per_trip = 32 * 4 * ((kst * nt <= 4 && nt <= 2) ? 2 : 1);   // 4 waves x MT blocks of 32 pixels per workgroup trip
    const long long trips = (npix + per_trip - 1) / per_trip;
    long long g = 2 * (long long)num_cus() * 2;                        // two workgroups per CU, two trips' worth of slack
    if (g > trips) g = trips;
    if (g < 1) g = 1;
    return (int)g;
}

template <class C>
int launch_pw2(const ConvArgs& a, hipStream_t st) {
    const int g = pw2_grid((long long)a.n * a.h_out * a.w_out, C::KST, C::NT);
    hipLaunchKernelGGL((conv1x1_f16x2_kernel<C>), dim3(g), dim3(256), 0, st, a);
    return rcf_launch_status();
}

inline bool pw2_cfg_ok(int kst, int nt) { return kst >= 1 && kst <= 4 && nt >= 1 && nt <= 4 && kst * nt <= 8; }

template <class F>
int dispatch_pw2(int kst, int nt, F&& f) {
#define RCF_PW2(K, N) if (kst == K && nt == N) return f(Pw2Cfg<K, N>{})
    RCF_PW2(1, 1); RCF_PW2(1, 2); RCF_PW2(1, 3); RCF_PW2(1, 4);
    RCF_PW2(2, 1); RCF_PW2(2, 2); RCF_PW2(2, 3); RCF_PW2(2, 4);
    RCF_PW2(3, 1); RCF_PW2(3, 2);
    RCF_PW2(4, 1); RCF_PW2(4, 2);
#undef RCF_PW2
    return RCF_EUNSUPPORTED;
}
