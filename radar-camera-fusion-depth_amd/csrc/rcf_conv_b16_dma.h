// conv_b16_kernel: the 3x3 / 2x2 forward and input-gradient convolution for bf16 NHWC tensors (RCF_STORE_BF16), included by
// rcf_conv_impl.h in the bf16 translation unit only.
//
// With bf16 tensors the operands are already in the MFMA's format, so nothing has to pass through the VALU on its way to LDS:
//  * the halo tile of a 16-channel chunk ([pixel][16 bf16] = two 16-B pieces per pixel, halves XOR-swizzled by bit 3 of the pixel
//    index exactly like conv_split_kernel's planes) arrives by LDS-DMA (global_load_lds_dwordx4) with PER-LANE source addresses --
//    the gather (zero padding through a zero page, nearest upsample, strided phases, the virtual tall image) is address arithmetic
//    done once per tile, the copy itself uses no registers and no ds_write;
//  * the chunk's packed weights ([tap][co][16 bf16], 18 KB for 64 output channels) arrive the same way;
//  * A and B are double-buffered, so ONE barrier per chunk publishes chunk q while the DMA of chunk q + 1 is already in flight
//    behind the 9 x MT x NT MFMAs of chunk q (conv_split_kernel: two barriers per chunk + one per kernel row, because its A tile is
//    produced by the waves themselves);
//  * epilogue: two lanes that hold neighbouring output channels exchange one value (DPP) and each stores ONE dword (two bf16) per
//    pixel pair instead of two 2-byte stores per pixel; BatchNorm statistics are taken from the rounded values, summed in fp32 over
//    the 16 values of an accumulator and in fp64 across tiles.
// Phase timing of conv_split_kernel with bf16 tensors showed why (tools/phase_timing.py, RCF_BENCH_PREC=bf16): MFMA rows 11-14 % of
// the wave time, row prologues 23-36 %, epilogue 11-39 %, barriers 15-25 %.
//
// Everything conv_split_kernel does for the network is kept: channel concat (two sources), nearest-upsample gather, strided
// (phase) gather + phase_sum, strided / offset output (phase convolutions), += accumulation, the fused inference epilogue
// (bias + LeakyReLU + residual), the virtual tall image, and stride 2 (LSTEP = 2).  Not kept: BatchNorm-on-load (a DMA cannot
// transform), channel counts that are not multiples of 16 (those layers stay on conv_split_kernel).
#pragma once

// P4 (KS = 2 only): the four OUTPUT phases of an up-2x forward (rcf_conv_desc.phase_sum == 2) from ONE staged tile.  Phase (a, b)
// is a 2x2 convolution of x with pad (1 - a, 1 - b), so the four phases of an output tile read the SAME 3x3-halo tile of x: it is
// staged once per channel chunk (halo geometry GK = 3) together with the 4 x 4 weight blocks of the chunk, the 16 (phase, tap)
// products run from it into four accumulator sets, and the epilogue runs once per phase.  (Round 4's one-launch form ran the four
// 2x2 convolutions back to back, each re-staging its own tile: the re-reads were meant to hit L2 but the PMC counters show x
// fetched ~4.5 x per launch at 450 x 800, profiles/r05_pmc_bf16_infer.json -- the kernel was HBM-bound on its own re-reads.)
// PM_ = 2: the four OUTPUT phases of a 3x3 stride-2 convolution's input gradient (phase_sum == 3) the same way.  Phase (a, b) writes
// dx(2y + a, 2x + b) and is a 2x2 convolution of dz with pad 0 whose taps (ty, tx) exist only for ty <= a, tx <= b (1 + 2 + 2 + 4 = 9
// of the 16: the per-phase launches multiply the other seven by zero weights): the tile of dz (halo of a 2x2) is staged once and the
// nine real (phase, tap) products run from it -- 9/16 of the matrix work and a quarter of the loads of the four launches.
template <int KS_, int NT_, int PX_, int MT_, int LSTEP_ = 1, int PM_ = 0>
struct DmaCfg {
    static constexpr int PM = PM_;          // 0: one convolution; 1: up-2x forward phases; 2: stride-2 input-gradient phases
    static constexpr bool P4 = PM_ != 0;
    static constexpr int KS = KS_, GK = PM_ == 1 ? 3 : KS_, T = P4 ? 16 : KS_ * KS_, NPH = P4 ? 4 : 1, LSTEP = LSTEP_;
    static constexpr int NSTEP = PM_ == 2 ? 9 : 16;   // (phase, tap) products per chunk
    static_assert(!P4 || (KS_ == 2 && LSTEP_ == 1 && PX_ == 32), "phase merging: 2x2 taps, stride 1, 32-pixel tile rows");
    static constexpr int NT = NT_, BN = 32 * NT_;
    static constexpr int PX = PX_, PY = 32 / PX_, MT = MT_, NW = 4, TH = PY * MT * NW;
    static constexpr int HXP = (PX - 1) * LSTEP + GK, HYP = (TH - 1) * LSTEP + GK, NPIX = HXP * HYP;
    static constexpr int NA = (2 * NPIX + 255) / 256;        // LDS-DMA instructions per thread and A tile (one 16-B piece per lane)
    static constexpr int A_BYTES = NA * 256 * 16;            // every lane of every instruction lands somewhere
    // A-tile layout in LDS.  HP (32-pixel tile rows, stride 1): [halo row][channel half][halo x][16 B].  The 32 lanes of an MFMA row
    // block read 32 CONSECUTIVE pixels of one halo row and one half, 16 B apart: conflict-free as it stands (16 lanes x 16 B = all 64
    // banks), and the address of tap (ky, kx) is the lane's base + a compile-time immediate -- no per-read address arithmetic, one
    // address register per MFMA row block.  A DMA instruction (64 consecutive 16-B slots) still fetches both halves of ~32 pixels, so
    // it touches as many cache lines as in the pixel-major layout ([half][pixel] would touch twice as many: measured 16 % slower).
    // Otherwise (16-pixel rows: two tile rows per MFMA row block; stride 2): [pixel][2 x 16 B] with the halves XOR-swizzled by bit 3 of
    // the pixel index, address recomputed per read.
    static constexpr bool HP = (PX_ == 32 && LSTEP_ == 1);
    static constexpr int B_BYTES = T * BN * 32;              // one chunk of packed weights
    static constexpr int NKB = B_BYTES / 1024;
    static constexpr int LDS_BYTES = 2 * (A_BYTES + B_BYTES);
    static_assert(B_BYTES % 1024 == 0, "weight chunk must be whole KiB");
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

// P4: the 16 (phase, tap) products of a chunk ordered by the halo position (ky, kx) they read, so that an A operand is fetched
// from LDS once per position (9 x MT reads per chunk instead of 16 x MT) -- phase (pa, pb), tap (ty, tx) reads (pa + ty, pb + tx)
struct P4Tab { int pos[16]; int t16[16]; };
// stride-2 input gradient: position = tap (ty, tx) of the 2x2 halo, shared by the phases it exists for (ty <= a, tx <= b)
constexpr P4Tab rcf_p4_tab_s2() {
    P4Tab t{};
    int n = 0;
    for (int ty = 0; ty < 2; ++ty)
        for (int tx = 0; tx < 2; ++tx)
            for (int pa = 0; pa < 2; ++pa)
                for (int pb = 0; pb < 2; ++pb) {
                    if (ty > pa || tx > pb) continue;
                    t.pos[n] = ty * 2 + tx;
                    t.t16[n] = (pa * 2 + pb) * 4 + ty * 2 + tx;
                    ++n;
                }
    return t;
}
constexpr P4Tab rcf_p4_tab() {
    P4Tab t{};
    int n = 0;
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx)
            for (int pa = 0; pa < 2; ++pa)
                for (int pb = 0; pb < 2; ++pb) {
                    const int ty = ky - pa, tx = kx - pb;
                    if (ty < 0 || ty > 1 || tx < 0 || tx > 1) continue;
                    t.pos[n] = ky * 3 + kx;
                    t.t16[n] = (pa * 2 + pb) * 4 + ty * 2 + tx;
                    ++n;
                }
    return t;
}

template <int CTRL>
__device__ __forceinline__ unsigned rcf_dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// BST (an input-gradient launch that is the only writer of dY of a BatchNorm + LeakyReLU block, ConvArgs.bz / bk): instead of the
// BatchNorm statistics of its own outputs the kernel leaves that block's backward sums, sum g and sum g * xhat, in `stats`
// (conv_split_kernel<..., BST> does the same for fp32 tensors; rcf_conv2d_dgrad_bn_sums).
template <class C, bool EPI, bool BST = false>
__global__ void __launch_bounds__(256, 2) conv_b16_kernel(ConvArgs a) {
    static_assert(!(EPI && BST), "the inference epilogue and the BatchNorm-backward sums exclude each other");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    // [A buf 0][A buf 1][B buf 0][B buf 1]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int li = lane & 31;
    const int lh = lane >> 5;

    int apix[C::MT];   // halo pixel of this lane's output pixel at tap (0, 0)
#pragma unroll
    for (int mi = 0; mi < C::MT; ++mi) {
        const int tr = (wave * C::MT + mi) * C::PY + li / C::PX;
        const int tc = li % C::PX;
        apix[mi] = tr * C::LSTEP * C::HXP + tc * C::LSTEP;
        if (C::HP) apix[mi] = ((tr * 2 + lh) * C::HXP + tc) * 16;   // HP layout: the byte offset of tap (0, 0); taps add immediates
    }
    // byte offset inside the A tile of this lane's operand for MFMA row block mi at tap (ky, kx)
    auto a_off = [&](int mi, int ky, int kx) __attribute__((always_inline)) -> int {
        if constexpr (C::HP) return apix[mi] + (ky * 2 * C::HXP + kx) * 16;
        else {
            const int p = apix[mi] + ky * C::HXP + kx;
            return p * 32 + ((lh ^ ((p >> 3) & 1)) * 16);
        }
    };
    const int bbase = li * 32 + ((lh ^ ((li >> 3) & 1)) * 16);

    f32x16 acc[C::NPH][C::MT][C::NT];
    const int nchunk = a.nchunk1 + a.nchunk2;
    // (P4: the packed weights keep the per-phase layout [phase][n-tile][chunk][4 taps][co][16]; a chunk's DMA collects its four blocks)
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(a.wp) + (size_t)blockIdx.y * nchunk * (C::B_BYTES / C::NPH);
    const int n0 = blockIdx.y * C::BN;
    const int psum = C::P4 ? 0 : a.phase_sum;   // P4: one item per chunk, no phase loop
    const int nitem = psum ? 4 * nchunk : nchunk;

    // A staging: piece s = i * 256 + tid of the tile -> halo pixel s >> 1, LDS half s & 1, source half (s & 1) ^ ((pixel >> 3) & 1)
    // byte offset of the piece's 16 B at channel chunk 0, from the start of the tile's first image in the source (a buffer descriptor
    // built per tile and source, so 32 bits reach); 0xffffffff for padding / outside pieces: the buffer unit answers those with ZEROS,
    // which land in LDS like data -- no zero page, no select, and the per-chunk address of a piece is this register + an SGPR
    unsigned pix[C::NA];
    int fimg = 0;        // that first image (wave-uniform)
#ifdef RCF_B16_DIAG
    int diag_issued = 0;
#endif
    auto setup = [&](int tile, bool first, int ph) {
        int t = tile;
        const int tx = t % a.tiles_x;
        t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        const int img = t / a.tiles_y;
        // phase_sum 1: the four INPUT phases of an up-2x input gradient summed; 2: the four OUTPUT phases of an up-2x forward, one after
        // the other on the same tile (pad 1 - a, 1 - b; the outputs go to (2y + a, 2x + b))
        const int pa = C::P4 ? (C::PM == 1 ? 1 : 0) : (psum == 2 ? 1 - (ph >> 1) : (psum ? (ph >> 1) : a.pad)), pb = C::P4 ? (C::PM == 1 ? 1 : 0) : (psum == 2 ? 1 - (ph & 1) : (psum ? (ph & 1) : a.pad_x));
        const int ioy = psum ? (ph >> 1) : a.ioy, iox = psum ? (ph & 1) : a.iox;
        const int iy0 = ty * C::TH * C::LSTEP - pa;
        const int ix0 = tx * C::PX * C::LSTEP - pb;
        const int hs = first ? a.h1 : a.h_in, ws = first ? a.w1 : a.w_in;
        const int gmode = first ? a.gather1 : RCF_GATHER_DIRECT;
        const unsigned pixb = (unsigned)(first ? a.c1 : a.c2) * 2u;   // bytes per source pixel
        fimg = img;
        if (a.vt) {   // the image of the tile's first real halo row
            fimg = (int)(((float)(iy0 < 0 ? 0 : iy0) + 0.5f) * a.inv_hp);
            fimg = fimg < a.nimg ? fimg : a.nimg - 1;
        }
        fimg = __builtin_amdgcn_readfirstlane(fimg);
#pragma unroll
        for (int i = 0; i < C::NA; ++i) {
            const int sp = i * 256 + tid;
            int p = sp >> 1, hy, hx;
            if constexpr (C::HP) {   // slot sp = (hy, half, hx)
                hy = sp / (2 * C::HXP);
                const int rem = sp - hy * (2 * C::HXP);
                hx = rem >= C::HXP ? rem - C::HXP : rem;
                p = hy < C::HYP ? hy * C::HXP + hx : C::NPIX;
            } else {
                hy = p / C::HXP;
                hx = p - hy * C::HXP;
            }
            const int ly = iy0 + hy, lx = ix0 + hx;
            int v = -1;
            if (p < C::NPIX && ly >= 0 && lx >= 0 && lx < a.w_in) {
                if (a.vt) {
                    const int im = (int)(((float)ly + 0.5f) * a.inv_hp);
                    const int y = ly - im * a.hp;
                    if (im < a.nimg && y < a.h_in) v = ((im - fimg) * hs + y) * ws + lx;
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
                    if (ok) v = py * ws + px;
                }
            }
            const int hh = C::HP ? ((sp % (2 * C::HXP)) >= C::HXP ? 1 : 0) : ((sp & 1) ^ ((p >> 3) & 1));   // channel half of the piece
            pix[i] = v >= 0 ? (unsigned)v * pixb + (unsigned)hh * 16u : 0xffffffffu;
        }
    };
    // item -> (phase, chunk); issue the DMA of one item's A tile and weight chunk into buffer `buf`
    auto issue = [&](int tile, int item, int buf) {
        const int ph = psum ? item / nchunk : 0;
        const int q = psum ? item - ph * nchunk : item;
        const bool first = q < a.nchunk1;
        if (q == 0 || q == a.nchunk1) setup(tile, first, ph);
        const unsigned char* src = reinterpret_cast<const unsigned char*>(first ? a.in1 : a.in2);
        const int csrc = first ? a.c1 : a.c2;
        const int hs = first ? a.h1 : a.h_in, ws = first ? a.w1 : a.w_in;
        const unsigned cbb = (unsigned)((first ? q : q - a.nchunk1) * 32);   // this chunk's 16 channels: byte offset inside a pixel
        const __amdgpu_buffer_rsrc_t rsa = rcf_rsrc(src + (size_t)fimg * hs * ws * csrc * 2);
        unsigned char* Ab = smem_b + buf * C::A_BYTES;
#pragma unroll
        for (int i = 0; i < C::NA; ++i)
            rcf_buffer_to_lds16(rsa, Ab + (i * 256 + wave_u * 64) * 16, pix[i], cbb);
        unsigned char* Bb = smem_b + 2 * C::A_BYTES + buf * C::B_BYTES;
#ifdef RCF_B16_DIAG   // diagnostics build (WRONG results): env RCF_B16_DIAG=1 -> the weight chunks are fetched for the first two items only
        if (a.xcd_band >= 64 && diag_issued >= 2) return;
        ++diag_issued;
#endif
        if constexpr (C::P4) {   // the chunk's block of each of the four phases, [phase][tap][co][16] in LDS
            constexpr int KPP = C::NKB / 4;   // KiB per phase
            static_assert(C::NKB % 4 == 0, "whole KiB per phase and wave");
#pragma unroll
            for (int i = 0; i < C::NKB / 4; ++i) {
                const int kb = i * 4 + wave_u;
                const int wph = kb / KPP, k = kb - wph * KPP;
                const unsigned char* src = wp + (size_t)wph * a.wp_phase_stride * 4 + (size_t)q * (C::B_BYTES / 4) + k * 1024 + lane * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(Bb + kb * 1024), 16, 0, 0);
            }
            return;
        }
        const unsigned char* wsrc = wp + (size_t)ph * a.wp_phase_stride * 4 + (size_t)q * C::B_BYTES + lane * 16;
#pragma unroll
        for (int i = 0; i < (C::NKB + 3) / 4; ++i) {
            int kb = i * 4 + wave_u;
            if ((i + 1) * 4 > C::NKB) kb = kb < C::NKB ? kb : C::NKB - 1;   // ragged tail: a duplicate copy of the last KiB is harmless
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + kb * 1024),
                                             (__attribute__((address_space(3))) void*)(Bb + kb * 1024), 16, 0, 0);
        }
    };

    // BatchNorm statistics: per lane two channels (its own and its pair partner's, see the epilogue), fp64 across tiles
    double st1[C::NT][2], st2[C::NT][2];
#pragma unroll
    for (int ni = 0; ni < C::NT; ++ni) { st1[ni][0] = st1[ni][1] = 0.0; st2[ni][0] = st2[ni][1] = 0.0; }

    TileWalk walk;
    int tile = walk.first(a);
    int q = 0;
    int buf = 0;
#ifdef RCF_PHASE_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    if (tile < a.ntiles) issue(tile, 0, 0);
    while (tile < a.ntiles) {
        int ntile = tile, nq = q + 1;
        if (nq == nitem) { nq = 0; ntile = walk.next(tile, a); }
        const bool more = ntile < a.ntiles;
        RCF_T(t_w0);
        rcf_wait_dma();       // this item's pieces issued by this wave have landed ...
        __syncthreads();      // ... and everybody's; everybody is also done reading the other buffer (previous item)
        RCF_T(t_w1);
        RCF_TACC(0, t_w1, t_w0);   // 0: DMA wait + barrier
        if (more) issue(ntile, nq, buf ^ 1);
        RCF_T(t_w2);
        RCF_TACC(1, t_w2, t_w1);   // 1: address arithmetic + DMA issue of the next item
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
        if constexpr (C::P4) {
            const unsigned char* Ab = smem_b + buf * C::A_BYTES;
            const unsigned char* Bb = smem_b + 2 * C::A_BYTES + buf * C::B_BYTES;
            constexpr P4Tab TB = C::PM == 2 ? rcf_p4_tab_s2() : rcf_p4_tab();
            constexpr int PW = C::PM == 2 ? 2 : 3;   // positions per halo row
            constexpr int MN = C::MT * C::NT;
            bf16x8 av[2][C::MT], bv[2][C::NT];
#pragma unroll
            for (int mi = 0; mi < C::MT; ++mi) av[0][mi] = as_bf16x8(*reinterpret_cast<const u32x4*>(Ab + a_off(mi, 0, 0)));
#pragma unroll
            for (int ni = 0; ni < C::NT; ++ni) bv[0][ni] = as_bf16x8(*reinterpret_cast<const u32x4*>(Bb + (TB.t16[0] * C::BN + ni * 32) * 32 + bbase));
            __builtin_amdgcn_sched_barrier(0);
            int as = 0;
#pragma unroll
            for (int s = 0; s < C::NSTEP; ++s) {
                const int bs = s & 1;
                const bool has_next = s + 1 < C::NSTEP;
                const int npos = TB.pos[has_next ? s + 1 : s], nt16 = TB.t16[has_next ? s + 1 : s];
                const bool new_a = has_next && npos != TB.pos[s];
                const int nas = new_a ? as ^ 1 : as;
                const int pi = TB.t16[s] >> 2;
                const int NRD = has_next ? C::NT + (new_a ? C::MT : 0) : 0;
                int nr = 0;
#pragma unroll
                for (int j = 0; j < MN; ++j) {
                    const int mi = j / C::NT, ni = j % C::NT;
                    acc[pi][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[as][mi], bv[bs][ni], acc[pi][mi][ni], 0, 0, 0);
#pragma unroll
                    for (int rep = 0; rep < 3; ++rep) {
                        if (nr < NRD && (MN == 1 || nr * (MN - 1) < (j + 1) * NRD)) {
                            __builtin_amdgcn_sched_barrier(0);
                            // read order: the weights of the next product first (always needed), then the A tiles of a new position
                            if (nr < C::NT) {
                                bv[bs ^ 1][nr] = as_bf16x8(*reinterpret_cast<const u32x4*>(Bb + (nt16 * C::BN + nr * 32) * 32 + bbase));
                            } else {
                                const int rmi = nr - C::NT;
                                av[nas][rmi] = as_bf16x8(*reinterpret_cast<const u32x4*>(Ab + a_off(rmi, npos / PW, npos % PW)));
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            ++nr;
                        }
                    }
                }
                as = nas;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            const unsigned char* Ab = smem_b + buf * C::A_BYTES;
            const unsigned char* Bb = smem_b + 2 * C::A_BYTES + buf * C::B_BYTES;
            bf16x8 av[2][C::MT], bv[2][C::NT];
            auto fetch = [&](int tap, int slot) {
                const int ky = tap / C::KS, kx = tap % C::KS;
#pragma unroll
                for (int mi = 0; mi < C::MT; ++mi) {
                    av[slot][mi] = as_bf16x8(*reinterpret_cast<const u32x4*>(Ab + a_off(mi, ky, kx)));
                }
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni)
                    bv[slot][ni] = as_bf16x8(*reinterpret_cast<const u32x4*>(Bb + (tap * C::BN + ni * 32) * 32 + bbase));
            };
            fetch(0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < C::T; ++tap) {
                const int cur = tap & 1, nxt = cur ^ 1;
                // The MT x NT MFMAs of this tap with the next tap's MT + NT operand reads issued ONE AT A TIME between them, in the
                // order the next tap's MFMAs will want them (a block of reads stalls the wave's MFMA issue for as long as the LDS
                // queue takes them); sched_barrier pins the hand-written order -- left alone, hipcc sinks the reads behind the MFMAs
                // and puts s_waitcnt lgkmcnt(0) in front of every second MFMA.
                constexpr int MN = C::MT * C::NT, NRD = C::MT + C::NT;
                const bool has_next = tap + 1 < C::T;
                const int nky = (tap + 1) / C::KS, nkx = (tap + 1) % C::KS;
                int nr = 0;
#pragma unroll
                for (int j = 0; j < MN; ++j) {
                    const int mi = j / C::NT, ni = j % C::NT;
                    acc[0][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[cur][mi], bv[cur][ni], acc[0][mi][ni], 0, 0, 0);
                    if (has_next) {
#pragma unroll
                        for (int rep = 0; rep < 3; ++rep) {
                            if (nr < NRD && nr * (MN - 1) < (j + 1) * NRD) {   // front-loaded: the last read leaves before the last MFMA
                                __builtin_amdgcn_sched_barrier(0);
                                // read order: A tile 0, all B tiles, then the remaining A tiles
                                if (nr == 0 || nr > C::NT) {
                                    const int rmi = nr == 0 ? 0 : nr - C::NT;
                                    av[nxt][rmi] = as_bf16x8(*reinterpret_cast<const u32x4*>(Ab + a_off(rmi, nky, nkx)));
                                } else {
                                    const int rni = nr - 1;
                                    bv[nxt][rni] = as_bf16x8(*reinterpret_cast<const u32x4*>(Bb + ((tap + 1) * C::BN + rni * 32) * 32 + bbase));
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                ++nr;
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        RCF_T(t_w3);
        RCF_TACC(2, t_w3, t_w2);   // 2: accumulator init + MFMAs + LDS reads
        if (phase_out ? (q - oph * nchunk == nchunk - 1) : (q == nitem - 1)) {
          // P4: the four phases' outputs of the tile.  Phases (a, 0) and (a, 1) hold the two halves of the same output row's pixel
          // pairs (2x, 2x + 1), so they are stored TOGETHER: lanes li (even) and li + 1 exchange one value per accumulator row -- the
          // even lane stores channels (co, co + 1) of phase (a, 0)'s pixel, the odd lane those of phase (a, 1)'s -- and the 32 lanes
          // of a row block write 128 contiguous bytes (with 32 output channels), like a stride-1 layer's store.  (Phase by phase every
          // store wrote 64-byte halves of four lines: 42 % of the wave time sat in the epilogue, tools/up2x_bench.py.)
          auto p4_epilogue = [&]() __attribute__((always_inline)) {
            if constexpr (C::PM == 1) {
            int t = tile;
            const int tx = t % a.tiles_x;
            t /= a.tiles_x;
            const int ty = t % a.tiles_y;
            const int img = t / a.tiles_y;
            const int oy0 = ty * C::TH, ox0 = tx * C::PX;
            const bool want_stats = !EPI && a.stats != nullptr;
            const int odd = li & 1;
            float eb[C::NT][2];
            if (EPI) {
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni) {
                    const int cp = (n0 + ni * 32 + li) & ~1;
                    eb[ni][0] = a.bias[cp < a.c_out ? cp : 0];
                    eb[ni][1] = a.bias[cp + 1 < a.c_out ? cp + 1 : 0];
                }
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni) asm volatile("" : "+v"(eb[ni][0]), "+v"(eb[ni][1]));
            }
            int fim = img;
            if (a.vt) {
                fim = (int)(((float)oy0 + 0.5f) * a.inv_hp);
                fim = fim < a.nimg ? fim : a.nimg - 1;
            }
            fim = __builtin_amdgcn_readfirstlane(fim);
            const size_t img_b = (size_t)a.ohp * a.owp * a.c_out * 2;           // bytes per image
            const unsigned pixb = (unsigned)a.c_out * 2u;                       // bytes per output pixel
            const unsigned rowb = (unsigned)a.owp * pixb;                       // bytes per output row
            const unsigned pstep = 2u * pixb;                                   // bytes between the pixel pairs of neighbouring x
            unsigned char* outb = reinterpret_cast<unsigned char*>(a.out);
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(outb + (size_t)fim * img_b, 0, 0x7fffffff, 0x00020000);
            const int wlim = a.w_out - ox0;   // (the physical width is 2 w_out: valid_desc)
            const unsigned xoff = (unsigned)(2 * ox0) * pixb + (unsigned)n0 * 2u;
            // per lane: byte offset of its dword inside a row block (column 4 lh of x, pixel 2x + odd, channel pair li & ~1)
            unsigned l0 = (unsigned)(4 * lh) * pstep + (unsigned)odd * pixb + (unsigned)(li & ~1) * 2u;
            int cl[C::NT];
#pragma unroll
            for (int ni = 0; ni < C::NT; ++ni) cl[ni] = (((n0 + ni * 32 + li) & ~1) < a.c_out) ? wlim - 4 * lh : 0;
            asm volatile("" : "+v"(l0));
            auto pair_rows = [&](auto a_tag, auto stats_tag) __attribute__((always_inline)) {
                constexpr int A = decltype(a_tag)::value;
                constexpr bool STATS = decltype(stats_tag)::value;
                constexpr int RPB = 16 / C::PY;   // accumulator rows per tile row
#pragma unroll
                for (int mi = 0; mi < C::MT; ++mi) {
                    float s1[C::NT][2], s2[C::NT][2];
#pragma unroll
                    for (int ni = 0; ni < C::NT; ++ni) { s1[ni][0] = s1[ni][1] = 0.f; s2[ni][0] = s2[ni][1] = 0.f; }
#pragma unroll
                    for (int rq = 0; rq < C::PY; ++rq) {
                        int oy = oy0 + (wave_u * C::MT + mi) * C::PY + rq;
                        int im = img;
                        bool rok = true;
                        if (a.vt) {
                            im = (int)(((float)oy + 0.5f) * a.inv_hp);
                            oy -= im * a.hp;
                            rok = im < a.nimg;
                        }
                        const int py = 2 * oy + A;
                        rok = rok && oy < a.h_out && py < a.ohp;
                        const unsigned rowoff = __builtin_amdgcn_readfirstlane((unsigned)((im - fim) * a.ohp + py) * rowb + xoff);
                        if (__builtin_amdgcn_readfirstlane((int)rok)) {
#pragma unroll
                            for (int ni = 0; ni < C::NT; ++ni) {
#pragma unroll
                                for (int k = 0; k < RPB; ++k) {
                                    const int r = rq * RPB + k;                       // accumulator row: x column 8 (r >> 2) + (r & 3) + 4 lh
                                    const int cc = (8 * (r >> 2)) % C::PX + (r & 3);
                                    const unsigned so = rowoff + (unsigned)cc * pstep;
                                    const bool ok = cc < cl[ni];
                                    unsigned lv = l0;
                                    asm volatile("" : "+v"(lv));   // (or hipcc keeps the 16 x NT offsets of a row block live across the whole epilogue)
                                    const unsigned vo = ok ? lv + (unsigned)(ni * 64) : 0xffffffffu;
                                    const float b0 = acc[2 * A][mi][ni][r], b1 = acc[2 * A + 1][mi][ni][r];
                                    const float mine = odd ? b1 : b0;                                 // my channel at my pixel (2x + odd)
                                    const float give = odd ? b0 : b1;                                 // my channel at the partner's pixel
                                    const float got = __uint_as_float(rcf_dpp_u32<0xB1>(__float_as_uint(give)));   // partner's channel at my pixel
                                    float lo = odd ? got : mine, hi = odd ? mine : got;               // channels cp, cp + 1
                                    if constexpr (EPI) {
                                        lo = rcf_lrelu(lo + eb[ni][0]);
                                        hi = rcf_lrelu(hi + eb[ni][1]);
                                    }
                                    const unsigned pk = rcf_f2b2(lo, hi);
                                    __builtin_amdgcn_raw_buffer_store_b32(pk, rs_out, vo, so, 0);
                                    if constexpr (STATS) {   // of the values the tensor holds
                                        const float rlo = ok ? __uint_as_float(pk << 16) : 0.f, rhi = ok ? __uint_as_float(pk & 0xffff0000u) : 0.f;
                                        s1[ni][0] += rlo; s2[ni][0] += rlo * rlo;
                                        s1[ni][1] += rhi; s2[ni][1] += rhi * rhi;
                                    }
                                }
                            }
                        }
                    }
                    if constexpr (STATS) {
#pragma unroll
                        for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
                            for (int e = 0; e < 2; ++e) { st1[ni][e] += (double)s1[ni][e]; st2[ni][e] += (double)s2[ni][e]; }
                    }
                }
            };
            if constexpr (EPI || BST) {
                pair_rows(std::integral_constant<int, 0>{}, std::false_type{});
                pair_rows(std::integral_constant<int, 1>{}, std::false_type{});
            } else if (want_stats) {
                pair_rows(std::integral_constant<int, 0>{}, std::true_type{});
                pair_rows(std::integral_constant<int, 1>{}, std::true_type{});
            } else {
                pair_rows(std::integral_constant<int, 0>{}, std::false_type{});
                pair_rows(std::integral_constant<int, 1>{}, std::false_type{});
            }
            }
          };
          // one output tile of accumulator set PI at output offset (e_ooy, e_oox)
          auto tile_epilogue = [&](auto ph_tag, const int e_ooy, const int e_oox) __attribute__((always_inline)) {
            constexpr int PI = decltype(ph_tag)::value;
            // ---- epilogue.  Lane (li, lh) holds channel co = n0 + ni * 32 + li of 16 pixels per accumulator.  Lanes li (even) and
            // li + 1 exchange one value per pixel pair: the even lane stores channels (co, co + 1) of the pair's first pixel, the odd
            // lane those of the second pixel -- one dword (two bf16) per lane and pixel pair.
            int t = tile;
            const int tx = t % a.tiles_x;
            t /= a.tiles_x;
            const int ty = t % a.tiles_y;
            const int img = t / a.tiles_y;
            const int oy0 = ty * C::TH, ox0 = tx * C::PX;
            const bool want_stats = !EPI && !BST && a.stats != nullptr;
            const int odd = li & 1;
            float bk[BST ? 2 : 1][C::NT][2];   // BST: scale, shift of channels cp, cp + 1
            if constexpr (BST) {
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni) {
                    const int cp = (n0 + ni * 32 + li) & ~1;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        bk[e][ni][0] = a.bk[e * a.c_out + (cp < a.c_out ? cp : 0)];
                        bk[e][ni][1] = a.bk[e * a.c_out + (cp + 1 < a.c_out ? cp + 1 : 0)];
                    }
                }
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
                    for (int e = 0; e < 2; ++e) asm volatile("" : "+v"(bk[e][ni][0]), "+v"(bk[e][ni][1]));
            }
            const bool do_add = EPI ? a.res != nullptr : a.accumulate != 0;
            float eb[C::NT][2];
            if (EPI) {
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni) {
                    const int cp = (n0 + ni * 32 + li) & ~1;
                    eb[ni][0] = a.bias[cp < a.c_out ? cp : 0];
                    eb[ni][1] = a.bias[cp + 1 < a.c_out ? cp + 1 : 0];
                }
                // land the bias loads here: used first inside the masked store blocks, they put an s_waitcnt vmcnt(0) -- which also
                // waits for the previous store -- in front of every store of the inference epilogue
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni) asm volatile("" : "+v"(eb[ni][0]), "+v"(eb[ni][1]));
            }
            // ---- ONE branch-free path for every tile (interior, image edge, virtual-tall separator rows, strided phase outputs, += /
            // residual, the BatchNorm-backward sums).  Address = buffer descriptor of the tile's first image (SGPRs) + a wave-uniform
            // row / pixel-pair / n-tile offset (an SGPR: the instruction's soffset) + ONE per-lane register; a lane whose pixel or
            // channel pair lies outside gets the offset 0xffffffff, which the buffer unit drops (stores) or answers with 0 (loads).
            // (Round 4's epilogue computed 64-bit addresses and bounds per store behind an exec-mask branch each: ~20 instructions per
            // store, 50-56 % of a wave's time on the 64- and 32-channel layers at 225 x 400 and above, tools/phase_timing.py.)
            // ADD (accumulate into out / add the residual) and the statistics are COMPILE-TIME variants: the plain path contains no
            // load at all -- gfx9 counts stores in vmcnt, so one load in the loop makes hipcc wait for every previous store.
            int fim = img;
            if (a.vt) {
                fim = (int)(((float)oy0 + 0.5f) * a.inv_hp);
                fim = fim < a.nimg ? fim : a.nimg - 1;
            }
            fim = __builtin_amdgcn_readfirstlane(fim);
            const size_t img_b = (size_t)a.ohp * a.owp * a.c_out * 2;           // bytes per image
            const unsigned rowb = (unsigned)a.owp * (unsigned)a.c_out * 2u;     // bytes per physical output row
            const unsigned pstep = (unsigned)a.os * (unsigned)a.c_out * 2u;     // bytes between neighbouring output pixels
            unsigned char* outb = reinterpret_cast<unsigned char*>(a.out);
            const unsigned char* addb = reinterpret_cast<const unsigned char*>(EPI ? (a.res != nullptr ? a.res : a.out) : a.out);
            const unsigned char* zb = reinterpret_cast<const unsigned char*>(BST ? a.bz : a.out);
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(outb + (size_t)fim * img_b, 0, 0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(addb) + (size_t)fim * img_b, 0, 0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(zb) + (size_t)fim * img_b, 0, 0x7fffffff, 0x00020000);
            int wlim = a.w_out - ox0;   // columns of this tile that exist: logical (< w_out) and physical (px = (ox0 + c) * os + oox < owp)
            {
                const int wphys = (a.owp - e_oox - ox0 * a.os + a.os - 1) / a.os;
                wlim = wlim < wphys ? wlim : wphys;
            }
            const unsigned xoff = (unsigned)(ox0 * a.os + e_oox) * (unsigned)a.c_out * 2u + (unsigned)n0 * 2u;
            // per lane: byte offset of its dword inside a pixel row block (column 4 lh + odd, channel pair li & ~1), and per n-tile the
            // number of valid columns left of it (0 for a channel pair outside the tensor): pixel pair g of the lane sits at column
            // cc + 4 lh + odd with cc = (8 (g >> 1)) % PX + 2 (g & 1), valid iff cc < cl[ni]
            unsigned l0 = (unsigned)(4 * lh + odd) * pstep + (unsigned)(li & ~1) * 2u;
            int cl[C::NT];
#pragma unroll
            for (int ni = 0; ni < C::NT; ++ni) cl[ni] = (((n0 + ni * 32 + li) & ~1) < a.c_out) ? wlim - (4 * lh + odd) : 0;
            asm volatile("" : "+v"(l0));   // opaque per tile: visible, hipcc hoists the tile-invariant offset pieces out of the tile loop and spills
            auto epilogue = [&](auto add_tag, auto stats_tag) __attribute__((always_inline)) {
                constexpr bool ADD = decltype(add_tag)::value, STATS = decltype(stats_tag)::value;
                constexpr int GPB = 8 / C::PY;   // pixel pairs of an accumulator per tile row: 8 (32-pixel rows) or 4
#pragma unroll
                for (int mi = 0; mi < C::MT; ++mi) {
                    float s1[C::NT][2], s2[C::NT][2];
#pragma unroll
                    for (int ni = 0; ni < C::NT; ++ni) { s1[ni][0] = s1[ni][1] = 0.f; s2[ni][0] = s2[ni][1] = 0.f; }
#pragma unroll
                    for (int rq = 0; rq < C::PY; ++rq) {
                        // the tile row of this group: wave-uniform image, row, validity and byte offset from the descriptor's base
                        int oy = oy0 + (wave_u * C::MT + mi) * C::PY + rq;
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
                            for (int ni = 0; ni < C::NT; ++ni) {
                                unsigned zw[BST ? GPB : 1], oldw[ADD ? GPB : 1];
                                if constexpr (BST || ADD) {
#pragma unroll
                                    for (int k = 0; k < GPB; ++k) {
                                        const int g = rq * GPB + k;
                                        const int cc = (8 * (g >> 1)) % C::PX + 2 * (g & 1);
                                        const unsigned vo = cc < cl[ni] ? l0 + (unsigned)(ni * 64) : 0xffffffffu;
                                        const unsigned so = rowoff + (unsigned)cc * pstep;
                                        if constexpr (BST) zw[k] = __builtin_amdgcn_raw_buffer_load_b32(rs_z, vo, so, 0);
                                        if constexpr (ADD) oldw[k] = __builtin_amdgcn_raw_buffer_load_b32(rs_add, vo, so, 0);
                                    }
                                }
#pragma unroll
                                for (int k = 0; k < GPB; ++k) {
                                    const int g = rq * GPB + k;
                                    const int rj = 2 * g;   // accumulator rows rj, rj + 1: two consecutive pixels of one tile row
                                    const int cc = (8 * (g >> 1)) % C::PX + 2 * (g & 1);
                                    const unsigned so = rowoff + (unsigned)cc * pstep;
                                    unsigned lv = l0;
                                    if constexpr (BST || ADD) asm volatile("" : "+v"(lv));   // (or the compiler keeps the offsets of the loads)
                                    const bool ok = cc < cl[ni];
                                    const unsigned vo = ok ? lv + (unsigned)(ni * 64) : 0xffffffffu;
                                    // (static register indices + a select: indexing the accumulator with the lane-dependent `odd` makes
                                    // hipcc walk all 16 registers with compare/select pairs -- 60 VALU instructions per value)
                                    const float a0 = acc[PI][mi][ni][rj], a1 = acc[PI][mi][ni][rj + 1];
                                    const float mine = odd ? a1 : a0;                                 // my channel at my pixel
                                    const float give = odd ? a0 : a1;                                 // my channel at the partner's pixel
                                    const float got = __uint_as_float(rcf_dpp_u32<0xB1>(__float_as_uint(give)));   // partner's channel at my pixel
                                    float lo = odd ? got : mine, hi = odd ? mine : got;               // channels cp, cp + 1
                                    float alo = 0.f, ahi = 0.f;
                                    if constexpr (ADD) {
                                        alo = __uint_as_float(oldw[k] << 16);
                                        ahi = __uint_as_float(oldw[k] & 0xffff0000u);
                                    }
                                    if constexpr (EPI) {
                                        lo = rcf_lrelu(lo + eb[ni][0]);
                                        hi = rcf_lrelu(hi + eb[ni][1]);
                                        if constexpr (ADD) { lo = rcf_lrelu(lo + alo); hi = rcf_lrelu(hi + ahi); }
                                    } else if constexpr (ADD) {
                                        lo += alo;
                                        hi += ahi;
                                    }
                                    const unsigned pk = rcf_f2b2(lo, hi);
                                    __builtin_amdgcn_raw_buffer_store_b32(pk, rs_out, vo, so, 0);
                                    if constexpr (BST) {   // of the gradient the tensor holds: g = dY * lrelu'(z * scale + shift); sum g, sum g * z
                                        const float rlo = ok ? __uint_as_float(pk << 16) : 0.f, rhi = ok ? __uint_as_float(pk & 0xffff0000u) : 0.f;
                                        const float zlo = __uint_as_float(zw[k] << 16), zhi = __uint_as_float(zw[k] & 0xffff0000u);
                                        const float glo = rlo * rcf_lrelu_grad(zlo * bk[0][ni][0] + bk[1][ni][0]);
                                        const float ghi = rhi * rcf_lrelu_grad(zhi * bk[0][ni][1] + bk[1][ni][1]);
                                        s1[ni][0] += glo; s2[ni][0] += glo * zlo;
                                        s1[ni][1] += ghi; s2[ni][1] += ghi * zhi;
                                    } else if constexpr (STATS) {   // of the values the tensor holds
                                        const float rlo = ok ? __uint_as_float(pk << 16) : 0.f, rhi = ok ? __uint_as_float(pk & 0xffff0000u) : 0.f;
                                        s1[ni][0] += rlo; s2[ni][0] += rlo * rlo;
                                        s1[ni][1] += rhi; s2[ni][1] += rhi * rhi;
                                    }
                                }
                            }
                        }
                    }
                    if constexpr (STATS || BST) {
#pragma unroll
                        for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
                            for (int e = 0; e < 2; ++e) { st1[ni][e] += (double)s1[ni][e]; st2[ni][e] += (double)s2[ni][e]; }
                    }
                }
            };
            if constexpr (C::PM == 2) {   // four phases x the variants: an input gradient takes no statistics
                if (do_add) epilogue(std::true_type{}, std::false_type{});
                else epilogue(std::false_type{}, std::false_type{});
            } else if constexpr (BST) {   // the only writer of dY: nothing to add to, and the sums it takes are the block's, not its own
                epilogue(std::false_type{}, std::false_type{});
            } else if constexpr (EPI) {   // inference: no statistics
                if (do_add) epilogue(std::true_type{}, std::false_type{});
                else epilogue(std::false_type{}, std::false_type{});
            } else if (do_add) {
                if (want_stats) epilogue(std::true_type{}, std::true_type{});
                else epilogue(std::true_type{}, std::false_type{});
            } else {
                if (want_stats) epilogue(std::false_type{}, std::true_type{});
                else epilogue(std::false_type{}, std::false_type{});
            }
          };
          if constexpr (C::PM == 1) p4_epilogue();
          else if constexpr (C::PM == 2) {   // (phase by phase: these accumulate into dx where it has a second consumer, and odd sizes give the phases different extents)
              tile_epilogue(std::integral_constant<int, 0>{}, 0, 0);
              tile_epilogue(std::integral_constant<int, 1>{}, 0, 1);
              tile_epilogue(std::integral_constant<int, 2>{}, 1, 0);
              tile_epilogue(std::integral_constant<int, 3>{}, 1, 1);
          } else tile_epilogue(std::integral_constant<int, 0>{}, phase_out ? (oph >> 1) : a.ooy, phase_out ? (oph & 1) : a.oox);
        }
        RCF_T(t_w4);
        RCF_TACC(3, t_w4, t_w3);   // 3: epilogue
        tile = ntile;
        q = nq;
        buf ^= 1;
    }
#ifdef RCF_PHASE_TIMING
    tacc[7] = __builtin_amdgcn_s_memtime() - t_begin;
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&rcf_phase_cycles[i], tacc[i]);
#endif

    if (a.stats != nullptr) {
        rcf_wait_dma();
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem_b);   // [4 waves][BN][2]
#pragma unroll
        for (int ni = 0; ni < C::NT; ++ni) {
            // channel cp (even lane's own) = slot 0 of both lanes of the pair, channel cp + 1 (odd lane's own) = slot 1 of both
            const double a10 = st1[ni][0], a11 = st1[ni][1], a20 = st2[ni][0], a21 = st2[ni][1];
            const double t10 = a10 + __shfl_xor(a10, 1), t11 = a11 + __shfl_xor(a11, 1);
            const double t20 = a20 + __shfl_xor(a20, 1), t21 = a21 + __shfl_xor(a21, 1);
            double t1 = (li & 1) ? t11 : t10, t2 = (li & 1) ? t21 : t20;
            t1 += __shfl_xor(t1, 32);
            t2 += __shfl_xor(t2, 32);
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
                // BST: sum g * xhat = invstd * (sum g * z - mean * sum g), in fp64 from this workgroup's sums
                if constexpr (BST) t2 = (double)a.bk[3 * a.c_out + co] * (t2 - (double)a.bk[2 * a.c_out + co] * t1);
                a.stats[((size_t)blockIdx.x * 2 + 0) * a.c_out + co] = t1;
                a.stats[((size_t)blockIdx.x * 2 + 1) * a.c_out + co] = t2;
            }
        }
    }
}

template <class C>
int dma_grid_x(int ntiles, int ntile_n) {
    static int resident = 0;
    if (resident == 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_b16_kernel<C, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  C::LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_b16_kernel<C, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  C::LDS_BYTES);
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_b16_kernel<C, false>, 256, C::LDS_BYTES) != hipSuccess || per_cu < 1)
            per_cu = 1;
        resident = per_cu * num_cus();
    }
    int gx = resident / ntile_n;
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    return gx;
}

template <class C, bool EPI = false>
int launch_dma(const ConvArgs& a_, int ntile_n, hipStream_t st) {
    ConvArgs a = a_;
#ifdef RCF_B16_DIAG
    { const char* dg = getenv("RCF_B16_DIAG"); if (dg && dg[0] == '1') a.xcd_band += 64; }
#endif
    const int gx = dma_grid_x<C>(a.ntiles, ntile_n);
    hipLaunchKernelGGL((conv_b16_kernel<C, EPI>), dim3(gx, ntile_n, 1), dim3(256), C::LDS_BYTES, st, a);
    return rcf_launch_status();
}

template <class C>
int launch_dma_bst(const ConvArgs& a, int ntile_n, hipStream_t st) {
    const int gx = dma_grid_x<C>(a.ntiles, ntile_n);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_b16_kernel<C, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  C::LDS_BYTES);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_b16_kernel<C, false, true>), dim3(gx, ntile_n, 1), dim3(256), C::LDS_BYTES, st, a);
    return rcf_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------------------
// conv1x1_b16_kernel: 1x1 convolutions (the encoder's fusion convs W1 d, W2 d and the ResNet projections, src/networks.py:863-866,
// src/net_utils.py:300-307) and their input gradients on bf16 tensors.  A 1x1 convolution is a [pixels x Cin] x [Cin x Cout] GEMM with
// a tiny K (16 ... 64 at the resolutions that matter): HBM-bound, and there is nothing to stage -- 32 pixels x 16 channels of an NHWC
// bf16 tensor ARE one contiguous KiB in exactly the MFMA's A layout (lane (pixel, k half) <- 16 B), so the operand goes from global
// memory straight into the MFMA's registers; the whole weight matrix (<= 8192 elements) lives in registers for the lifetime of the
// wave.  No LDS, no barrier in the main loop.  (The f32-MFMA implicit-GEMM kernel ran these layers at 13-40 TFLOP/s = ~1.2 TB/s.)
template <int KST_, int NT_>
struct PwCfg {
    // k-steps of 16 input channels, 32-co tiles, 32-pixel blocks per trip (one block for 96 / 128 output channels: with two, the
    // accumulators next to the register-resident weight matrix spilled 60-102 VGPRs)
    // 128 / 256 input channels (the deep levels' input gradients) [r4]: the weight block of ONE workgroup must stay in registers, so a
    // workgroup takes 64 (KST 8) or 32 (KST 16) output channels and blockIdx.y walks the rest; the A operand is re-read per y from L2
    static constexpr int KST = KST_, NT = NT_, MT = KST_ >= 8 ? 1 : (NT_ <= 2 ? 4 : 1);
};

template <class C>
__global__ void __launch_bounds__(256, 2) conv1x1_b16_kernel(ConvArgs a) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int lh = lane >> 5;
    const int odd = li & 1;
    // weights: [chunk s][co][16] bf16, halves swizzled by (co >> 3) & 1 (pack_weights_split_kernel with BN = 32 NT, one n-tile)
    bf16x8 bw[C::KST][C::NT];
    const int co0 = blockIdx.y * 32 * C::NT;   // this workgroup's n-tile
    {
        const unsigned char* wp = reinterpret_cast<const unsigned char*>(a.wp) + (size_t)blockIdx.y * (C::KST * 32 * C::NT * 32);
#pragma unroll
        for (int s = 0; s < C::KST; ++s)
#pragma unroll
            for (int ni = 0; ni < C::NT; ++ni) {
                const int co = ni * 32 + li;
                bw[s][ni] = as_bf16x8(*reinterpret_cast<const u32x4*>(wp + ((size_t)(s * 32 * C::NT + co) * 32) + ((lh ^ ((co >> 3) & 1)) * 16)));
            }
    }
    const long long npix = (long long)a.n * a.h_out * a.w_out;
    const long long nblk32 = (npix + 31) / 32;
    const int nwaves = gridDim.x * 4;
    const unsigned short* src = reinterpret_cast<const unsigned short*>(a.in1);
    unsigned short* outp = reinterpret_cast<unsigned short*>(a.out) + co0;   // this n-tile's first channel: the epilogue below counts from it
    const int c_lim = a.c_out - co0;
    const bool want_stats = a.stats != nullptr;
    double st1[C::NT][2], st2[C::NT][2];
#pragma unroll
    for (int ni = 0; ni < C::NT; ++ni) { st1[ni][0] = st1[ni][1] = 0.0; st2[ni][0] = st2[ni][1] = 0.0; }

    for (long long blk0 = ((long long)blockIdx.x * 4 + wave) * C::MT; blk0 < nblk32; blk0 += (long long)nwaves * C::MT) {
        // A operands of MT blocks: lane (li, lh) <- channels 16 s + 8 lh .. + 7 of output pixel blk * 32 + li (source pixel through the stride)
        u32x4 av[C::MT][C::KST];
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
            const long long p = (blk0 + mt) * 32 + li;
            bool ok = p < npix;
            size_t sp = 0;
            if (ok) {
                if (a.gather1 == RCF_GATHER_ZERO_INSERT) {   // input gradient of a stride-2 1x1 convolution: dz sits at the even positions
                    const int ox = (int)(p % a.w_out);
                    const long long t = p / a.w_out;
                    const int oy = (int)(t % a.h_out);
                    const int im = (int)(t / a.h_out);
                    ok = !((oy | ox) & 1) && (oy >> 1) < a.h1 && (ox >> 1) < a.w1;
                    sp = ((size_t)im * a.h1 + (size_t)(oy >> 1)) * a.w1 + (size_t)(ox >> 1);
                } else if (a.stride == 1) sp = (size_t)p;
                else {
                    const int ox = (int)(p % a.w_out);
                    const long long t = p / a.w_out;
                    const int oy = (int)(t % a.h_out);
                    const int im = (int)(t / a.h_out);
                    sp = ((size_t)im * a.h_in + (size_t)oy * a.stride) * a.w_in + (size_t)ox * a.stride;
                }
            }
#pragma unroll
            for (int s = 0; s < C::KST; ++s)
                av[mt][s] = ok ? *reinterpret_cast<const u32x4*>(src + sp * a.c1 + s * 16 + lh * 8) : u32x4{0u, 0u, 0u, 0u};
        }
        f32x16 acc[C::MT][C::NT];
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
            for (int ni = 0; ni < C::NT; ++ni) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][ni][r] = 0.f;
#pragma unroll
                for (int s = 0; s < C::KST; ++s)
                    acc[mt][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(av[mt][s]), bw[s][ni], acc[mt][ni], 0, 0, 0);
            }
        // epilogue: lane holds channel ni * 32 + li of pixels row(r); pairs of lanes exchange one value and store one dword per pixel pair
        auto epilogue = [&](auto add_tag) __attribute__((always_inline)) {
            constexpr bool ADD = decltype(add_tag)::value;
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt) {
                float s1[C::NT][2], s2[C::NT][2];
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni) { s1[ni][0] = s1[ni][1] = 0.f; s2[ni][0] = s2[ni][1] = 0.f; }
                const long long pb0 = (blk0 + mt) * 32;
                unsigned oldw[8][C::NT];
                if (ADD) {
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const long long p = pb0 + rcf_mfma_row(2 * g, lh) + odd;
#pragma unroll
                        for (int ni = 0; ni < C::NT; ++ni) {
                            const int cp = (ni * 32 + li) & ~1;
                            oldw[g][ni] = *reinterpret_cast<const unsigned*>(outp + ((p < npix && cp < c_lim) ? (size_t)p * a.c_out + cp : 0));
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 8; ++g)   // one wait for all of them here, not a vmcnt(0) in front of every masked store below
#pragma unroll
                        for (int ni = 0; ni < C::NT; ++ni) asm volatile("" : "+v"(oldw[g][ni]));
                }
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const int rj = 2 * g;                                    // accumulator rows rj, rj + 1 are consecutive pixels
                    const long long p = pb0 + rcf_mfma_row(rj, lh) + odd;   // the pixel this lane stores
#pragma unroll
                    for (int ni = 0; ni < C::NT; ++ni) {
                        const int cp = (ni * 32 + li) & ~1;
                        const bool ok = p < npix && cp < c_lim;
                        const float a0 = acc[mt][ni][rj], a1 = acc[mt][ni][rj + 1];
                        const float mine = odd ? a1 : a0, give = odd ? a0 : a1;
                        const float got = __uint_as_float(rcf_dpp_u32<0xB1>(__float_as_uint(give)));
                        float lo = odd ? got : mine, hi = odd ? mine : got;
                        if (ADD) {
                            lo += __uint_as_float(oldw[g][ni] << 16);
                            hi += __uint_as_float(oldw[g][ni] & 0xffff0000u);
                        }
                        const unsigned pk = rcf_f2b2(lo, hi);
                        if (ok) {
                            *reinterpret_cast<unsigned*>(outp + (size_t)p * a.c_out + cp) = pk;
                            if (want_stats) {
                                const float rlo = __uint_as_float(pk << 16), rhi = __uint_as_float(pk & 0xffff0000u);
                                s1[ni][0] += rlo; s2[ni][0] += rlo * rlo;
                                s1[ni][1] += rhi; s2[ni][1] += rhi * rhi;
                            }
                        }
                    }
                }
                if (want_stats) {
#pragma unroll
                    for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
                        for (int e = 0; e < 2; ++e) { st1[ni][e] += (double)s1[ni][e]; st2[ni][e] += (double)s2[ni][e]; }
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
            const double a10 = st1[ni][0], a11 = st1[ni][1], a20 = st2[ni][0], a21 = st2[ni][1];
            const double t10 = a10 + __shfl_xor(a10, 1), t11 = a11 + __shfl_xor(a11, 1);
            const double t20 = a20 + __shfl_xor(a20, 1), t21 = a21 + __shfl_xor(a21, 1);
            double t1 = (li & 1) ? t11 : t10, t2 = (li & 1) ? t21 : t20;
            t1 += __shfl_xor(t1, 32);
            t2 += __shfl_xor(t2, 32);
            if (lh == 0) {
                red[((wave * C::NT + ni) * 32 + li) * 2 + 0] = t1;
                red[((wave * C::NT + ni) * 32 + li) * 2 + 1] = t2;
            }
        }
        __syncthreads();
        if (tid < 32 * C::NT && tid < c_lim) {
            double t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                t1 += red[((w * C::NT + tid / 32) * 32 + (tid & 31)) * 2 + 0];
                t2 += red[((w * C::NT + tid / 32) * 32 + (tid & 31)) * 2 + 1];
            }
            a.stats[((size_t)blockIdx.x * 2 + 0) * a.c_out + co0 + tid] = t1;
            a.stats[((size_t)blockIdx.x * 2 + 1) * a.c_out + co0 + tid] = t2;
        }
    }
}

// output channels one workgroup of the pointwise kernel takes (in 32-channel tiles) and the pixel blocks per trip
inline int pw_nt(int kst, int nt_total) {
    if (kst >= 16) return 1;
    if (kst >= 8 || (kst == 4 && nt_total == 4)) return nt_total < 2 ? nt_total : 2;   // (64 -> 128 in one workgroup spilled 34 VGPRs)
    return nt_total;
}
inline int pw_mt(int kst, int nt) { return kst >= 8 ? 1 : (nt <= 2 ? 4 : 1); }

// grid.x of the pointwise kernel = number of BatchNorm partial rows it writes
inline int pw_grid(long long npix, int kst, int nt, int gy) {
    const int per_trip = 32 * 4 * pw_mt(kst, nt);                      // 4 waves x MT blocks of 32 pixels per workgroup trip
    const long long trips = (npix + per_trip - 1) / per_trip;
    long long g = 2 * (long long)num_cus() * 2 / gy;                   // two workgroups per CU, two trips' worth of slack
    if (g > trips) g = trips;
    if (g < 1) g = 1;
    return (int)g;
}

template <class C>
int launch_pw(const ConvArgs& a, hipStream_t st) {
    static_assert(C::MT == (C::KST >= 8 ? 1 : (C::NT <= 2 ? 4 : 1)), "pw_mt mirrors PwCfg::MT");
    const int gy = (a.c_out + 32 * C::NT - 1) / (32 * C::NT);
    const int g = pw_grid((long long)a.n * a.h_out * a.w_out, C::KST, C::NT, gy);
    hipLaunchKernelGGL((conv1x1_b16_kernel<C>), dim3(g, gy), dim3(256), 0, st, a);
    return rcf_launch_status();
}

template <class F>
int dispatch_pw(int kst, int nt, F&& f) {
#define RCF_PW(K, N) if (kst == K && nt == N) return f(PwCfg<K, N>{})
    RCF_PW(1, 1); RCF_PW(1, 2); RCF_PW(1, 3); RCF_PW(1, 4);
    RCF_PW(2, 1); RCF_PW(2, 2); RCF_PW(2, 3); RCF_PW(2, 4);
    RCF_PW(3, 1); RCF_PW(3, 2);
    RCF_PW(4, 1); RCF_PW(4, 2); RCF_PW(4, 3);
    RCF_PW(8, 1); RCF_PW(8, 2); RCF_PW(16, 1);
#undef RCF_PW
    return RCF_EUNSUPPORTED;
}
