// Wave-specialised form of the two-plane fp16 split convolution (conv_split_kernel<SplitCfg<..., NPL = 2>>): forward and input gradient
// of the fp32 configuration's 3x3 / 2x2 / 4x4-stem / stride-2 layers (src/net_utils.py:29-91, :156-198, :253-323, :473-569).
// Included by rcf_conv_impl.h inside its anonymous namespace (uses ConvArgs, SplitCfg, rcf_f16_planes, rcf_mfma_split, ...).
//
// Why.  In conv_split_kernel every wave does everything in turn: global loads -> fp32 -> two-plane conversion -> ds_write -> barrier ->
// MFMA rows -> epilogue.  tools/phase_timing.py (round 3) put a wave's MFMA rows at 18-25 % of its time; with two workgroups (two waves
// per SIMD) the matrix pipe cannot be busier than twice that (PMC: 0.34).  Here the roles are separate waves:
//   * waves 0-3 (one per SIMD), CONSUMERS: nothing but ds_read_b128 + MFMA per kernel row, one barrier per row, and the epilogue.
//     Same tiles, same LDS operand layout, same hand-ordered MFMA / LDS-read interleave, same epilogue as conv_split_kernel.
//   * waves 4-5, PRODUCERS: tile bookkeeping (the gather addresses), the fp32 halo loads of the chunk AFTER the next one into
//     registers, the scale + two-plane split + ds_write of the NEXT chunk into the other A buffer, the LDS-DMA of the next kernel
//     row of packed weights.  Their VALU work runs beside the consumers' MFMAs on the same SIMDs (separate pipes) instead of in
//     front of them.
// 384 threads, two workgroups per CU = three waves per SIMD (<= 168 registers: a consumer holds 64 accumulators + 64 operand registers
// and no staging state; a producer holds the raw tile).  The A tile is double-buffered where two workgroups still fit the LDS (all
// 256-pixel tiles); the 512-pixel x 32-co configuration keeps one buffer and the two-barrier hand-over.
// Barrier discipline: both roles walk the same (tile, chunk, kernel row) sequence and execute the same barriers.
#pragma once
#ifndef RCF_WS_VARIANT
#define RCF_WS_VARIANT 0      // probe builds only (the harness that instantiated this kernel went with the round-4 A-tile layout it targets; kept as the record of the experiment, not built): bit flags that cut parts of the consumer away (wrong results)
#endif
#ifndef RCF_WS_PROBE_ROLE
#define RCF_WS_PROBE_ROLE 0   // probe builds only (tools/probe): 1 = compile the consumer role alone, 2 = the producer role alone
#endif

constexpr int WS_THREADS = 384, WS_NPROD = 128;

template <class C>
struct WsLayout {
    static constexpr bool ADB = 2 * (2 * C::A_BYTES + 2 * C::B_PIECE_BYTES + C::COEF_BYTES) <= 160 * 1024;   // A double-buffered
    static constexpr int NBUF = ADB ? 2 : 1;
    static constexpr int LDS_BYTES = NBUF * C::A_BYTES + 2 * C::B_PIECE_BYTES + C::COEF_BYTES;
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
    static_assert(ADB, "the wave-specialised kernel double-buffers the A tile (256-pixel tiles)");
    static constexpr int NA = (C::NPIX + 31) / 32;   // halo pixels per producer thread (128 producer threads = 32 pixels x 4 channel quads)
    static_assert(NA <= 60, "s_waitcnt vmcnt has six bits");
};

template <class F, int... K>
__device__ __forceinline__ void rcf_static_for(F&& f, std::integer_sequence<int, K...>) { (f(std::integral_constant<int, K>{}), ...); }

// s_waitcnt vmcnt(N) only (expcnt / lgkmcnt untouched), N a compile-time constant: vmcnt is split over simm16[3:0] and [15:14]
template <int N>
__device__ __forceinline__ void rcf_wait_vm() { __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14)); }

template <class C, bool BST = false>
__global__ void __launch_bounds__(WS_THREADS, 3) conv_split_ws_kernel(ConvArgs a) {
    using L = WsLayout<C>;
    using SI = SAct;
    using SO = SAct;
    static_assert(C::NPL == 2 && !SAct::B16, "fp32 tensors on two fp16 planes");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* As = smem_b;
    unsigned char* Bs = smem_b + L::NBUF * C::A_BYTES;
    float* coef_lds = reinterpret_cast<float*>(smem_b + L::NBUF * C::A_BYTES + 2 * C::B_PIECE_BYTES);   // [2][c1 + c2]: scale, shift
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31;
    const int lh = lane >> 5;
    if (a.coef1 != nullptr || a.coef2 != nullptr) {   // BN-on-load table: identity (1, 0) for a source without coefficients
        const int ctot = a.c1 + a.c2;
        for (int i = tid; i < ctot; i += WS_THREADS) {
            const bool s1 = i < a.c1;
            const float* cf = s1 ? a.coef1 : a.coef2;
            const int cs = s1 ? a.c1 : a.c2, ci = s1 ? i : i - a.c1;
            coef_lds[i] = cf ? cf[ci] : 1.f;
            coef_lds[ctot + i] = cf ? cf[cs + ci] : 0.f;
        }
        __syncthreads();
    }
    const SplitScales sc = rcf_split_scales(a.amax_a1, a.amax_a2, a.amax_b);
    const int nchunk = a.nchunk1 + a.nchunk2;
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(a.wp) + (size_t)blockIdx.y * nchunk * C::WCHUNK_BYTES;
    const int n0 = blockIdx.y * C::BN;
    const int nitem = a.phase_sum ? 4 * nchunk : nchunk;   // A tiles per output tile
    auto chunk_base = [&](int item) __attribute__((always_inline)) -> const unsigned char* {
        const int ph = a.phase_sum ? item / nchunk : 0;
        const int q = a.phase_sum ? item - ph * nchunk : item;
        return wp + (size_t)ph * a.wp_phase_stride * 4 + (size_t)q * C::WCHUNK_BYTES;
    };
    double st1[C::NT], st2[C::NT];
#pragma unroll
    for (int ni = 0; ni < C::NT; ++ni) { st1[ni] = 0.0; st2[ni] = 0.0; }

    if (wave >= 4) {
#if RCF_WS_PROBE_ROLE != 1
        // =============================================================================================== PRODUCER (waves 4, 5)
        // Lean by construction: the gather mode is a wave-uniform switch OUTSIDE the pixel loop, per-lane conditions are selects (no
        // exec-mask branches), and the work of an item is spread over its kernel rows -- in row ky the thread converts its quads
        // [ky * NA / KSY, (ky + 1) * NA / KSY) of the NEXT item into the other A buffer and reloads exactly those registers with the
        // item after it, so every load has a whole item to land and no row carries more than its share in front of the barrier.
        const int ptid = tid - 256;
        constexpr int NA = L::NA;
        int pix[NA];        // halo pixel -> source pixel index of the tile being LOADED (-1: zero)
        f32x4 ra[NA];
        unsigned okm = 0u;  // bit i: ra[i] holds real data (else the clamped load is replaced by zero at store time)
        // the item whose quads sit in ra (being converted) and the item being loaded (one later)
        bool cv_ttf = false; int cv_tch = 0;
        const float* ld_src = a.in1; int ld_csrc = a.c1, ld_cld = 0, ld_tch = 0; bool ld_cok = false, ld_ttf = false;

        auto setup_mode = [&](auto mode_tag, int tile, bool first, int ph) __attribute__((always_inline)) {
            constexpr int MODE = decltype(mode_tag)::value;   // 0 direct, 1 nearest, 2 zero-insert, 3 strided2, 4 virtual tall
            int t = tile;
            const int tx = t % a.tiles_x;
            t /= a.tiles_x;
            const int ty = t % a.tiles_y;
            const int img = t / a.tiles_y;
            const int pa = a.phase_sum ? (ph >> 1) : a.pad, pbx = a.phase_sum ? (ph & 1) : a.pad_x;
            const int ioy = a.phase_sum ? (ph >> 1) : a.ioy, iox = a.phase_sum ? (ph & 1) : a.iox;
            const int iy0 = ty * C::TH * C::LSTEP - pa;
            const int ix0 = tx * C::PX * C::LSTEP - pbx;
            const int hs = first ? a.h1 : a.h_in, ws = first ? a.w1 : a.w_in;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int p = (ptid >> 2) + i * 32;
                const int hy = p / C::HXP;
                const int hx = p - hy * C::HXP;
                const int ly = iy0 + hy, lx = ix0 + hx;
                bool in = (p < C::NPIX) & (ly >= 0) & (lx >= 0) & (lx < a.w_in);
                int v;
                if constexpr (MODE == 4) {
                    const int im = (int)(((float)ly + 0.5f) * a.inv_hp);
                    const int y = ly - im * a.hp;
                    in = in & (im < a.nimg) & (y < a.h_in);
                    v = (im * hs + y) * ws + lx;
                } else {
                    in = in & (ly < a.h_in);
                    int py = ly, px = lx;
                    if constexpr (MODE == 1) {
                        py = min((int)floorf((float)ly * a.sy), hs - 1);
                        px = min((int)floorf((float)lx * a.sx), ws - 1);
                    } else if constexpr (MODE == 3) {
                        py = 2 * ly + ioy;
                        px = 2 * lx + iox;
                        in = in & (py < hs) & (px < ws);
                    } else if constexpr (MODE == 2) {
                        in = in & (((ly | lx) & 1) == 0);
                        py = ly >> 1;
                        px = lx >> 1;
                        in = in & (py < hs) & (px < ws);
                    }
                    v = (img * hs + py) * ws + px;
                }
                pix[i] = in ? v : -1;
            }
        };
        auto setup = [&](int tile, bool first, int ph) __attribute__((always_inline)) {
            const int gmode = first ? a.gather1 : RCF_GATHER_DIRECT;
            if (a.vt) setup_mode(std::integral_constant<int, 4>{}, tile, first, ph);
            else if (gmode == RCF_GATHER_NEAREST) setup_mode(std::integral_constant<int, 1>{}, tile, first, ph);
            else if (gmode == RCF_GATHER_STRIDED2) setup_mode(std::integral_constant<int, 3>{}, tile, first, ph);
            else if (gmode == RCF_GATHER_ZERO_INSERT) setup_mode(std::integral_constant<int, 2>{}, tile, first, ph);
            else setup_mode(std::integral_constant<int, 0>{}, tile, first, ph);
        };
        // begin loading an item: its source / channel quad (and, at a source boundary, the tile's gather addresses)
        auto load_begin = [&](int tile, int item) __attribute__((always_inline)) {
            const int ph = a.phase_sum ? item / nchunk : 0;
            const int q = a.phase_sum ? item - ph * nchunk : item;
            const bool first = q < a.nchunk1;
            if (q == 0 || q == a.nchunk1) setup(tile, first, ph);
            ld_src = first ? a.in1 : a.in2;
            ld_csrc = first ? a.c1 : a.c2;
            const int cch = (first ? q : q - a.nchunk1) * 16 + (ptid & 3) * 4;
            ld_cok = cch < ld_csrc;
            ld_cld = ld_cok ? cch : 0;
            ld_tch = (first ? 0 : a.c1) + ld_cld;
            ld_ttf = (first ? a.coef1 : a.coef2) != nullptr;
        };
        // quads [i0, i1) of the item being loaded: unconditional loads from a clamped (always valid) address, zero-selected at store time
        auto load_part = [&](auto i0_tag, auto i1_tag) __attribute__((always_inline)) {
            constexpr int I0 = decltype(i0_tag)::value, I1 = decltype(i1_tag)::value;
#pragma unroll
            for (int i = I0; i < I1; ++i) {
                const bool ok = ld_cok & (pix[i] >= 0);
                okm = (okm & ~(1u << i)) | (ok ? (1u << i) : 0u);
                ra[i] = rcf_ld4<SI>(ld_src, (size_t)(pix[i] < 0 ? 0 : pix[i]) * ld_csrc + ld_cld);
            }
        };
        auto store_part_impl = [&](auto bn_tag, auto i0_tag, auto i1_tag, int buf) __attribute__((always_inline)) {
            constexpr bool BN = decltype(bn_tag)::value;
            constexpr int I0 = decltype(i0_tag)::value, I1 = decltype(i1_tag)::value;
            f32x4 tsc = {1.f, 1.f, 1.f, 1.f}, tsh = {0.f, 0.f, 0.f, 0.f};
            if (BN) {
                tsc = *reinterpret_cast<const f32x4*>(coef_lds + cv_tch);
                tsh = *reinterpret_cast<const f32x4*>(coef_lds + a.c1 + a.c2 + cv_tch);
            }
            unsigned char* Ab = As + buf * C::A_BYTES;
#pragma unroll
            for (int i = I0; i < I1; ++i) {
                const int p = (ptid >> 2) + i * 32;
                if ((i + 1) * 32 <= C::NPIX || p < C::NPIX) {   // (only the last quad of a thread can lie beyond the tile)
                    float xin[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float xv = ra[i][e];
                        if (BN) xv = rcf_lrelu(xv * tsc[e] + tsh[e]);   // the producing block's BatchNorm + LeakyReLU, applied on load
                        xin[e] = ((okm >> i) & 1u) ? xv : 0.f;
                    }
                    const int cq = ptid & 3;
                    unsigned char* dst = Ab + p * 32 + (((cq >> 1) ^ ((p >> 3) & 1)) * 16) + (cq & 1) * 8;
                    const rcf_f16_pair q0 = rcf_f16_planes(xin[0] * sc.sa, xin[1] * sc.sa), q1 = rcf_f16_planes(xin[2] * sc.sa, xin[3] * sc.sa);
                    const u32x2 w0 = {q0.p0, q1.p0}, w1 = {q0.p1, q1.p1};
                    *reinterpret_cast<u32x2*>(dst) = w0;
                    *reinterpret_cast<u32x2*>(dst + C::A_PLANE_BYTES) = w1;
                }
            }
        };
        auto store_part = [&](auto i0_tag, auto i1_tag, int buf) __attribute__((always_inline)) {
            if (cv_ttf) store_part_impl(std::true_type{}, i0_tag, i1_tag, buf);
            else store_part_impl(std::false_type{}, i0_tag, i1_tag, buf);
        };
        auto next_of = [&](int tile, int q, int* ntile, int* nq) __attribute__((always_inline)) {
            *ntile = tile;
            *nq = q + 1;
            if (*nq == nitem) { *nq = 0; *ntile = tile + gridDim.x; }
        };
        using I0 = std::integral_constant<int, 0>;
        using IN = std::integral_constant<int, NA>;

#ifdef RCF_PHASE_TIMING
        unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
        int tile = blockIdx.x, q = 0, cur = 0;
        if (tile < a.ntiles) {
            load_begin(tile, 0);
            load_part(I0{}, IN{});
            cv_ttf = ld_ttf; cv_tch = ld_tch;
            store_part(I0{}, IN{}, 0);
        }
        __syncthreads();   // publishes A[0] (and the consumers' weight slot 0)
        {
            int t1, q1;
            next_of(tile, q, &t1, &q1);
            if (tile < a.ntiles && t1 < a.ntiles) {   // ra: the item after the current one, from here on
                load_begin(t1, q1);
                load_part(I0{}, IN{});
            }
        }
        while (tile < a.ntiles) {
            int ntile, nq, n2tile, n2q;
            next_of(tile, q, &ntile, &nq);
            next_of(ntile, nq, &n2tile, &n2q);
            const bool more = ntile < a.ntiles, more2 = n2tile < a.ntiles;
            cv_ttf = ld_ttf; cv_tch = ld_tch;          // what sits in ra now is the next item
            bool began = false;
            rcf_static_for([&](auto ky_tag) __attribute__((always_inline)) {
                constexpr int ky = decltype(ky_tag)::value;
                using P0 = std::integral_constant<int, (ky * NA) / C::KSY>;
                using P1 = std::integral_constant<int, ((ky + 1) * NA) / C::KSY>;
                RCF_T(t_p0);
                if (more) {
                    store_part(P0{}, P1{}, cur ^ 1);
                    if (more2) {
                        if (!began) { load_begin(n2tile, n2q); began = true; }
                        load_part(P0{}, P1{});
                    }
                }
                RCF_T(t_p1);
                RCF_TACC(4, t_p1, t_p0);   // 4: producer work of the row (convert + ds_write + reload)
                if (!(RCF_WS_VARIANT & 4) || ky == C::KSY - 1) __syncthreads();
                RCF_T(t_p2);
                RCF_TACC(5, t_p2, t_p1);   // 5: producer at the row's barrier
            }, std::make_integer_sequence<int, C::KSY>{});
            cur ^= 1;
            tile = ntile;
            q = nq;
        }
#ifdef RCF_PHASE_TIMING
        tacc[6] = __builtin_amdgcn_s_memtime() - t_begin;   // 6: whole producer wave
        if (lane == 0)
            for (int i = 4; i < 7; ++i) atomicAdd(&rcf_phase_cycles[i], tacc[i]);
#endif
#endif
    } else {
#if RCF_WS_PROBE_ROLE != 2
        // =============================================================================================== CONSUMER (waves 0-3)
        int apix[C::MT];   // halo pixel of this lane's output pixel at tap (0, 0)
#pragma unroll
        for (int mi = 0; mi < C::MT; ++mi) {
            const int tr = (wave * C::MT + mi) * C::PY + li / C::PX;
            const int tc = li % C::PX;
            apix[mi] = tr * C::LSTEP * C::HXP + tc * C::LSTEP;
        }
        const int bbase = li * 32 + ((lh ^ ((li >> 3) & 1)) * 16);   // row co = ni*32 + li; (ni*32) keeps (co>>3)&1 == (li>>3)&1
        f32x16 acc[C::MT][C::NT];
        bf16x8 av[2][C::NPL][C::MT], bv[2][C::NPL][C::NT];
        auto fetch_a = [&](int ky, int kx, int slot, int buf) __attribute__((always_inline)) {
            const unsigned char* Ab = As + buf * C::A_BYTES;
#pragma unroll
            for (int mi = 0; mi < C::MT; ++mi) {
                int ap = apix[mi];
                asm volatile("" : "+v"(ap));   // recompute the swizzled address per tap (hoisted, the 2 x T addresses cost registers)
                const int p = ap + ky * C::HXP + kx;
                const int ao = p * 32 + ((lh ^ ((p >> 3) & 1)) * 16);
#pragma unroll
                for (int pl = 0; pl < C::NPL; ++pl) av[slot][pl][mi] = as_bf16x8(*reinterpret_cast<const u32x4*>(Ab + pl * C::A_PLANE_BYTES + ao));
            }
        };
        auto fetch_b = [&](int kx, int slot, int bslot) __attribute__((always_inline)) {
            const unsigned char* Bp = Bs + bslot * C::B_PIECE_BYTES;
#pragma unroll
            for (int pl = 0; pl < C::NPL; ++pl)
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni)
                    bv[slot][pl][ni] = as_bf16x8(*reinterpret_cast<const u32x4*>(Bp + pl * C::B_PLANE_BYTES + (kx * C::BN + ni * 32) * 32 + bbase));
        };

        // one kernel row of pre-split weights: straight copy global -> LDS piece `buf` by LDS-DMA (each wave instruction moves 1 KiB to a
        // wave-uniform LDS base + 16 B x lane).  The consumers issue it -- they have no other loads in flight, so the vmcnt(0) in front of
        // the row's barrier waits for nothing else, and the producers' halo loads keep a whole item to land.
        auto copy_b = [&](const unsigned char* cbase, int ky, int buf) __attribute__((always_inline)) {
            constexpr int NKB = C::B_PIECE_BYTES / 1024;
            static_assert(C::B_PIECE_BYTES % 1024 == 0, "weight piece must be whole KiB");
            const unsigned char* wsrc = cbase + (size_t)ky * C::B_PIECE_BYTES + lane * 16;
#pragma unroll
            for (int i = 0; i < (NKB + 3) / 4; ++i) {
                int kb = i * 4 + wave;
                if ((i + 1) * 4 > NKB) kb = kb < NKB ? kb : NKB - 1;   // ragged tail: a duplicate copy of the last KiB is harmless
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + kb * 1024),
                                                 (__attribute__((address_space(3))) void*)(Bs + buf * C::B_PIECE_BYTES + kb * 1024), 16, 0, 0);
            }
        };
#ifdef RCF_PHASE_TIMING
        unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
        int tile = blockIdx.x, q = 0, pb = 0, cur = 0;
        const unsigned char* cb_cur = wp;
        if (tile < a.ntiles) copy_b(cb_cur, 0, 0);
        rcf_wait_dma();
        __syncthreads();   // A[0] and weight slot 0 are there
        if (tile < a.ntiles) fetch_a(0, 0, 0, 0);
        while (tile < a.ntiles) {
            int ntile = tile, nq = q + 1;
            if (nq == nitem) { nq = 0; ntile = tile + gridDim.x; }
            const bool more = ntile < a.ntiles;
            const unsigned char* cb_next = more ? chunk_base(nq) : wp;
            if (q == 0) {
#pragma unroll
                for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
            }
            const unsigned char* Ab = As + cur * C::A_BYTES;
#pragma unroll
            for (int ky = 0; ky < C::KSY; ++ky) {
                const bool last_row = ky == C::KSY - 1;
                // every row starts in register set 0: its A operands were fetched before the barrier that published its weights
                RCF_T(t_c0);
                fetch_b(0, 0, pb);
                if (!(RCF_WS_VARIANT & 2)) {
                    if (!last_row) copy_b(cb_cur, ky + 1, pb ^ 1);   // nobody reads that slot during this row
                    else if (more) copy_b(cb_next, 0, pb ^ 1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kx = 0; kx < C::KSX; ++kx) {
                    const int cs = kx & 1;
                    // the 3 x MT x NT MFMAs of this tap in product-major, accumulator-round-robin order with the next tap's LDS reads
                    // issued ONE AT A TIME between them (conv_split_kernel: issued as a block they stall the wave's MFMA issue)
                    constexpr int MN = C::MT * C::NT, NMF = C::NP * MN, NRD = C::NPL * (C::MT + C::NT);
                    const bool has_next = kx + 1 < C::KSX;
                    int nr = 0;
                    int ao_next[C::MT];
#pragma unroll
                    for (int j = 0; j < NMF; ++j) {
                        constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};   // two planes: a1b0, a0b1, a0b0
                        const int pj = j / MN, mi = (j % MN) / C::NT, ni = j % C::NT;
                        acc[mi][ni] = rcf_mfma_split<C::NPL>(av[cs][PA2[pj]][mi], bv[cs][PB2[pj]][ni], acc[mi][ni]);
                        if (has_next) {
#pragma unroll
                            for (int rep = 0; rep < 3; ++rep) {
                                if (nr < NRD && (nr + 1) * NMF <= (j + 1) * NRD && !(RCF_WS_VARIANT & 8)) {
                                    __builtin_amdgcn_sched_barrier(0);
                                    if (nr < C::NPL * C::MT) {
                                        const int rmi = nr / C::NPL, pl = nr % C::NPL;
                                        if (pl == 0) {
                                            int ap = apix[rmi];
                                            asm volatile("" : "+v"(ap));
                                            const int p = ap + ky * C::HXP + kx + 1;
                                            ao_next[rmi] = p * 32 + ((lh ^ ((p >> 3) & 1)) * 16);
                                        }
                                        av[cs ^ 1][pl][rmi] = as_bf16x8(*reinterpret_cast<const u32x4*>(Ab + pl * C::A_PLANE_BYTES + ao_next[rmi]));
                                    } else {
                                        const int rb = nr - C::NPL * C::MT, pl = rb / C::NT, rni = rb % C::NT;
                                        bv[cs ^ 1][pl][rni] = as_bf16x8(*reinterpret_cast<const u32x4*>(
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
                RCF_T(t_c1);
                RCF_TACC(0, t_c1, t_c0);   // 0: the row's B reads, DMA issue, MFMAs + interleaved LDS reads
                if (last_row) {
                    if (q == nitem - 1 && !(RCF_WS_VARIANT & 1)) {
                        // ---- epilogue (conv_split_kernel's): rescale, strided / phase output, +=, BatchNorm statistics or backward sums
                        int t = tile;
                        const int tx = t % a.tiles_x;
                        t /= a.tiles_x;
                        const int ty = t % a.tiles_y;
                        const int img = t / a.tiles_y;
                        const int oy0 = ty * C::TH;
                        const int ox0 = tx * C::PX;
                        const bool want_stats = a.stats != nullptr && !(RCF_WS_VARIANT & 16);
                        const bool add_old = a.accumulate != 0;
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
                                for (int e = 0; e < 2; ++e) asm volatile("" : "+v"(bk[e][ni]));   // land them here (see `old` below)
                        }
                        // undo the operand scales: two exact multiplications by powers of two
#pragma unroll
                        for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
                            for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
                                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = acc[mi][ni][r] * sc.ia * sc.ib;
#pragma unroll
                        for (int mi = 0; mi < C::MT; ++mi) {
#pragma unroll
                            for (int r0 = 0; r0 < 16; r0 += 4) {
                                // accumulator rows r0 .. r0 + 3 of a lane are four consecutive pixels of ONE tile row
                                size_t pbase[4];
                                bool pok[4];
                                {
                                    const int row = rcf_mfma_row(r0, lh);
                                    int oy = oy0 + (wave * C::MT + mi) * C::PY + row / C::PX;
                                    const int ox = ox0 + row % C::PX;
                                    int im = img;
                                    if (a.vt) {   // virtual row -> (image, row); separator rows produce no output
                                        im = (int)(((float)oy + 0.5f) * a.inv_hp);
                                        oy -= im * a.hp;
                                        if (im >= a.nimg) oy = a.h_out;
                                    }
                                    const int py = oy * a.os + a.ooy, px = ox * a.os + a.oox;
                                    const bool rowvalid = oy < a.h_out && py < a.ohp;
                                    const size_t base0 = (((size_t)im * a.ohp + py) * a.owp + px) * a.c_out;
                                    const int pstep = a.os * a.c_out;
#pragma unroll
                                    for (int j = 0; j < 4; ++j) {
                                        pok[j] = rowvalid && ox + j < a.w_out && px + j * a.os < a.owp;
                                        pbase[j] = base0 + (size_t)(j * pstep);
                                    }
                                }
                                float zv[BST ? 4 : 1][C::NT];
                                if constexpr (BST) {   // z of the BatchNorm block at the group's outputs
#pragma unroll
                                    for (int j = 0; j < 4; ++j)
#pragma unroll
                                        for (int ni = 0; ni < C::NT; ++ni) {
                                            const int co = n0 + ni * 32 + li;
                                            zv[j][ni] = rcf_ld1<SO>(a.bz, (pok[j] && co < a.c_out) ? pbase[j] + co : 0);
                                        }
                                }
                                if (add_old) {   // gradient accumulation: all old values of the group in flight together, landed inside
                                                 // this branch (gfx9 counts stores in vmcnt: a pending load at the join puts a
                                                 // vmcnt(0) in front of every store of the plain path too)
                                    float old[4][C::NT];
#pragma unroll
                                    for (int j = 0; j < 4; ++j)
#pragma unroll
                                        for (int ni = 0; ni < C::NT; ++ni) {
                                            const int co = n0 + ni * 32 + li;
                                            old[j][ni] = rcf_ld1<SO>(a.out, (pok[j] && co < a.c_out) ? pbase[j] + co : 0);
                                        }
#pragma unroll
                                    for (int j = 0; j < 4; ++j)
#pragma unroll
                                        for (int ni = 0; ni < C::NT; ++ni) acc[mi][ni][r0 + j] += old[j][ni];
                                }
                                if constexpr (BST) {   // land the z loads in front of the stores
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
                                        if (pok[j] && co < a.c_out) {
                                            const float v = acc[mi][ni][r0 + j];
                                            rcf_st1<SO>(a.out, pbase[j] + co, v);
                                            if constexpr (BST) {   // sum g and sum g * z of the gradient the tensor holds, in fp64
                                                const float zz = zv[j][ni];
                                                const float g = v * rcf_lrelu_grad(zz * bk[0][ni] + bk[1][ni]);
                                                st1[ni] += (double)g;
                                                st2[ni] += (double)g * (double)zz;
                                            } else if (want_stats) {   // fp64 per value: E[x^2] - mean^2 must not depend on the tiling
                                                const double dv = (double)v;
                                                st1[ni] += dv;
                                                st2[ni] += dv * dv;
                                            }
                                        }
                                    }
                            }
                        }
                    }
                } else {
                    fetch_a(ky + 1, 0, 0, cur);   // next row's first A operands: the tile does not change inside a chunk
                }
                RCF_T(t_c2);
                RCF_TACC(1, t_c2, t_c1);   // 1: epilogue / next row's first A reads
                if (!(RCF_WS_VARIANT & 2)) rcf_wait_dma();    // the next weight piece has landed
                if (!(RCF_WS_VARIANT & 4) || last_row) __syncthreads();   // ... and is published; after the last row so is the next A tile (written by the producers)
                RCF_T(t_c3);
                RCF_TACC(2, t_c3, t_c2);   // 2: DMA wait + the row's barrier
                if (last_row && more) fetch_a(0, 0, 0, cur ^ 1);
                pb ^= 1;
            }
            cur ^= 1;
            tile = ntile;
            q = nq;
            cb_cur = cb_next;
        }
#ifdef RCF_PHASE_TIMING
        tacc[3] = __builtin_amdgcn_s_memtime() - t_begin;   // 3: whole consumer wave
        if (lane == 0)
            for (int i = 0; i < 4; ++i) atomicAdd(&rcf_phase_cycles[i], tacc[i]);
#endif
#endif
    }

    if (a.stats != nullptr) {
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem_b);   // [4 consumer waves][BN][2]
        if (wave < 4) {
#pragma unroll
            for (int ni = 0; ni < C::NT; ++ni) {
                const double t1 = st1[ni] + __shfl_xor(st1[ni], 32);
                const double t2 = st2[ni] + __shfl_xor(st2[ni], 32);
                if (lh == 0) {
                    red[(wave * C::BN + ni * 32 + li) * 2 + 0] = t1;
                    red[(wave * C::BN + ni * 32 + li) * 2 + 1] = t2;
                }
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
                // BST: sum g * xhat = invstd * (sum g * z - mean * sum g), formed in fp64 from this workgroup's fp64 sums
                if constexpr (BST) t2 = (double)a.bk[3 * a.c_out + co] * (t2 - (double)a.bk[2 * a.c_out + co] * t1);
                a.stats[((size_t)blockIdx.x * 2 + 0) * a.c_out + co] = t1;
                a.stats[((size_t)blockIdx.x * 2 + 1) * a.c_out + co] = t2;
            }
        }
    }
}
