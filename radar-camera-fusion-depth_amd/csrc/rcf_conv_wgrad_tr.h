// Weight gradient on the 16-bit matrix pipe, round-6 form: tiles staged AS THEY ARE (NHWC, channel-contiguous), MFMA operands
// fetched with gfx950's transposing LDS read, staging and matrix work on DIFFERENT waves.  Included by rcf_conv_impl.h (inside its
// anonymous namespace, after conv_wgrad_split_kernel, whose arithmetic -- operand planes, partial products, MFMA order per tile row,
// tile walk, workspace layout -- it keeps: same tiling => bitwise the same partial sums).
//
// What conv_wgrad_split_kernel paid for (tools/phase_timing_wgrad.py, round 5): one wave per SIMD did everything, so the 32-58 % of
// its time spent on "address arithmetic + load issue" and "wait global loads + transpose into LDS" was time the matrix pipe idled;
// the channel-major LDS image (8 consecutive pixels of a channel = one ds_read_b128) forced an 8 pixel x 4 channel register transpose
// per staging unit and 8-byte global loads for bf16 tensors.
//
// Here:
//  * LDS image = [plane][32-channel block][halo pixel][32 ch x 2 B]: a pixel's 32 channels are 64 contiguous bytes, exactly as in
//    HBM (bf16 tensors) or after the plane split of four fp32 channels (8 B per plane: one ds_write_b64, no lane permutation).
//  * ds_read_b64_tr_b16: each 16-lane group reads a [4 pixels][16 channels] block -- lane i supplies the 8-byte address of pixel
//    i / 4, channel quad i % 4 and receives channel i of the four pixels.  Two reads give a lane the 8 consecutive pixels of ITS
//    channel that v_mfma_f32_32x32x16 wants; the 32 lanes of a half read 4 pixels x 64 B = 256 contiguous bytes: all 64 banks once.
//    The lane -> pixel / K-slot assignment is the one the channel-major kernel had, so the kx = 1 operand is still built with
//    v_alignbit from the aligned 8 pixels + the next two, and the halo rows still roll through a register ring.
//  * 512 threads: waves 0-3 are CONSUMERS (the old MFMA loop, nothing else: 144 accumulators + operands), waves 4-7 are
//    PRODUCERS that stage tile t + 1 into the other LDS buffer while tile t is multiplied -- ONE barrier per tile.  bf16 tensors:
//    the producers only issue LDS-DMA (buffer_load ... lds, 16 B per lane, straight into the image).  fp32 tensors: buffer loads of
//    four channels, scale, split into fp16 planes, ds_write_b64 per plane.
//  * addresses: a thread's units keep their place in the tile, so an interior tile (all of it inside one image) costs NO vector
//    arithmetic per load: per-thread constant offset (VGPR) + per-tile / per-round scalar offset (the instruction's soffset).  Only
//    tiles that touch a border, a separator row of the virtual tall image or a ragged channel chunk take the per-unit general path;
//    outside elements get offset 0xffffffff, which the buffer unit answers with zeros (also through LDS-DMA).
#pragma once

// Diagnostics build only (-DRCF_WGRAD_DIAG, tools/probe/wgrad_diag.py; WRONG results): the environment variable RCF_WGRAD_DIAG = 1 drops
// the per-workgroup partial write, 2 the staging after the first tile (consumers alone), 3 the MFMA steps (producers alone).  Round 6,
// fp32 tensors, 64 -> 64 @225x400: whole kernel 152 us, consumers alone 128, producers alone 93, partial write ~3: the matrix loop
// itself -- 216 MFMAs, 80 transposing reads and 96 operand shifts per tile on one wave per SIMD -- is what bounds the kernel.
#ifdef RCF_WGRAD_DIAG
#define RCF_WDIAG(cond) (cond)
#else
#define RCF_WDIAG(cond) true
#endif
typedef short rcf_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 rcf_lds_tr16(const unsigned char* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rcf_s16x4*)p));
}

template <int WCI_, int WCO_, int KS_, int TH_, int NPL_>
struct WtCfg {
    static_assert(NPL_ == 1 || NPL_ == 2, "one bf16 plane or two fp16 planes (the three-plane tier keeps conv_wgrad_split_kernel: 264 registers)");
    static constexpr int KS = KS_, T = KS_ * KS_;
    static constexpr int NPL = NPL_, NP = NPL_ == 2 ? 3 : 1;
    static constexpr int WCI = WCI_, WCO = WCO_, KSPLIT = 4 / (WCI_ * WCO_);
    static constexpr int NCI = 32 * WCI_, NCO = 32 * WCO_;
    static constexpr int PX = 16, TH = TH_, HXP = PX + KS_ - 1, HYP = TH_ + KS_ - 1;
    static constexpr int NXP = HXP * HYP, NDPX = PX * TH_;     // pixels of the x halo tile / the dz tile
    static constexpr int NS = TH_ / KSPLIT;                    // MFMA steps (tile rows) per consumer wave and tile
    // bytes of one 32-channel block of a plane.  fp32 tensors (ds_write_b64 from 16-lane groups that straddle two blocks): block
    // pitch == 64 B (mod 128 B) keeps the two blocks on different banks.  bf16 tensors (LDS-DMA, 1 KiB = 16 pixels per
    // instruction): whole KiB, the ragged last instruction writes zeros into the slack.
    template <bool B16> static constexpr int xblk() { return B16 ? ((NXP + 15) / 16) * 1024 : NXP * 64 + (NXP % 2 == 0 ? 64 : 0); }
    template <bool B16> static constexpr int dblk() { return B16 ? NDPX * 64 : NDPX * 64 + 64; }
    template <bool B16> static constexpr int buf_bytes() { return NPL * (WCI * xblk<B16>() + WCO * dblk<B16>()); }
    static constexpr int RED_BYTES = (KSPLIT > 1) ? WCI * WCO * T * 16 * 64 * 4 : 0;
    template <bool B16> static constexpr int lds_bytes() { return 2 * buf_bytes<B16>() > RED_BYTES ? 2 * buf_bytes<B16>() : RED_BYTES; }
    static_assert(lds_bytes<false>() <= 160 * 1024 && lds_bytes<true>() <= 160 * 1024, "two tile buffers per CU");
};

template <class C, class SX = SAct, class SD = SAct>
__global__ void __launch_bounds__(512, 1) conv_wgrad_tr_kernel(ConvArgs a) {
    constexpr bool B16 = SX::B16;
    static_assert(SX::B16 == SD::B16, "x and dz share the storage type");
    static_assert(!B16 || C::NPL == 1, "bf16 tensors go with bf16 operands");
    constexpr int XBLK = C::template xblk<B16>(), DBLK = C::template dblk<B16>();
    constexpr int XPL = C::WCI * XBLK, DPL = C::WCO * DBLK;          // bytes per operand plane
    constexpr int XBYTES = C::NPL * XPL, BUF = C::template buf_bytes<B16>();
    constexpr int KS = C::KS, HXP = C::HXP, HYP = C::HYP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int cw = wave & 3;

    const int Q = blockIdx.y;
    const bool first = Q < a.nchunk1;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(first ? a.in1 : a.in2);
    const int csrc = first ? a.c1 : a.c2;
    const int cb = (first ? Q : Q - a.nchunk1) * C::NCI;   // first channel of this chunk inside its source
    const int hs = first ? a.h1 : a.h_in;
    const int ws = first ? a.w1 : a.w_in;
    const int gmode = first ? a.gather1 : RCF_GATHER_DIRECT;
    const bool s2 = gmode == RCF_GATHER_STRIDED2;
    const bool nearest = gmode == RCF_GATHER_NEAREST;   // F.interpolate's floor(dst * in / out): general path only
    const int co0 = blockIdx.z * C::NCO;
    // phase_sum == 1: the four phase weight gradients of a 3x3 stride-2 convolution in ONE launch -- workgroup (slot, phase) gathers
    // x at (2y + a, 2x + b) against the SAME dz tile, the four phases of a slot on the same XCD (block b -> XCD b % 8) sharing it in
    // that L2; each phase keeps its own workspace row block (conv_wgrad_split_kernel's decode)
    int slot = blockIdx.x, nslot = gridDim.x, wslot = blockIdx.x;
    int ph_ioy = a.ioy, ph_iox = a.iox;
    if (a.phase_sum == 1) {
        nslot = gridDim.x >> 2;
        int ph;
        if ((nslot & 7) == 0) { ph = (blockIdx.x >> 3) & 3; slot = (blockIdx.x & 7) + 8 * (blockIdx.x >> 5); }
        else { ph = blockIdx.x & 3; slot = blockIdx.x >> 2; }
        ph_ioy = ph >> 1; ph_iox = ph & 1;
        wslot = ph * nslot + slot;
    }

#ifdef RCF_PHASE_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif

    // ================================================================ producers: tile -> LDS buffer
    // geometry shared by both staging forms (wave-uniform)
    const int pm = s2 ? 2 : 1;                                      // source pixels per logical pixel
    const int g_ioy = s2 ? ph_ioy : 0, g_iox = s2 ? ph_iox : 0;
    const unsigned pixbx = (unsigned)csrc * SX::BYTES, rowbx = (unsigned)ws * pixbx;
    const unsigned pixbd = (unsigned)a.c_out * SD::BYTES, rowbd = (unsigned)a.owp * pixbd;
    int wlx = a.w_in;                                               // logical columns that exist in the source
    if (s2) { const int wph = (ws - g_iox + 1) / 2; wlx = wlx < wph ? wlx : wph; }
    int wld = a.w_out;
    { const int wph = (a.owp - a.oox + a.os - 1) / a.os; wld = wld < wph ? wld : wph; }
    const unsigned cbb = (unsigned)cb * SX::BYTES, cob = (unsigned)co0 * SD::BYTES;

    struct TileGeo { int img, oy0, ox0, iy0, ix0; };
    auto geo = [&](int tile) {
        int t = tile;
        const int tx = t % a.tiles_x;
        t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        TileGeo g;
        g.img = t / a.tiles_y;
        g.oy0 = ty * C::TH; g.ox0 = tx * C::PX;
        g.iy0 = g.oy0 - a.pad; g.ix0 = g.ox0 - a.pad_x;
        return g;
    };
    // general path: byte offset of logical input pixel (ly, lx) from the start of image fimg, or 0xffffffff (reads zeros)
    auto xoff_general = [&](const TileGeo& g, int fimg, int hy, int hx, bool ok, unsigned chb) -> unsigned {
        const int ly = g.iy0 + hy, lx = g.ix0 + hx;
        ok = ok && (unsigned)lx < (unsigned)wlx;
        int im = g.img, y = ly;
        if (a.vt) {
            im = (int)(((float)ly + 0.5f) * a.inv_hp);
            y = ly - im * a.hp;
            ok = ok && ly >= 0 && im < a.nimg && y < a.h_in;
        } else {
            ok = ok && (unsigned)ly < (unsigned)a.h_in;
        }
        int py = pm * y + g_ioy, px = pm * lx + g_iox;
        if (nearest) {   // (never with the virtual tall image: select_wgrad)
            py = min((int)floorf((float)ly * a.sy), hs - 1);
            px = min((int)floorf((float)lx * a.sx), ws - 1);
        }
        ok = ok && py < hs;
        return ok ? (unsigned)((im - fimg) * hs + py) * rowbx + (unsigned)px * pixbx + chb : 0xffffffffu;
    };
    auto doff_general = [&](const TileGeo& g, int fimg, int r, int x, bool ok, unsigned chb) -> unsigned {
        int oy = g.oy0 + r, im = g.img;
        if (a.vt) {
            im = (int)(((float)oy + 0.5f) * a.inv_hp);
            oy -= im * a.hp;
            ok = ok && im < a.nimg;
        }
        const int py = oy * a.os + a.ooy;
        const int ox = g.ox0 + x;
        ok = ok && oy < a.h_out && py < a.ohp && (unsigned)ox < (unsigned)wld;
        return ok ? (unsigned)((im - fimg) * a.ohp + py) * rowbd + (unsigned)(ox * a.os + a.oox) * pixbd + chb : 0xffffffffu;
    };
    // first image of a tile's descriptor (wave-uniform) and whether the whole tile lies inside it
    auto x_first_img = [&](const TileGeo& g) {
        int fimg = g.img;
        if (a.vt) {
            fimg = (int)(((float)(g.iy0 < 0 ? 0 : g.iy0) + 0.5f) * a.inv_hp);
            fimg = fimg < a.nimg ? fimg : a.nimg - 1;
        }
        return __builtin_amdgcn_readfirstlane(fimg);
    };
    auto d_first_img = [&](const TileGeo& g) {
        int fimg = g.img;
        if (a.vt) {
            fimg = (int)(((float)g.oy0 + 0.5f) * a.inv_hp);
            fimg = fimg < a.nimg ? fimg : a.nimg - 1;
        }
        return __builtin_amdgcn_readfirstlane(fimg);
    };
    auto x_interior = [&](const TileGeo& g, int fimg, int& y0) {   // y0: first halo row inside image fimg
        y0 = a.vt ? g.iy0 - fimg * a.hp : g.iy0;
        return !nearest && y0 >= 0 && y0 + HYP <= a.h_in && pm * (y0 + HYP - 1) + g_ioy < hs && g.ix0 >= 0 && g.ix0 + HXP <= wlx;
    };
    auto d_interior = [&](const TileGeo& g, int fimg, int& y0) {
        y0 = a.vt ? g.oy0 - fimg * a.hp : g.oy0;
        return y0 + C::TH <= a.h_out && (y0 + C::TH - 1) * a.os + a.ooy < a.ohp && g.ox0 + C::PX <= wld;
    };

    SplitScales sc = {1.f, 1.f, 1.f, 1.f};   // NPL == 2: scales of x (A operand) and dz (B operand)
    if constexpr (C::NPL == 2) sc = rcf_split_scales(a.amax_a1, a.amax_a2, a.amax_b);

    const int ptid = tid & 255;
    const int pw = cw;
    // ---- bf16 tensors: LDS-DMA.  Instruction k of a tile covers 16 consecutive LDS pixels (64 slots of 16 B = 8 channels) of block
    // k / IPB; producer wave pw issues k = pw, pw + 4, ...
    constexpr int IPBX = XBLK / 1024, NIX = C::WCI * IPBX, NJX = (NIX + 3) / 4;
    constexpr int IPBD = DBLK / 1024, NID = C::WCO * IPBD, NJD = (NID + 3) / 4;
    // ---- fp32 tensors: units of one pixel x four channels.  Main rounds: the 16 leftmost halo columns, RPR rows per round (a
    // thread keeps its column and channel quad: round i is round 0 + i * RPR rows); extra rounds: the KS - 1 rightmost columns
    constexpr int CQ = 8 * C::WCI, RPR = 256 / (CQ * 16), NMX = (HYP + RPR - 1) / RPR;
    constexpr int NEX = (HYP * (KS - 1) * CQ + 255) / 256;
    constexpr int CQD = 8 * C::WCO, RPRD = 256 / (CQD * 16), NMD = C::TH / RPRD;
    static_assert(C::TH % RPRD == 0, "whole rounds of dz rows");

    if (producer) {
        if constexpr (B16) {
            // per-lane constants of this wave's instructions
            unsigned kx[NJX], kd[NJD];   // (the general path recomputes a unit's place in the tile from k and the lane: rare, and 14 registers fewer)
#pragma unroll
            for (int j = 0; j < NJX; ++j) {
                const int k = pw + 4 * j, blk = k / IPBX, kk = k % IPBX;
                const int pix = 16 * kk + (lane >> 2), q = lane & 3;
                const int hy = pix / HXP, hx = pix - hy * HXP;
                const bool ok = k < NIX && pix < C::NXP && cb + blk * 32 + q * 8 < csrc;
                kx[j] = ok ? (unsigned)(pm * hy) * rowbx + (unsigned)(pm * hx) * pixbx + (unsigned)(blk * 32 + q * 8) * 2u : 0xffffffffu;
            }
#pragma unroll
            for (int j = 0; j < NJD; ++j) {
                const int k = pw + 4 * j, blk = k / IPBD, kk = k % IPBD;
                const int x = lane >> 2, q = lane & 3;
                const bool ok = k < NID && co0 + blk * 32 + q * 8 < a.c_out;
                kd[j] = ok ? (unsigned)(a.os * kk) * rowbd + (unsigned)(a.os * x) * pixbd + (unsigned)(blk * 32 + q * 8) * 2u : 0xffffffffu;
            }
            auto stage = [&](int tile, int buf) {
                const TileGeo g = geo(tile);
                unsigned char* Xb = smem_b + buf * BUF;
                unsigned char* Db = Xb + XBYTES;
                {
                    const int fimg = x_first_img(g);
                    const __amdgpu_buffer_rsrc_t rs = rcf_rsrc(src + (size_t)fimg * hs * rowbx);
                    int y0;
                    if (x_interior(g, fimg, y0)) {
                        const unsigned so = (unsigned)(pm * y0 + g_ioy) * rowbx + (unsigned)(pm * g.ix0 + g_iox) * pixbx + cbb;
#pragma unroll
                        for (int j = 0; j < NJX; ++j) {
                            const int k = pw + 4 * j;
                            if (k < NIX) rcf_buffer_to_lds16(rs, Xb + (k / IPBX) * XBLK + (k % IPBX) * 1024, kx[j], so);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < NJX; ++j) {
                            const int k = pw + 4 * j;
                            const int pix = 16 * (k % IPBX) + (lane >> 2), hy = pix / HXP, hx = pix - hy * HXP;
                            const unsigned chb = (unsigned)((k / IPBX) * 32 + (lane & 3) * 8) * 2u;
                            const unsigned vo = xoff_general(g, fimg, hy, hx, kx[j] != 0xffffffffu, chb);
                            if (k < NIX) rcf_buffer_to_lds16(rs, Xb + (k / IPBX) * XBLK + (k % IPBX) * 1024, vo, cbb);
                        }
                    }
                }
                {
                    const int fimg = d_first_img(g);
                    const __amdgpu_buffer_rsrc_t rs = rcf_rsrc(reinterpret_cast<const unsigned char*>(a.dz) + (size_t)fimg * a.ohp * rowbd);
                    int y0;
                    if (d_interior(g, fimg, y0)) {
                        const unsigned so = (unsigned)(y0 * a.os + a.ooy) * rowbd + (unsigned)(g.ox0 * a.os + a.oox) * pixbd + cob;
#pragma unroll
                        for (int j = 0; j < NJD; ++j) {
                            const int k = pw + 4 * j;
                            if (k < NID) rcf_buffer_to_lds16(rs, Db + (k / IPBD) * DBLK + (k % IPBD) * 1024, kd[j], so);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < NJD; ++j) {
                            const int k = pw + 4 * j;
                            const unsigned chb = (unsigned)((k / IPBD) * 32 + (lane & 3) * 8) * 2u;
                            const unsigned vo = doff_general(g, fimg, k % IPBD, lane >> 2, kd[j] != 0xffffffffu, chb);
                            if (k < NID) rcf_buffer_to_lds16(rs, Db + (k / IPBD) * DBLK + (k % IPBD) * 1024, vo, cob);
                        }
                    }
                }
            };
            int tile = slot;
            if (tile < a.ntiles) stage(tile, 0);
            rcf_wait_dma();
            __syncthreads();
            int buf = 0;
            while (tile < a.ntiles) {
                RCF_T(t_p0);
                const int ntile = tile + nslot;
                if (ntile < a.ntiles && RCF_WDIAG(a.xcd_band != 78)) stage(ntile, buf ^ 1);
                RCF_T(t_p1);
                RCF_TACC(1, t_p1, t_p0);   // 1: producer: address arithmetic + DMA issue
                rcf_wait_dma();
                __syncthreads();
                RCF_T(t_p2);
                RCF_TACC(2, t_p2, t_p1);   // 2: producer: DMA wait + barrier (idle while the consumers multiply)
                tile = ntile;
                buf ^= 1;
            }
        } else {
            // ---- per-thread constants
            const int cq = ptid % CQ, pp = ptid / CQ, prow = pp >> 4, pcol = pp & 15;
            const bool cvx = cb + cq * 4 < csrc;
            const unsigned chbx = (unsigned)(cq * 4) * SX::BYTES;
            const unsigned kx0 = cvx ? (unsigned)(pm * prow) * rowbx + (unsigned)(pm * pcol) * pixbx + chbx : 0xffffffffu;
            const int ldsx0 = (cq >> 3) * XBLK + (prow * HXP + pcol) * 64 + (cq & 7) * 8;
            unsigned kxe[NEX > 0 ? NEX : 1];
            int ldse[NEX > 0 ? NEX : 1], phe[NEX > 0 ? NEX : 1];
#pragma unroll
            for (int j = 0; j < NEX; ++j) {
                const int pe = (ptid + 256 * j) / CQ;
                const int rowe = pe / (KS > 1 ? KS - 1 : 1), cole = 16 + pe % (KS > 1 ? KS - 1 : 1);
                const bool ok = rowe < HYP && cvx;
                kxe[j] = ok ? (unsigned)(pm * rowe) * rowbx + (unsigned)(pm * cole) * pixbx + chbx : 0xffffffffu;
                ldse[j] = rowe < HYP ? (cq >> 3) * XBLK + (rowe * HXP + cole) * 64 + (cq & 7) * 8 : -1;
                phe[j] = (rowe << 8) | cole | (ok ? 0 : (int)0x80000000);
            }
            const int cqd = ptid % CQD, ppd = ptid / CQD, prd = ppd >> 4, pcd = ppd & 15;
            const bool cvd = co0 + cqd * 4 < a.c_out;
            const unsigned chbd = (unsigned)(cqd * 4) * SD::BYTES;
            const unsigned kd0 = cvd ? (unsigned)(a.os * prd) * rowbd + (unsigned)(a.os * pcd) * pixbd + chbd : 0xffffffffu;
            const int ldsd0 = (cqd >> 3) * DBLK + (prd * 16 + pcd) * 64 + (cqd & 7) * 8;

            // two register sets: the loads of tile t + 2 are in flight while tile t + 1 is split and written -- a tile's HBM latency
            // (2-3 us under load, about one tile's MFMA time) is hidden behind a whole tile instead of sitting in front of the
            // conversion.  Every issue() is unconditional straight-line code (beyond the last tile it re-reads the workgroup's first
            // one), so the compiler's s_waitcnt before a commit() counts exactly the newer set's loads.
            constexpr int NRX = NMX + NEX;
            f32x4 rxa[NRX], rda[NMD], rxb[NRX], rdb[NMD];
            auto put = [&](unsigned char* dst, int plane_bytes, const f32x4& v, float scale) {
                if constexpr (C::NPL == 1) {   // bf16 operands: round to nearest even
                    u32x2 w;
                    w[0] = rcf_f2b2(v[0], v[1]);
                    w[1] = rcf_f2b2(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(dst) = w;
                } else {
                    const rcf_f16_pair q0 = rcf_f16_planes(v[0] * scale, v[1] * scale);
                    const rcf_f16_pair q1 = rcf_f16_planes(v[2] * scale, v[3] * scale);
                    u32x2 w0, w1;
                    w0[0] = q0.p0; w0[1] = q1.p0;
                    w1[0] = q0.p1; w1[1] = q1.p1;
                    *reinterpret_cast<u32x2*>(dst) = w0;
                    *reinterpret_cast<u32x2*>(dst + plane_bytes) = w1;
                }
            };
            // (offsets first -- the interior / general decision is a wave-uniform branch around ARITHMETIC only -- then all loads of
            // the tile in one basic block: with loads inside the two branches the compiler's wait-count state at the join made the
            // commit of the OLDER set wait for the newer set's loads too, vmcnt(0), and the second register set bought nothing)
            auto issue = [&](int tile_, f32x4 (&rx)[NRX], f32x4 (&rd)[NMD]) __attribute__((always_inline)) {
                const TileGeo g = geo(tile_ < a.ntiles ? tile_ : slot);
                unsigned vox[NRX], vod[NMD];
                unsigned sox, sod, stepx, stepd;
                const int fimgx = x_first_img(g), fimgd = d_first_img(g);
                int y0;
                if (x_interior(g, fimgx, y0)) {
                    sox = (unsigned)(pm * y0 + g_ioy) * rowbx + (unsigned)(pm * g.ix0 + g_iox) * pixbx + cbb;
                    stepx = (unsigned)(RPR * pm) * rowbx;
#pragma unroll
                    for (int i = 0; i < NMX; ++i) {
                        const bool partial = RPR * i + RPR > HYP;   // compile-time: only the last round of an odd HYP
                        vox[i] = (partial && RPR * i + prow >= HYP) ? 0xffffffffu : kx0;
                    }
#pragma unroll
                    for (int j = 0; j < NEX; ++j) vox[NMX + j] = kxe[j];
                } else {
                    sox = cbb;
                    stepx = 0u;
#pragma unroll
                    for (int i = 0; i < NMX; ++i) vox[i] = xoff_general(g, fimgx, RPR * i + prow, pcol, cvx && RPR * i + prow < HYP, chbx);
#pragma unroll
                    for (int j = 0; j < NEX; ++j) vox[NMX + j] = xoff_general(g, fimgx, (phe[j] >> 8) & 0xff, phe[j] & 0xff, phe[j] >= 0, chbx);
                }
                if (d_interior(g, fimgd, y0)) {
                    sod = (unsigned)(y0 * a.os + a.ooy) * rowbd + (unsigned)(g.ox0 * a.os + a.oox) * pixbd + cob;
                    stepd = (unsigned)(RPRD * a.os) * rowbd;
#pragma unroll
                    for (int i = 0; i < NMD; ++i) vod[i] = kd0;
                } else {
                    sod = cob;
                    stepd = 0u;
#pragma unroll
                    for (int i = 0; i < NMD; ++i) vod[i] = doff_general(g, fimgd, RPRD * i + prd, pcd, cvd, chbd);
                }
                const __amdgpu_buffer_rsrc_t rsx = rcf_rsrc(src + (size_t)fimgx * hs * rowbx);
                const __amdgpu_buffer_rsrc_t rsd = rcf_rsrc(reinterpret_cast<const unsigned char*>(a.dz) + (size_t)fimgd * a.ohp * rowbd);
#pragma unroll
                for (int i = 0; i < NMX; ++i) rx[i] = rcf_buffer_load_f32x4(rsx, vox[i], sox + (unsigned)i * stepx);
#pragma unroll
                for (int j = 0; j < NEX; ++j) rx[NMX + j] = rcf_buffer_load_f32x4(rsx, vox[NMX + j], sox);
#pragma unroll
                for (int i = 0; i < NMD; ++i) rd[i] = rcf_buffer_load_f32x4(rsd, vod[i], sod + (unsigned)i * stepd);
            };
            auto commit = [&](int buf, const f32x4 (&rx)[NRX], const f32x4 (&rd)[NMD]) __attribute__((always_inline)) {
                unsigned char* Xb = smem_b + buf * BUF;
                unsigned char* Db = Xb + XBYTES;
#pragma unroll
                for (int i = 0; i < NMX; ++i)
                    if (RPR * i + RPR <= HYP || RPR * i + prow < HYP) put(Xb + ldsx0 + i * RPR * HXP * 64, XPL, rx[i], sc.sa);
#pragma unroll
                for (int j = 0; j < NEX; ++j)
                    if (ldse[j] >= 0) put(Xb + ldse[j], XPL, rx[NMX + j], sc.sa);
#pragma unroll
                for (int i = 0; i < NMD; ++i) put(Db + ldsd0 + i * RPRD * 1024, DPL, rd[i], sc.sb);
            };
            int tile = slot;
            issue(tile, rxa, rda);
            issue(tile + nslot, rxb, rdb);
            commit(0, rxa, rda);
            __syncthreads();
            int buf = 0;
            // iteration of tile t (the consumers multiply it from `buf`): loads of t + 2 into the set committed last time, then tile
            // t + 1 (loaded a whole iteration ago) -> the other buffer.  Unrolled by two so that the register sets are static.
            while (tile < a.ntiles) {
                {
                    RCF_T(t_p0);
                    if (RCF_WDIAG(a.xcd_band != 78)) issue(tile + 2 * nslot, rxa, rda);
                    if (tile + nslot < a.ntiles && RCF_WDIAG(a.xcd_band != 78)) commit(buf ^ 1, rxb, rdb);
                    RCF_T(t_p1);
                    RCF_TACC(1, t_p1, t_p0);   // 1: producer: load issue of tile t + 2, plane split + LDS writes of tile t + 1
                    __syncthreads();
                    RCF_T(t_p2);
                    RCF_TACC(2, t_p2, t_p1);   // 2: producer: barrier (idle while the consumers multiply)
                    tile += nslot;
                    buf ^= 1;
                }
                if (tile >= a.ntiles) break;
                {
                    RCF_T(t_p0);
                    if (RCF_WDIAG(a.xcd_band != 78)) issue(tile + 2 * nslot, rxb, rdb);
                    if (tile + nslot < a.ntiles && RCF_WDIAG(a.xcd_band != 78)) commit(buf ^ 1, rxa, rda);
                    RCF_T(t_p1);
                    RCF_TACC(1, t_p1, t_p0);
                    __syncthreads();
                    RCF_T(t_p2);
                    RCF_TACC(2, t_p2, t_p1);
                    tile += nslot;
                    buf ^= 1;
                }
            }
        }
        // the consumers' slice reduction below passes KSPLIT - 1 barrier pairs
        if (C::KSPLIT > 1)
            for (int s = C::KSPLIT - 1; s >= 1; --s) { __syncthreads(); __syncthreads(); }
#ifdef RCF_PHASE_TIMING
        tacc[6] = __builtin_amdgcn_s_memtime() - t_begin;   // 6: producer wave, whole
        if (lane == 0) { atomicAdd(&rcf_phase_cycles[1], tacc[1]); atomicAdd(&rcf_phase_cycles[2], tacc[2]); atomicAdd(&rcf_phase_cycles[6], tacc[6]); }
#endif
        return;
    }

    // ================================================================ consumers: the MFMA loop of conv_wgrad_split_kernel on tr reads
    const int li = lane & 31, lh = lane >> 5;
    const int wi = cw % C::WCI, wj = (cw / C::WCI) % C::WCO, wk = cw / (C::WCI * C::WCO);
    // lane (16-lane group g = lane >> 4, i = lane & 15) supplies pixel 8 * lh + i / 4 (+ 4, + 8 by immediate), channels 16 * (g & 1) + 4 * (i % 4) ...
    const int tr_off = (8 * lh + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    const unsigned char* xb0 = smem_b + wi * XBLK + tr_off + wk * (HXP * 64);
    const unsigned char* db0 = smem_b + XBYTES + wj * DBLK + tr_off + wk * (C::PX * 64);

    f32x16 acc[C::T];
#pragma unroll
    for (int tap = 0; tap < C::T; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tap][r] = 0.f;

    constexpr int ROLL = C::KSPLIT == 1;
    constexpr int NSLOT = ROLL ? KS + 1 : 2 * KS;
    constexpr int NEWROWS = ROLL ? 1 : KS;          // halo rows fetched per step
    constexpr int NPL = C::NPL, NP = C::NP;
    constexpr int XRD = KS > 1 ? 3 : 2;             // tr reads per halo row and plane: pixels 0-3, 4-7 (, 8-11: two of them used)
    constexpr int NRD = NEWROWS * XRD * NPL + 2 * NPL;    // LDS reads per step
    constexpr int NMF = NP * C::T;                  // MFMAs per step
    u32x4 xlo[NSLOT][NPL];     // [row slot][plane]  pixels 8h .. 8h+7
    unsigned xhi[NSLOT][NPL];  //                    pixels 8h+8, 8h+9
    u32x4 dzv[2][NPL];         // [set][plane]
    u32x4 xs1[KS][NPL];        // kx = 1 operands of the current step

    auto rd_x = [&](const unsigned char* xb, int slot_, int pl, int row, int sub) __attribute__((always_inline)) {
        const u32x2 v = rcf_lds_tr16(xb + pl * XPL + row * (HXP * 64) + sub * 256);
        if (sub == 0) { xlo[slot_][pl][0] = v[0]; xlo[slot_][pl][1] = v[1]; }
        else if (sub == 1) { xlo[slot_][pl][2] = v[0]; xlo[slot_][pl][3] = v[1]; }
        else xhi[slot_][pl] = v[0];
    };
    auto rd_d = [&](const unsigned char* db, int set, int pl, int row, int sub) __attribute__((always_inline)) {
        const u32x2 v = rcf_lds_tr16(db + pl * DPL + row * (C::PX * 64) + sub * 256);
        if (sub == 0) { dzv[set][pl][0] = v[0]; dzv[set][pl][1] = v[1]; }
        else { dzv[set][pl][2] = v[0]; dzv[set][pl][3] = v[1]; }
    };

    int tile = slot;
    int buf = 0;
    __syncthreads();   // buffer 0 is full
    while (tile < a.ntiles) {
        RCF_T(t_g0);
        const unsigned char* xb = xb0 + buf * BUF;
        const unsigned char* db = db0 + buf * BUF;
        // prologue of the tile: operands of this wave's first row
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int sub = 0; sub < XRD; ++sub) rd_x(xb, ky, pl, ky, sub);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) { rd_d(db, 0, pl, 0, 0); rd_d(db, 0, pl, 0, 1); }
        __builtin_amdgcn_sched_barrier(0);

        if (RCF_WDIAG(a.xcd_band != 79))
#pragma unroll
        for (int s = 0; s < C::NS; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            const bool has_next = s + 1 < C::NS;
            const int rn = (s + 1) * C::KSPLIT;   // next row of this wave (relative to wk)
#pragma unroll
            for (int j = 0; j < NMF; ++j) {
                // order: all kx = 0 taps, (kx = 2,) then kx = 1, whose operands are being built meanwhile; inside a group the
                // partial products run smallest first and the kernel rows alternate (conv_wgrad_split_kernel's order)
                constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};
                constexpr int KXO[3] = {0, KS == 3 ? 2 : 1, 1};
                const int grp = j / (NP * KS), pj = (j % (NP * KS)) / KS, ky = j % KS;
                const int kx = KXO[grp];
                const int tap = ky * KS + kx;
                const int sl = ROLL ? (s + ky) % (KS + 1) : cur * KS + ky;
                const int pa = NPL == 2 ? PA2[pj % 3] : 0, pbl = NPL == 2 ? PB2[pj % 3] : 0;
                u32x4 av;
                if (kx == 0) av = xlo[sl][pa];
                else if (kx == 1) av = xs1[ky][pa];
                else {
                    const u32x4 lo = xlo[sl][pa];
                    av[0] = lo[1]; av[1] = lo[2]; av[2] = lo[3]; av[3] = xhi[sl][pa];
                }
                acc[tap] = rcf_mfma_split<NPL>(as_bf16x8(av), as_bf16x8(dzv[cur][pbl]), acc[tap]);
                // one kx = 1 operand (4 v_alignbit) behind each of the first NPL * KS MFMAs: all of them before the kx = 1 group starts
                if (KS > 1 && j < NPL * KS) {
                    __builtin_amdgcn_sched_barrier(0);
                    const int sky = j / NPL, spl = j % NPL;
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
                    if (nr < NEWROWS * XRD * NPL) {
                        const int rrow = (nr / XRD) / NPL, rpl = (nr / XRD) % NPL, sub = nr % XRD;
                        const int rky = ROLL ? KS - 1 : rrow;
                        const int nsl = ROLL ? (s + KS) % (KS + 1) : nxt * KS + rrow;
                        rd_x(xb, nsl, rpl, rn + rky, sub);
                    } else {
                        const int rdp = nr - NEWROWS * XRD * NPL;   // (plane, half)
                        rd_d(db, nxt, rdp / 2, rn, rdp % 2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        RCF_T(t_g1);
        RCF_TACC(4, t_g1, t_g0);   // 4: MFMA steps (+ interleaved LDS reads, operand shifts)
        __syncthreads();            // the producers have filled the other buffer; everybody is done with this one
        RCF_T(t_g2);
        RCF_TACC(0, t_g2, t_g1);   // 0: barrier (waiting for the producers / the slowest consumer)
        tile += nslot;
        buf ^= 1;
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
                for (int tap = 0; tap < C::T; ++tap) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[tap][r] + red[(tap * 16 + r) * 64 + lane];
                        // the sum is pinned HERE (an empty asm that "uses" the register: no instruction).  Left alone, the adds are sunk
                        // into the partial-write block below (their only use) and all 144 reads end up in flight at once -- beside 144
                        // accumulators that does not fit the 256 registers of a two-waves-per-SIMD kernel (39 spilled VGPRs in the 16-row
                        // bf16 variants)
                        __asm__ volatile("" : "+v"(v));
                        acc[tap][r] = v;
                    }
                    __builtin_amdgcn_sched_barrier(0);   // one tap's 16 reads at a time
                }
            }
        }
    }
    if (wk == 0 && RCF_WDIAG(a.xcd_band != 77)) {
        const int q32_0 = first ? 0 : (a.c1 + 31) / 32;   // 32-channel chunk index of this source's first chunk in k
        const int co = co0 + wj * 32 + li;
        float* wsp = a.ws + (size_t)wslot * a.ktot * a.cop;
#pragma unroll
        for (int tap = 0; tap < C::T; ++tap) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = cb + wi * 32 + rcf_mfma_row(r, lh);   // channel inside its source
                if ((ch & ~31) < csrc && co < a.cop) {
                    const int k = ((q32_0 + (ch >> 5)) * C::T + tap) * 32 + (ch & 31);
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
