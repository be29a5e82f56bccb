// RadarNet stage-1 device ops that FusionNet does not have, NHWC fp32 on gfx950:
//   * ROI max pooling of image features around each radar point -- torchvision.ops.roi_pool as called at
//     src/networks.py:1232-1247 (torchvision 0.11.3 semantics restated: rounded box * spatial_scale, +1 extent,
//     floor/ceil bin edges, max over the bin, -FLT_MAX start, 0 for an empty bin) -- forward with argmax, backward scatter;
//   * net_utils.FullyConnected (src/net_utils.py:201-247): Linear + bias + LeakyReLU(0.2), forward and backward, with an
//     optional feature -> (pixel, channel) placement so the last layer of FullyConnectedEncoder (src/networks.py:1007-1067)
//     writes straight into the NHWC latent next to the pooled image features (the .view(N, C, -1, W) + torch.cat of
//     src/networks.py:1251-1255 never exist as separate tensors);
//   * RadarNetModel.compute_loss (src/radarnet_model.py:131-171): binary_cross_entropy_with_logits with pos_weight,
//     masked by the validity map, sum / sum(validity).
// All HBM / latency bound and small next to the encoder and decoder convolutions, which reuse the FusionNet kernels.
#include <float.h>
#include "rcf_common.h"

namespace {

// ---------------------------------------------------------------- ROI max pooling
// One thread per (roi, ph, pw, channel quad).  in: (N, H, W, C); rois: (R, 5) = (batch index, x1, y1, x2, y2);
// out: (R, PH, PW, out_cstride) at channel offset out_coff; argmax: (R, PH, PW, C) int32 = y * W + x of the maximum (-1: empty).
template <class S>
__global__ void __launch_bounds__(256) roi_pool_fwd_kernel(const float* __restrict__ in, const float* __restrict__ rois,
                                                           float* __restrict__ out, int* __restrict__ argmax, int n_roi, int h, int w,
                                                           int c, int ph_n, int pw_n, float scale, int out_cstride, int out_coff) {
    const int c4n = c >> 2;
    const long long total = (long long)n_roi * ph_n * pw_n * c4n;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        long long t = g;
        const int cg = (int)(t % c4n); t /= c4n;
        const int pw = (int)(t % pw_n); t /= pw_n;
        const int ph = (int)(t % ph_n);
        const int r = (int)(t / ph_n);
        const float* roi = rois + (size_t)r * 5;
        const int b = (int)roi[0];
        const int x0 = (int)roundf(roi[1] * scale), y0 = (int)roundf(roi[2] * scale);
        const int x1 = (int)roundf(roi[3] * scale), y1 = (int)roundf(roi[4] * scale);
        const int rw = max(x1 - x0 + 1, 1), rh = max(y1 - y0 + 1, 1);
        const float bh = (float)rh / (float)ph_n, bw = (float)rw / (float)pw_n;
        int hs = (int)floorf((float)ph * bh), he = (int)ceilf((float)(ph + 1) * bh);
        int ws = (int)floorf((float)pw * bw), we = (int)ceilf((float)(pw + 1) * bw);
        hs = min(max(hs + y0, 0), h); he = min(max(he + y0, 0), h);
        ws = min(max(ws + x0, 0), w); we = min(max(we + x0, 0), w);
        const bool empty = he <= hs || we <= ws;
        f32x4 m = empty ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
        int am[4] = {-1, -1, -1, -1};
        for (int y = hs; y < he; ++y)
            for (int x = ws; x < we; ++x) {
                const f32x4 v = rcf_ld4<S>(in, (((size_t)b * h + y) * w + x) * c + cg * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (v[j] > m[j]) { m[j] = v[j]; am[j] = y * w + x; }
            }
        const size_t o = ((size_t)r * ph_n + ph) * pw_n + pw;
        rcf_st4<S>(out, o * out_cstride + out_coff + cg * 4, m);
#pragma unroll
        for (int j = 0; j < 4; ++j) argmax[o * c + cg * 4 + j] = am[j];
    }
}

// din (N, H, W, C) must be zeroed by the caller (or hold an earlier contribution): rois overlap, so this is a scatter-add.
template <class S>
__global__ void __launch_bounds__(256) roi_pool_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ argmax,
                                                           const float* __restrict__ rois, float* __restrict__ din, int n_roi, int h,
                                                           int w, int c, int ph_n, int pw_n, int dout_cstride, int dout_coff) {
    const long long total = (long long)n_roi * ph_n * pw_n * c;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const int ch = (int)(g % c);
        const long long o = g / c;
        const int r = (int)(o / ((long long)ph_n * pw_n));
        const int a = argmax[g];
        if (a < 0) continue;
        const int b = (int)rois[(size_t)r * 5];
        atomicAdd(din + ((size_t)b * h * w + a) * c + ch, rcf_ld1<S>(dout, o * dout_cstride + dout_coff + ch));   // din is fp32 for either storage
    }
}

// The same backward as a GATHER [r4]: one thread per input pixel and four channels walks the rois of its image and the bins that
// contain the pixel (the forward's own bin-edge expressions: a pixel sits in 1 bin per axis when the pooled size equals the roi size,
// in up to ceil(1 / bin) + 1 otherwise), and adds dout where the bin's argmax is this pixel.  No atomics, no zero fill of din, no fp32
// detour for bf16 gradients (the scatter above moved 3.6 GB per level-1 call of the RadarNet step, 0.74 ms; this form reads dout and
// argmax once and writes din once), and a fixed summation order (roi, bin row, bin column).
constexpr int ROI_GATHER_MAX = 1024;   // rois the gather form keeps in LDS (more: use the scatter form)

template <class S>
__global__ void __launch_bounds__(256) roi_pool_bwd_gather_kernel(const float* __restrict__ dout, const int* __restrict__ argmax,
                                                                  const float* __restrict__ rois, float* __restrict__ din, int accumulate,
                                                                  int n_roi, int n, int h, int w, int c, int ph_n, int pw_n, float scale,
                                                                  int dout_cstride, int dout_coff) {
    // blockIdx.y = image.  The rois of this image, in roi order, with their integer box and bin sizes: built once per workgroup (a
    // per-thread scan of all rois cost one dependent scalar load per roi and pixel group: 1.2 of the kernel's 1.4 ms at level 1)
    __shared__ int s_flag[ROI_GATHER_MAX];
    __shared__ int s_r[ROI_GATHER_MAX];
    __shared__ int s_box[ROI_GATHER_MAX][4];     // x0, y0, rw, rh
    __shared__ float s_bin[ROI_GATHER_MAX][2];   // bh, bw
    __shared__ int s_cnt;
    const int b = blockIdx.y;
    for (int r = threadIdx.x; r < n_roi; r += 256) s_flag[r] = ((int)rois[(size_t)r * 5] == b) ? 1 : 0;
    __syncthreads();
    for (int r = threadIdx.x; r < n_roi; r += 256) {
        if (!s_flag[r]) continue;
        int pos = 0;
        for (int q = 0; q < r; ++q) pos += s_flag[q];
        const float* roi = rois + (size_t)r * 5;
        const int x0 = (int)roundf(roi[1] * scale), y0 = (int)roundf(roi[2] * scale);
        const int x1 = (int)roundf(roi[3] * scale), y1 = (int)roundf(roi[4] * scale);
        const int rw = max(x1 - x0 + 1, 1), rh = max(y1 - y0 + 1, 1);
        s_r[pos] = r;
        s_box[pos][0] = x0; s_box[pos][1] = y0; s_box[pos][2] = rw; s_box[pos][3] = rh;
        s_bin[pos][0] = (float)rh / (float)ph_n;
        s_bin[pos][1] = (float)rw / (float)pw_n;
    }
    if (threadIdx.x == 0) {
        int cnt = 0;
        for (int q = 0; q < n_roi; ++q) cnt += s_flag[q];
        s_cnt = cnt;
    }
    __syncthreads();
    const int cnt = s_cnt;
    const int c4 = c >> 2;
    const long long total = (long long)h * w * c4;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const int cg = (int)(g % c4);
        const long long t = g / c4;
        const int x = (int)(t % w);
        const int y = (int)(t / w);
        const size_t idx = (((size_t)b * h + y) * w + x) * c + cg * 4;
        f32x4 acc = accumulate ? rcf_ld4<S>(din, idx) : f32x4{0.f, 0.f, 0.f, 0.f};
        const int me = y * w + x;
        for (int i = 0; i < cnt; ++i) {
            const int x0 = s_box[i][0], y0 = s_box[i][1], rw = s_box[i][2], rh = s_box[i][3];
            const int ty = y - y0, tx = x - x0;
            if (ty < 0 || ty > rh || tx < 0 || tx > rw) continue;
            const int r = s_r[i];
            const float bh = s_bin[i][0], bw = s_bin[i][1];
            const int ph_lo = max(0, (int)floorf((float)ty / bh) - 1), ph_hi = min(ph_n - 1, (int)floorf((float)(ty + 1) / bh) + 1);
            const int pw_lo = max(0, (int)floorf((float)tx / bw) - 1), pw_hi = min(pw_n - 1, (int)floorf((float)(tx + 1) / bw) + 1);
            for (int ph = ph_lo; ph <= ph_hi; ++ph) {
                int hs = (int)floorf((float)ph * bh), he = (int)ceilf((float)(ph + 1) * bh);   // roi_pool_fwd_kernel's edges
                hs = min(max(hs + y0, 0), h); he = min(max(he + y0, 0), h);
                if (y < hs || y >= he) continue;
                for (int pw = pw_lo; pw <= pw_hi; ++pw) {
                    int ws = (int)floorf((float)pw * bw), we = (int)ceilf((float)(pw + 1) * bw);
                    ws = min(max(ws + x0, 0), w); we = min(max(we + x0, 0), w);
                    if (x < ws || x >= we) continue;
                    const size_t o = ((size_t)r * ph_n + ph) * pw_n + pw;
                    const int4 am = *reinterpret_cast<const int4*>(argmax + o * c + cg * 4);
                    const f32x4 d = rcf_ld4<S>(dout, o * dout_cstride + dout_coff + cg * 4);
                    if (am.x == me) acc[0] += d[0];
                    if (am.y == me) acc[1] += d[1];
                    if (am.z == me) acc[2] += d[2];
                    if (am.w == me) acc[3] += d[3];
                }
            }
        }
        rcf_st4<S>(din, idx, acc);
    }
}

// ---------------------------------------------------------------- fully connected + bias + LeakyReLU
// y[m][f] = lrelu(sum_k x[m][k] * W[f][k] + b[f]), M <= FC_MAX_M rows held as per-thread accumulators, one thread per output
// feature (W is read exactly once, coalesced over k by the wave).  Output address of (m, f): with hw == 1 plain [m][f]; otherwise
// feature f = c * hw + p goes to NHWC position m * (hw * cstride) + p * cstride + coff + c.
constexpr int FC_MAX_M = 64;

__device__ __forceinline__ size_t fc_addr(int m, int f, int n_out, int hw, int cstride, int coff) {
    if (hw <= 1) return (size_t)m * n_out + f;
    const int cch = f / hw, p = f - cch * hw;
    return ((size_t)m * hw + p) * cstride + coff + cch;
}

// Which feature a thread owns: with the NHWC output layout (hw > 1) consecutive threads take consecutive CHANNELS of one position, so
// that their y / dy elements are neighbours in memory (feature order f = c * hw + p would put them a whole pixel row apart: the last
// layer's backward spent 0.4 ms in 2-byte gathers) [r4]
__device__ __forceinline__ int fc_feature_of_thread(int t, int n_out, int hw) {
    if (hw <= 1) return t;
    const int nc = n_out / hw;
    return (t % nc) * hw + t / nc;
}

template <int MB, class S>
__global__ void __launch_bounds__(256) fc_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                     const float* __restrict__ bias, float* __restrict__ y, int m_total, int n_in,
                                                     int n_out, int act, int hw, int cstride, int coff) {
    // blockIdx.y takes MB rows [r4]: one thread per feature with all 64 rows as accumulators made the small layers of the MLP (n_out
    // 32 ... 128: ONE workgroup) a 230-us serial loop each; per output the fmaf chain over k is the same
    extern __shared__ float xs[];   // [rows][n_in]
    const int m0 = blockIdx.y * MB;
    const int m_rows = min(MB, m_total - m0);
    for (int i = threadIdx.x; i < m_rows * n_in; i += 256) xs[i] = x[(size_t)m0 * n_in + i];
    __syncthreads();
    const int tf = blockIdx.x * 256 + threadIdx.x;
    if (tf >= n_out) return;
    const int f = fc_feature_of_thread(tf, n_out, hw);
    float acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = 0.f;
    const float* wr = wgt + (size_t)f * n_in;
    for (int k = 0; k < n_in; ++k) {
        const float wv = wr[k];
#pragma unroll
        for (int m = 0; m < MB; ++m)
            if (m < m_rows) acc[m] = fmaf(xs[m * n_in + k], wv, acc[m]);
    }
    const float bv = bias[f];
#pragma unroll
    for (int m = 0; m < MB; ++m)
        if (m < m_rows) {
            float v = acc[m] + bv;
            if (act) v = rcf_lrelu(v);
            rcf_st1<S>(y, fc_addr(m0 + m, f, n_out, hw, cstride, coff), v);
        }
}

// Backward: g = dy * lrelu'(y); dW[f][k] = sum_m g[m][f] x[m][k]; db[f] = sum_m g[m][f] (one thread per output feature).
constexpr int FC_KCH = 8;   // input features per workgroup of fc_bwd_kernel (blockIdx.y)

template <int MB, class S>
__global__ void __launch_bounds__(256) fc_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                     const float* __restrict__ dy, float* __restrict__ dw, float* __restrict__ db,
                                                     int m_rows, int n_in, int n_out, int act, int hw, int cstride, int coff,
                                                     int accumulate) {
    // blockIdx.y takes FC_KCH input features [r4]; per dW element the fmaf chain over the rows is the same as with one workgroup per
    // 256 output features looping over all of n_in
    extern __shared__ float xs[];   // [m_rows][FC_KCH]
    const int k0 = blockIdx.y * FC_KCH;
    const int nk = min(FC_KCH, n_in - k0);
    for (int i = threadIdx.x; i < m_rows * FC_KCH; i += 256) {
        const int m = i / FC_KCH, kk = i - m * FC_KCH;
        xs[i] = kk < nk ? x[(size_t)m * n_in + k0 + kk] : 0.f;
    }
    __syncthreads();
    const int tf = blockIdx.x * 256 + threadIdx.x;
    if (tf >= n_out) return;
    const int f = fc_feature_of_thread(tf, n_out, hw);
    float g[MB];
    float gs = 0.f;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        g[m] = 0.f;
        if (m < m_rows) {
            const size_t a = fc_addr(m, f, n_out, hw, cstride, coff);
            g[m] = rcf_ld1<S>(dy, a) * (act ? rcf_lrelu_grad(rcf_ld1<S>(y, a)) : 1.f);
            gs += g[m];
        }
    }
    if (blockIdx.y == 0) db[f] = accumulate ? db[f] + gs : gs;
    for (int kk = 0; kk < nk; ++kk) {
        float dwv = accumulate ? dw[(size_t)f * n_in + k0 + kk] : 0.f;   // later row blocks continue the same fmaf chain
#pragma unroll
        for (int m = 0; m < MB; ++m)
            if (m < m_rows) dwv = fmaf(g[m], xs[m * FC_KCH + kk], dwv);
        dw[(size_t)f * n_in + k0 + kk] = dwv;
    }
}

// Input gradient dx[m][k] = sum_f g[m][f] W[f][k]: a block takes FC_DXF features, keeps their g in LDS and accumulates its partial
// [m][k]; partials are summed (fp64) by fc_dx_reduce_kernel.  64 features per block [r4]: with 256 the last layer's 261 workgroups ran
// 8192 dependent fmaf per thread (0.38 ms); four times the workgroups, a quarter of the chain.
constexpr int FC_DXF = 64;

template <class S>
__global__ void __launch_bounds__(256) fc_dx_partial_kernel(const float* __restrict__ wgt, const float* __restrict__ y,
                                                            const float* __restrict__ dy, float* __restrict__ part, int m_rows,
                                                            int n_in, int n_out, int act, int hw, int cstride, int coff) {
    extern __shared__ float gs[];   // [FC_DXF features][m_rows]
    const int f0 = blockIdx.x * FC_DXF;
    for (int i = threadIdx.x; i < FC_DXF * m_rows; i += 256) {
        const int fl = i / m_rows, m = i - fl * m_rows;
        const int f = f0 + fl;
        float g = 0.f;
        if (f < n_out) {
            const size_t a = fc_addr(m, f, n_out, hw, cstride, coff);
            g = rcf_ld1<S>(dy, a) * (act ? rcf_lrelu_grad(rcf_ld1<S>(y, a)) : 1.f);
        }
        gs[i] = g;
    }
    __syncthreads();
    const int nf = min(FC_DXF, n_out - f0);
    for (int o = threadIdx.x; o < m_rows * n_in; o += 256) {
        const int m = o / n_in, k = o - m * n_in;
        float acc = 0.f;
        for (int fl = 0; fl < nf; ++fl) acc = fmaf(gs[fl * m_rows + m], wgt[(size_t)(f0 + fl) * n_in + k], acc);
        part[(size_t)blockIdx.x * m_rows * n_in + o] = acc;
    }
}

__global__ void __launch_bounds__(256) fc_dx_reduce_kernel(const float* __restrict__ part, float* __restrict__ dx, int n_part, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int p = 0; p < n_part; ++p) s += (double)part[(size_t)p * n + i];
    dx[i] = (float)s;
}

// ---------------------------------------------------------------- masked BCE with logits
constexpr int BCE_BLOCKS = 1024;

__device__ __forceinline__ float bce_term(float x, float t, float pw) {
    // torch.binary_cross_entropy_with_logits with pos_weight: (1 - t) x + (1 + (pw - 1) t) (log1p(exp(-|x|)) + max(-x, 0))
    const float lw = 1.f + (pw - 1.f) * t;
    return (1.f - t) * x + lw * (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f));
}

__global__ void __launch_bounds__(256) bce_partial_kernel(const float* __restrict__ logit, const float* __restrict__ target,
                                                          const float* __restrict__ valid, double* __restrict__ part, long long n,
                                                          float pw) {
    __shared__ double sm[2][4];
    double s0 = 0.0, s1 = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = valid[i];
        s0 += (double)(v * bce_term(logit[i], target[i], pw));
        s1 += (double)v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); }
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = s0; sm[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3];
        part[blockIdx.x * 2 + 1] = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
    }
}

// sums[0] = sum(valid * loss), sums[1] = sum(valid); loss = sums[0] / sums[1]
__global__ void __launch_bounds__(64) bce_final_kernel(const double* __restrict__ part, int nb, double* __restrict__ sums,
                                                       float* __restrict__ loss) {
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < nb; i += 64) { s0 += part[i * 2]; s1 += part[i * 2 + 1]; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); }
    if (threadIdx.x == 0) {
        sums[0] = s0;
        sums[1] = s1;
        loss[0] = (float)(s0 / s1);
    }
}

__global__ void __launch_bounds__(256) bce_bwd_kernel(const float* __restrict__ logit, const float* __restrict__ target,
                                                      const float* __restrict__ valid, const double* __restrict__ sums,
                                                      const float* __restrict__ upstream, float* __restrict__ dlogit, long long n,
                                                      float pw) {
    const float k = upstream[0] / (float)sums[1];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float x = logit[i], t = target[i];
        const float lw = 1.f + (pw - 1.f) * t;
        dlogit[i] = k * valid[i] * ((1.f - t) - lw * (1.f - rcf_sigmoid(x)));
    }
}

unsigned grid_for(long long n, unsigned cap) {
    long long b = (n + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}   // namespace

template <class S>
static int roi_pool_fwd_impl(const float* in, const float* rois, float* out, int* argmax, int n_roi, int n, int h, int w, int c,
                                int pooled_h, int pooled_w, float spatial_scale, int out_cstride, int out_coff, void* stream) {
    if (!in || !rois || !out || !argmax || n_roi <= 0 || n <= 0 || h <= 0 || w <= 0 || pooled_h <= 0 || pooled_w <= 0) return RCF_EINVAL;
    if (c < 4 || (c & 3) || out_cstride < out_coff + c || (out_cstride & 3) || (out_coff & 3)) return RCF_EUNSUPPORTED;
    const long long total = (long long)n_roi * pooled_h * pooled_w * (c >> 2);
    hipLaunchKernelGGL((roi_pool_fwd_kernel<S>), dim3(grid_for(total, 16384)), dim3(256), 0, (hipStream_t)stream, in, rois, out, argmax, n_roi,
                       h, w, c, pooled_h, pooled_w, spatial_scale, out_cstride, out_coff);
    return rcf_launch_status();
}

template <class S>
static int roi_pool_bwd_impl(const float* dout, const int* argmax, const float* rois, float* din, int n_roi, int n, int h, int w,
                                int c, int pooled_h, int pooled_w, int dout_cstride, int dout_coff, void* stream) {
    if (!dout || !argmax || !rois || !din || n_roi <= 0 || n <= 0 || h <= 0 || w <= 0 || pooled_h <= 0 || pooled_w <= 0) return RCF_EINVAL;
    if (c <= 0 || dout_cstride < dout_coff + c) return RCF_EUNSUPPORTED;
    const long long total = (long long)n_roi * pooled_h * pooled_w * c;
    hipLaunchKernelGGL((roi_pool_bwd_kernel<S>), dim3(grid_for(total, 16384)), dim3(256), 0, (hipStream_t)stream, dout, argmax, rois, din,
                       n_roi, h, w, c, pooled_h, pooled_w, dout_cstride, dout_coff);
    return rcf_launch_status();
}

template <class S>
static int roi_pool_bwd_gather_impl(const float* dout, const int* argmax, const float* rois, float* din, int din_accumulate, int n_roi,
                                       int n, int h, int w, int c, int pooled_h, int pooled_w, float spatial_scale, int dout_cstride,
                                       int dout_coff, void* stream) {
    if (!dout || !argmax || !rois || !din || n_roi <= 0 || n <= 0 || h <= 0 || w <= 0 || pooled_h <= 0 || pooled_w <= 0) return RCF_EINVAL;
    if (c < 4 || (c & 3) || dout_cstride < dout_coff + c || (dout_cstride & 3) || (dout_coff & 3) || n_roi > ROI_GATHER_MAX || n > 65535)
        return RCF_EUNSUPPORTED;
    const long long per_image = (long long)h * w * (c >> 2);
    const unsigned gx = grid_for(per_image, (unsigned)(8192 / n > 1 ? 8192 / n : 1));   // a few pixel groups per workgroup: the roi table is built once
    hipLaunchKernelGGL((roi_pool_bwd_gather_kernel<S>), dim3(gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, dout, argmax, rois, din,
                       din_accumulate, n_roi, n, h, w, c, pooled_h, pooled_w, spatial_scale, dout_cstride, dout_coff);
    return rcf_launch_status();
}

// Rows are processed in blocks of at most FC_MAX_M (the per-thread accumulator count): a frame may carry any number of radar
// points (pipeline.radarnet_forward runs every point of a frame through one call, like src/radarnet_main.py:534-561).
template <class S>
static int fc_fwd_impl(const float* x, const float* w, const float* bias, float* y, int m_rows, int n_in, int n_out, int act,
                          int hw, int cstride, int coff, void* stream) {
    if (!x || !w || !bias || !y || m_rows <= 0 || n_in <= 0 || n_out <= 0) return RCF_EINVAL;
    if ((size_t)FC_MAX_M * n_in * 4 > 48 * 1024 || (hw > 1 && (n_out % hw != 0 || cstride < coff + n_out / hw))) return RCF_EUNSUPPORTED;
    const unsigned nb = (unsigned)((n_out + 255) / 256);
    constexpr int ROWS = 16;   // rows per workgroup (blockIdx.y)
    const unsigned ny = (unsigned)((m_rows + ROWS - 1) / ROWS);
    if (ny > 65535u) return RCF_EUNSUPPORTED;
    hipLaunchKernelGGL((fc_fwd_kernel<ROWS, S>), dim3(nb, ny), dim3(256), (size_t)ROWS * n_in * sizeof(float), (hipStream_t)stream, x, w, bias,
                       y, m_rows, n_in, n_out, act, hw, cstride, coff);
    return rcf_launch_status();
}

extern "C" size_t rcf_fc_bwd_workspace_floats(int m_rows, int n_in, int n_out) {
    if (m_rows <= 0 || n_in <= 0 || n_out <= 0) return 0;
    return (size_t)((n_out + FC_DXF - 1) / FC_DXF) * m_rows * n_in;
}

template <class S>
static int fc_bwd_impl(const float* x, const float* w, const float* y, const float* dy, float* dw, float* db, float* dx,
                          float* workspace, int m_rows, int n_in, int n_out, int act, int hw, int cstride, int coff, void* stream) {
    if (!x || !w || !y || !dy || !dw || !db || m_rows <= 0 || n_in <= 0 || n_out <= 0) return RCF_EINVAL;
    if (dx && !workspace) return RCF_EINVAL;
    if ((size_t)FC_MAX_M * n_in * 4 > 48 * 1024 || (hw > 1 && (n_out % hw != 0 || cstride < coff + n_out / hw))) return RCF_EUNSUPPORTED;
    const unsigned nb = (unsigned)((n_out + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    const size_t ystride = hw > 1 ? (size_t)hw * cstride : (size_t)n_out;
    if (dx) {
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_dx_partial_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      FC_DXF * FC_MAX_M * (int)sizeof(float));
            attr_done = true;
        }
    }
    // dW and db are sums over ALL rows: the first block overwrites, later blocks accumulate; dx rows are independent per block
    for (int m0 = 0; m0 < m_rows; m0 += FC_MAX_M) {
        const int mb = m_rows - m0 < FC_MAX_M ? m_rows - m0 : FC_MAX_M;
        const size_t lds = (size_t)mb * FC_KCH * sizeof(float);
        const float* xb = x + (size_t)m0 * n_in;
        const float* yb = rcf_at_host<S>(y, (size_t)m0 * ystride);
        const float* dyb = rcf_at_host<S>(dy, (size_t)m0 * ystride);
        const int acc = m0 > 0 ? 1 : 0;
        const dim3 gb(nb, (unsigned)((n_in + FC_KCH - 1) / FC_KCH));
        if (mb <= 16) hipLaunchKernelGGL((fc_bwd_kernel<16, S>), gb, dim3(256), lds, st, xb, yb, dyb, dw, db, mb, n_in, n_out, act, hw, cstride, coff, acc);
        else if (mb <= 32) hipLaunchKernelGGL((fc_bwd_kernel<32, S>), gb, dim3(256), lds, st, xb, yb, dyb, dw, db, mb, n_in, n_out, act, hw, cstride, coff, acc);
        else hipLaunchKernelGGL((fc_bwd_kernel<64, S>), gb, dim3(256), lds, st, xb, yb, dyb, dw, db, mb, n_in, n_out, act, hw, cstride, coff, acc);
        if (dx) {
            const unsigned nbx = (unsigned)((n_out + FC_DXF - 1) / FC_DXF);
            float* wsb = workspace + (size_t)nbx * m0 * n_in;
            hipLaunchKernelGGL((fc_dx_partial_kernel<S>), dim3(nbx), dim3(256), (size_t)FC_DXF * mb * sizeof(float), st, w, yb, dyb, wsb, mb,
                               n_in, n_out, act, hw, cstride, coff);
            const int n = mb * n_in;
            hipLaunchKernelGGL(fc_dx_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, wsb, dx + (size_t)m0 * n_in, (int)nbx, n);
        }
    }
    return rcf_launch_status();
}

extern "C" size_t rcf_bce_workspace_doubles(void) { return (size_t)BCE_BLOCKS * 2; }

extern "C" int rcf_bce_loss_fwd(const float* logit, const float* target, const float* valid, double* workspace, double* sums,
                                float* loss, long long n, float pos_weight, void* stream) {
    if (!logit || !target || !valid || !workspace || !sums || !loss || n <= 0) return RCF_EINVAL;
    const unsigned nb = grid_for(n, BCE_BLOCKS);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bce_partial_kernel, dim3(nb), dim3(256), 0, st, logit, target, valid, workspace, n, pos_weight);
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(64), 0, st, workspace, (int)nb, sums, loss);
    return rcf_launch_status();
}

extern "C" int rcf_bce_loss_bwd(const float* logit, const float* target, const float* valid, const double* sums,
                                const float* upstream, float* dlogit, long long n, float pos_weight, void* stream) {
    if (!logit || !target || !valid || !sums || !upstream || !dlogit || n <= 0) return RCF_EINVAL;
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, logit, target, valid, sums, upstream,
                       dlogit, n, pos_weight);
    return rcf_launch_status();
}

// ---- exported instances: NAME for fp32 NHWC tensors, NAME_b16 for bf16 NHWC tensors (same argument lists; see include/rcf_hip.h)
extern "C" int rcf_roi_pool_fwd(const float* in, const float* rois, float* out, int* argmax, int n_roi, int n, int h, int w, int c,
                                int pooled_h, int pooled_w, float spatial_scale, int out_cstride, int out_coff, void* stream) { return roi_pool_fwd_impl<StF32>(in, rois, out, argmax, n_roi, n, h, w, c, pooled_h, pooled_w, spatial_scale, out_cstride, out_coff, stream); }
extern "C" int rcf_roi_pool_fwd_b16(const float* in, const float* rois, float* out, int* argmax, int n_roi, int n, int h, int w, int c,
                                int pooled_h, int pooled_w, float spatial_scale, int out_cstride, int out_coff, void* stream) { return roi_pool_fwd_impl<StB16>(in, rois, out, argmax, n_roi, n, h, w, c, pooled_h, pooled_w, spatial_scale, out_cstride, out_coff, stream); }
extern "C" int rcf_roi_pool_bwd(const float* dout, const int* argmax, const float* rois, float* din, int n_roi, int n, int h, int w,
                                int c, int pooled_h, int pooled_w, int dout_cstride, int dout_coff, void* stream) { return roi_pool_bwd_impl<StF32>(dout, argmax, rois, din, n_roi, n, h, w, c, pooled_h, pooled_w, dout_cstride, dout_coff, stream); }
extern "C" int rcf_roi_pool_bwd_gather(const float* dout, const int* argmax, const float* rois, float* din, int din_accumulate, int n_roi,
                                       int n, int h, int w, int c, int pooled_h, int pooled_w, float spatial_scale, int dout_cstride,
                                       int dout_coff, void* stream) {
    return roi_pool_bwd_gather_impl<StF32>(dout, argmax, rois, din, din_accumulate, n_roi, n, h, w, c, pooled_h, pooled_w, spatial_scale,
                                           dout_cstride, dout_coff, stream);
}
extern "C" int rcf_roi_pool_bwd_gather_b16(const float* dout, const int* argmax, const float* rois, float* din, int din_accumulate,
                                           int n_roi, int n, int h, int w, int c, int pooled_h, int pooled_w, float spatial_scale,
                                           int dout_cstride, int dout_coff, void* stream) {
    return roi_pool_bwd_gather_impl<StB16>(dout, argmax, rois, din, din_accumulate, n_roi, n, h, w, c, pooled_h, pooled_w, spatial_scale,
                                           dout_cstride, dout_coff, stream);
}
extern "C" int rcf_roi_pool_bwd_b16(const float* dout, const int* argmax, const float* rois, float* din, int n_roi, int n, int h, int w,
                                int c, int pooled_h, int pooled_w, int dout_cstride, int dout_coff, void* stream) { return roi_pool_bwd_impl<StB16>(dout, argmax, rois, din, n_roi, n, h, w, c, pooled_h, pooled_w, dout_cstride, dout_coff, stream); }
extern "C" int rcf_fc_fwd(const float* x, const float* w, const float* bias, float* y, int m_rows, int n_in, int n_out, int act,
                          int hw, int cstride, int coff, void* stream) { return fc_fwd_impl<StF32>(x, w, bias, y, m_rows, n_in, n_out, act, hw, cstride, coff, stream); }
extern "C" int rcf_fc_fwd_b16(const float* x, const float* w, const float* bias, float* y, int m_rows, int n_in, int n_out, int act,
                          int hw, int cstride, int coff, void* stream) { return fc_fwd_impl<StB16>(x, w, bias, y, m_rows, n_in, n_out, act, hw, cstride, coff, stream); }
extern "C" int rcf_fc_bwd(const float* x, const float* w, const float* y, const float* dy, float* dw, float* db, float* dx,
                          float* workspace, int m_rows, int n_in, int n_out, int act, int hw, int cstride, int coff, void* stream) { return fc_bwd_impl<StF32>(x, w, y, dy, dw, db, dx, workspace, m_rows, n_in, n_out, act, hw, cstride, coff, stream); }
extern "C" int rcf_fc_bwd_b16(const float* x, const float* w, const float* y, const float* dy, float* dw, float* db, float* dx,
                          float* workspace, int m_rows, int n_in, int n_out, int act, int hw, int cstride, int coff, void* stream) { return fc_bwd_impl<StB16>(x, w, y, dy, dw, db, dx, workspace, m_rows, n_in, n_out, act, hw, cstride, coff, stream); }
