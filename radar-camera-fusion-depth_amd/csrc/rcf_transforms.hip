// Batched input augmentation of the FusionNet training loop -- fusionnet_transforms.Transforms.transform
// (src/fusionnet_transforms.py:46-178) without per-sample Python loops and without the torch.max host synchronisation (:82):
// per-sample brightness / contrast / saturation (torchvision.transforms.functional.adjust_* semantics restated: blend with zero /
// the mean of the truncated grey image / the truncated grey pixel, clamp to [0, bound], cast back to the image dtype), the
// float conversion and range normalisation, and the horizontal / vertical flips of images and range maps, NCHW fp32 on gfx950.
// Whether the images are on the 0..255 scale (then the reference works on .int() images, :81-83) is decided ON DEVICE from the
// global maximum; every sample's decisions and factors are device tensors, so nothing synchronises with the host.
#include "rcf_common.h"

namespace {

__device__ __forceinline__ float tf_blend(float a, float b, float ratio, float bound, bool as_int) {
    // torchvision _blend: (ratio * img1 + (1 - ratio) * img2).clamp(0, bound).to(img1.dtype)
    float v = ratio * a + (1.0f - ratio) * b;
    v = fminf(fmaxf(v, 0.f), bound);
    return as_int ? truncf(v) : v;
}

__device__ __forceinline__ float tf_gray(float r, float g, float b, bool as_int) {
    // torchvision rgb_to_grayscale: (0.2989 r + 0.587 g + 0.114 b).to(img.dtype)
    const float l = 0.2989f * r + 0.587f * g + 0.114f * b;
    return as_int ? truncf(l) : l;
}

// max over all images (non-negative data): unsigned bit pattern max
__global__ void __launch_bounds__(256) tf_max_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ out_bits) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, x[i]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(fmaxf(m, 0.f)));
}

// per-image sum of the (truncated) grey value of the brightness-adjusted image -> partial[n][blocks]
__global__ void __launch_bounds__(256) tf_gray_sum_kernel(const float* __restrict__ img, const unsigned* __restrict__ max_bits,
                                                         const unsigned char* __restrict__ do_b, const float* __restrict__ f_b,
                                                         double* __restrict__ part, int hw) {
    __shared__ double sm[4];
    const int b = blockIdx.y;
    const bool as_int = __uint_as_float(*max_bits) > 1.0f;
    const float bound = as_int ? 255.f : 1.f;
    const float* p = img + (size_t)b * 3 * hw;
    const bool db = do_b != nullptr && do_b[b];
    const float fb = db ? f_b[b] : 1.f;
    double s = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
        float r = p[i], g = p[hw + i], bl = p[2 * hw + i];
        if (as_int) { r = truncf(r); g = truncf(g); bl = truncf(bl); }
        if (db) { r = tf_blend(r, 0.f, fb, bound, as_int); g = tf_blend(g, 0.f, fb, bound, as_int); bl = tf_blend(bl, 0.f, fb, bound, as_int); }
        s += (double)tf_gray(r, g, bl, as_int);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[(size_t)b * gridDim.x + blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// everything else, one thread per pixel of one image: photometric chain, normalisation, flips (gather from the source position)
__global__ void __launch_bounds__(256) tf_apply_kernel(const float* __restrict__ img, float* __restrict__ out,
                                                       const unsigned* __restrict__ max_bits, const unsigned char* __restrict__ do_b,
                                                       const float* __restrict__ f_b, const unsigned char* __restrict__ do_c,
                                                       const float* __restrict__ f_c, const unsigned char* __restrict__ do_s,
                                                       const float* __restrict__ f_s, const unsigned char* __restrict__ do_hf,
                                                       const unsigned char* __restrict__ do_vf, const double* __restrict__ part,
                                                       int n_part, int h, int w, int norm_mode) {
    const int b = blockIdx.y;
    const int hw = h * w;
    const bool as_int = __uint_as_float(*max_bits) > 1.0f;
    const float bound = as_int ? 255.f : 1.f;
    const bool db = do_b != nullptr && do_b[b], dc = do_c != nullptr && do_c[b], ds = do_s != nullptr && do_s[b];
    const bool hf = do_hf != nullptr && do_hf[b], vf = do_vf != nullptr && do_vf[b];
    float mean = 0.f;
    if (dc) {
        double s = 0.0;
        for (int i = 0; i < n_part; ++i) s += part[(size_t)b * n_part + i];
        mean = (float)(s / (double)hw);
    }
    const float fb = db ? f_b[b] : 1.f, fc = dc ? f_c[b] : 1.f, fs = ds ? f_s[b] : 1.f;
    const float* p = img + (size_t)b * 3 * hw;
    float* o = out + (size_t)b * 3 * hw;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
        const int y = i / w, x = i - y * w;
        const int sy = vf ? h - 1 - y : y, sx = hf ? w - 1 - x : x;
        const int si = sy * w + sx;
        float r = p[si], g = p[hw + si], bl = p[2 * hw + si];
        if (as_int) { r = truncf(r); g = truncf(g); bl = truncf(bl); }                 // images.int() (:81-83)
        if (db) { r = tf_blend(r, 0.f, fb, bound, as_int); g = tf_blend(g, 0.f, fb, bound, as_int); bl = tf_blend(bl, 0.f, fb, bound, as_int); }
        if (dc) { r = tf_blend(r, mean, fc, bound, as_int); g = tf_blend(g, mean, fc, bound, as_int); bl = tf_blend(bl, mean, fc, bound, as_int); }
        if (ds) {
            const float l = tf_gray(r, g, bl, as_int);
            r = tf_blend(r, l, fs, bound, as_int); g = tf_blend(g, l, fs, bound, as_int); bl = tf_blend(bl, l, fs, bound, as_int);
        }
        if (norm_mode == 1) { r = r / 255.0f; g = g / 255.0f; bl = bl / 255.0f; }       // [0, 1]   (:196-199)
        else if (norm_mode == 2) { r = 2.0f * (r / 255.0f) - 1.0f; g = 2.0f * (g / 255.0f) - 1.0f; bl = 2.0f * (bl / 255.0f) - 1.0f; }
        o[i] = r; o[hw + i] = g; o[2 * hw + i] = bl;
    }
}

// flips of the range maps (N, C, H, W)
__global__ void __launch_bounds__(256) tf_flip_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                      const unsigned char* __restrict__ do_hf, const unsigned char* __restrict__ do_vf,
                                                      int c, int h, int w) {
    const int b = blockIdx.y;
    const bool hf = do_hf != nullptr && do_hf[b], vf = do_vf != nullptr && do_vf[b];
    const long long per = (long long)c * h * w;
    const float* p = in + (size_t)b * per;
    float* o = out + (size_t)b * per;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % w);
        const int y = (int)((i / w) % h);
        const long long ch = i / ((long long)w * h);
        const int sy = vf ? h - 1 - y : y, sx = hf ? w - 1 - x : x;
        o[i] = p[(ch * h + sy) * w + sx];
    }
}

constexpr int TF_PARTS = 64;

}   // namespace

extern "C" size_t rcf_transform_workspace_bytes(int n) { return 16 + (size_t)n * TF_PARTS * sizeof(double); }

// img / out: (N, 3, H, W) fp32 (out may not alias img when a flip is requested).  do_* : N bytes each (nullable = never),
// f_* : N floats.  norm_mode: 0 keep [0,255], 1 -> [0,1], 2 -> [-1,1].  workspace: rcf_transform_workspace_bytes(N), 16-B aligned.
extern "C" int rcf_transform_images(const float* img, float* out, int n, int h, int w, const unsigned char* do_brightness,
                                    const float* f_brightness, const unsigned char* do_contrast, const float* f_contrast,
                                    const unsigned char* do_saturation, const float* f_saturation, const unsigned char* do_hflip,
                                    const unsigned char* do_vflip, int norm_mode, void* workspace, void* stream) {
    if (!img || !out || !workspace || n <= 0 || h <= 0 || w <= 0 || norm_mode < 0 || norm_mode > 2) return RCF_EINVAL;
    if ((do_brightness && !f_brightness) || (do_contrast && !f_contrast) || (do_saturation && !f_saturation)) return RCF_EINVAL;
    if (img == out && (do_hflip || do_vflip)) return RCF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    unsigned* max_bits = reinterpret_cast<unsigned*>(workspace);
    double* part = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(workspace) + 16);
    if (hipMemsetAsync(max_bits, 0, 16, st) != hipSuccess) return rcf_launch_status();
    const long long total = (long long)n * 3 * h * w;
    unsigned mb = (unsigned)((total + 255) / 256);
    if (mb > 2048) mb = 2048;
    hipLaunchKernelGGL(tf_max_kernel, dim3(mb), dim3(256), 0, st, img, total, max_bits);
    if (do_contrast)
        hipLaunchKernelGGL(tf_gray_sum_kernel, dim3(TF_PARTS, n), dim3(256), 0, st, img, max_bits, do_brightness, f_brightness, part, h * w);
    unsigned gb = (unsigned)(((long long)h * w + 255) / 256);
    if (gb > 1024) gb = 1024;
    hipLaunchKernelGGL(tf_apply_kernel, dim3(gb, n), dim3(256), 0, st, img, out, max_bits, do_brightness, f_brightness, do_contrast,
                       f_contrast, do_saturation, f_saturation, do_hflip, do_vflip, part, TF_PARTS, h, w, norm_mode);
    return rcf_launch_status();
}

extern "C" int rcf_transform_flip(const float* in, float* out, int n, int c, int h, int w, const unsigned char* do_hflip,
                                  const unsigned char* do_vflip, void* stream) {
    if (!in || !out || in == out || n <= 0 || c <= 0 || h <= 0 || w <= 0) return RCF_EINVAL;
    long long per = (long long)c * h * w;
    unsigned gb = (unsigned)((per + 255) / 256);
    if (gb > 2048) gb = 2048;
    hipLaunchKernelGGL(tf_flip_kernel, dim3(gb, n), dim3(256), 0, (hipStream_t)stream, in, out, do_hflip, do_vflip, c, h, w);
    return rcf_launch_status();
}
