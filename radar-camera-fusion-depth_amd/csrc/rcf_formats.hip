// On-disk sample formats of the FusionNet loaders finished on the device.  The host only inflates the PNGs
// (PIL / zlib) and hands over the INTEGER pixels it got -- 3 bytes per image pixel and 2 per range-map pixel instead of the 12 + 4
// bytes of the float32 arrays the reference builds on the host (src/data_utils.py:167-335) -- and one launch per tensor does the
// rest for the whole batch: the per-sample crop of datasets.random_crop (src/datasets.py:19-109), HWC -> CHW, the float32
// conversion, the division by the encoding multiplier (256 for depth, 2^14 for responses; FusionNet's datasets read the response
// with the depth decoder, src/datasets.py:413-415), `z[z <= 0] = 0` and the validity map.  The inverse (save_depth /
// save_response: np.uint32(z * multiplier)) and the lidar / radar point plot of the dataset setup (points_to_depth_map,
// setup/setup_dataset_nuscenes_with_denseGT.py:814-840) are here as well.  All integer / index work is bit-exact; the float
// steps are single IEEE operations (int -> float conversion, one division or one multiplication), so they are bit-exact too.
#include "rcf_common.h"

namespace {

// image: u8 [n][H][W][3] -> f32 [n][3][h][w], rows/columns [y0, y0 + h) x [x0, x0 + w) of sample b (crop[b] = {y0, x0})
__global__ void __launch_bounds__(256) decode_image_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int H, int W,
                                                          int h, int w, const int* __restrict__ crop, int normalize) {
    const int b = blockIdx.z, y = blockIdx.y;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    const int y0 = crop ? crop[2 * b] : 0, x0 = crop ? crop[2 * b + 1] : 0;
    const unsigned char* p = src + (((size_t)b * H + (y0 + y)) * W + (x0 + x)) * 3;
    const size_t plane = (size_t)h * w;
    float* q = dst + (size_t)b * 3 * plane + (size_t)y * w + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = (float)p[c];
        q[c * plane] = normalize ? v / 255.0f : v;      // load_image: image / 255.0 (src/data_utils.py:196)
    }
}

// range map: integer [n][H][W] -> f32 [n][1][h][w] (and optionally the validity map), src/data_utils.py:200-269, :288-318
template <typename T>
__global__ void __launch_bounds__(256) decode_map_kernel(const T* __restrict__ src, float* __restrict__ dst, float* __restrict__ valid,
                                                        int H, int W, int h, int w, const int* __restrict__ crop, float multiplier,
                                                        int clamp) {
    const int b = blockIdx.z, y = blockIdx.y;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    const int y0 = crop ? crop[2 * b] : 0, x0 = crop ? crop[2 * b + 1] : 0;
    float z = (float)src[((size_t)b * H + (y0 + y)) * W + (x0 + x)] / multiplier;
    if (clamp && z <= 0.f) z = 0.f;                     // load_depth: z[z <= 0] = 0.0; load_response has no such line
    const size_t o = ((size_t)b * h + y) * w + x;
    dst[o] = z;
    if (valid != nullptr) valid[o] = z > 0.f ? 1.f : z;   // v = z.copy(); v[z > 0] = 1.0
}

// np.uint32(z * multiplier): float32 product, C conversion to a 64-bit integer (truncation), low 32 bits
__global__ void __launch_bounds__(256) encode_map_kernel(const float* __restrict__ z, unsigned* __restrict__ out, long long count,
                                                        float multiplier) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float v = z[i] * multiplier;
    out[i] = (unsigned)(unsigned long long)(long long)v;
}

// points_to_depth_map: depth_map[round(y), round(x)] = depth[k] for k = 0, 1, ... (the LAST point of a pixel wins).
// Pass 1: per pixel the largest k + 1 (atomicMax on a zeroed int map); pass 2 reads that point's depth.  np.round is
// round-half-to-even (rintf); negative indices wrap once like numpy's; anything else out of range is skipped and counted.
__global__ void __launch_bounds__(256) plot_points_kernel(const float* __restrict__ xs, const float* __restrict__ ys, int n_points, int H,
                                                         int W, int* __restrict__ owner, int* __restrict__ n_bad) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_points) return;
    int x = (int)rintf(xs[k]), y = (int)rintf(ys[k]);
    if (x < 0) x += W;
    if (y < 0) y += H;
    if (x < 0 || x >= W || y < 0 || y >= H) {
        atomicAdd(n_bad, 1);
        return;
    }
    atomicMax(owner + (size_t)y * W + x, k + 1);
}

__global__ void __launch_bounds__(256) plot_gather_kernel(const int* __restrict__ owner, const float* __restrict__ depth,
                                                         float* __restrict__ out, int count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const int k = owner[i];
    out[i] = k > 0 ? depth[k - 1] : 0.f;
}

}   // namespace

extern "C" {

int rcf_decode_image_u8(const unsigned char* src, float* dst, int n, int src_h, int src_w, int h, int w, const int* crop_yx, int normalize,
                        void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (src == nullptr || dst == nullptr || n < 0 || h <= 0 || w <= 0 || h > src_h || w > src_w) return RCF_EINVAL;
    if (n == 0) return RCF_OK;
    if (h > 65535 || n > 65535) return RCF_EUNSUPPORTED;
    hipLaunchKernelGGL(decode_image_kernel, dim3((w + 255) / 256, h, n), dim3(256), 0, stream, src, dst, src_h, src_w, h, w, crop_yx,
                       normalize);
    return rcf_launch_status();
}

int rcf_decode_map(const void* src, int src_type, float* dst, float* validity, int n, int src_h, int src_w, int h, int w,
                   const int* crop_yx, float multiplier, int clamp_nonpositive, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (src == nullptr || dst == nullptr || n < 0 || h <= 0 || w <= 0 || h > src_h || w > src_w || !(multiplier > 0.f))
        return RCF_EINVAL;
    if (n == 0) return RCF_OK;
    if (h > 65535 || n > 65535) return RCF_EUNSUPPORTED;
    const dim3 grid((w + 255) / 256, h, n), block(256);
    switch (src_type) {
    case RCF_PIXEL_U8:
        hipLaunchKernelGGL(decode_map_kernel<unsigned char>, grid, block, 0, stream, static_cast<const unsigned char*>(src), dst, validity,
                           src_h, src_w, h, w, crop_yx, multiplier, clamp_nonpositive);
        break;
    case RCF_PIXEL_U16:
        hipLaunchKernelGGL(decode_map_kernel<unsigned short>, grid, block, 0, stream, static_cast<const unsigned short*>(src), dst,
                           validity, src_h, src_w, h, w, crop_yx, multiplier, clamp_nonpositive);
        break;
    case RCF_PIXEL_I32:
        hipLaunchKernelGGL(decode_map_kernel<int>, grid, block, 0, stream, static_cast<const int*>(src), dst, validity, src_h, src_w, h,
                           w, crop_yx, multiplier, clamp_nonpositive);
        break;
    default:
        return RCF_EUNSUPPORTED;
    }
    return rcf_launch_status();
}

int rcf_encode_map_u32(const float* z, unsigned* out, long long count, float multiplier, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (z == nullptr || out == nullptr || count < 0) return RCF_EINVAL;
    if (count == 0) return RCF_OK;
    hipLaunchKernelGGL(encode_map_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, z, out, count, multiplier);
    return rcf_launch_status();
}

size_t rcf_points_to_depth_map_workspace_bytes(int h, int w) { return ((size_t)h * w + 1) * sizeof(int); }

int rcf_points_to_depth_map(const float* xs, const float* ys, const float* depth, int n_points, float* depth_map, int h, int w,
                            void* workspace, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (depth_map == nullptr || workspace == nullptr || h <= 0 || w <= 0 || n_points < 0) return RCF_EINVAL;
    if (n_points > 0 && (xs == nullptr || ys == nullptr || depth == nullptr)) return RCF_EINVAL;
    int* owner = static_cast<int*>(workspace);
    int* n_bad = owner + (size_t)h * w;     // left in the workspace for the caller: number of points outside the image
    hipError_t e = hipMemsetAsync(workspace, 0, rcf_points_to_depth_map_workspace_bytes(h, w), stream);
    if (e != hipSuccess) return (int)e;
    if (n_points > 0)
        hipLaunchKernelGGL(plot_points_kernel, dim3((n_points + 255) / 256), dim3(256), 0, stream, xs, ys, n_points, h, w, owner, n_bad);
    hipLaunchKernelGGL(plot_gather_kernel, dim3((h * w + 255) / 256), dim3(256), 0, stream, owner, depth, depth_map, h * w);
    return rcf_launch_status();
}

}   // extern "C"
