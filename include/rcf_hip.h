/*
 * rcf_hip.h -- C ABI of librcf_hip.so: the FusionNet forward/backward hot path of
 * nesl/radar-camera-fusion-depth as hand-written HIP kernels for MI355X (gfx950).
 *
 * The reference has no native code and no FFI: its hot path is the stock torch.nn
 * modules called from src/net_utils.py / src/networks.py / src/fusionnet_model.py.  Each entry point
 * below therefore cites the reference Python interface (file:line, relative to the reference root)
 * whose computation it replaces; INTEGRATION.md shows the ctypes binding a reference maintainer adds.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked host.
 *   - activations are fp32, NHWC contiguous: x[n][y][x][c].  Weights cross the boundary in the
 *     reference's own layout, OIHW contiguous (torch.nn.Conv2d.weight, src/net_utils.py:63).
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); it never
 *     synchronises, allocates or frees.  Workspaces are caller-owned; sizes come from *_query.
 *   - stateless and re-entrant for ONE device per process -- a HARD LIMIT, not a default: the deployment model is one process
 *     per GPU (torch.distributed over RCCL).  The only process state is a cache of per-kernel launch facts (occupancy, CU count,
 *     dynamic-LDS attribute) and a device-global zero page, filled idempotently on first use FOR THE DEVICE CURRENT AT THAT TIME;
 *     a second device in the same process would reuse the first one's facts.  Work is enqueued on `stream`, which must belong to
 *     the device that is current in the calling thread (the Python host makes the tensors' device current around every call).
 *     Return value: 0 = ok; RCF_E* < 0 = invalid argument /
 *     unsupported shape; > 0 = a hipError_t from a launch.  Nothing throws or aborts.
 */
#ifndef RCF_HIP_H
#define RCF_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCF_OK 0
#define RCF_EINVAL (-1)      /* null pointer, non-positive extent, inconsistent shape */
#define RCF_EUNSUPPORTED (-2) /* shape outside what the kernels implement */

#define RCF_GATHER_DIRECT 0     /* source 1 is read as is */
#define RCF_GATHER_NEAREST 1    /* source 1 is nearest-upsampled to (h_in,w_in): F.interpolate(x,size), src/net_utils.py:196 */
#define RCF_GATHER_ZERO_INSERT 2 /* source 1 is zero-dilated by 2 (transposed conv: dgrad of a stride-2 conv) */
#define RCF_GATHER_STRIDED2 3   /* logical pixel (y,x) is physical pixel (2y+in_off_y, 2x+in_off_x) of source 1: one phase of a tensor */

/* Phase decomposition (rcf_phase_weights): a 3x3 conv on an exactly-2x nearest-upsampled tensor is four 2x2 convs
 * on the source (4/9 of the MACs); the input gradient of a 3x3 stride-2 conv is four 2x2 convs on dZ. */
#define RCF_PHASE_UP2X_FWD 0   /* Wp[a][b][o][i][t][u] = sum of the 3x3 taps that land on source tap (t,u) for output phase (a,b) */
#define RCF_PHASE_UP2X_DGRAD 1 /* the same, channel-transposed and tap-flipped: maps dZ phase (a,b) to dX */
#define RCF_PHASE_S2_DGRAD 2   /* W[o][i][ky][kx] selected per input-pixel phase (a,b) of a stride-2 conv, channel-transposed */

#define RCF_W_FORWARD 0 /* weight tensor is used as stored: out channel O, in channel I */
#define RCF_W_DGRAD 1   /* flipped taps and swapped roles: this conv maps dZ (O channels) to dX (a slice of I) */

#define RCF_ACT_NONE 0
#define RCF_ACT_LEAKY_RELU 1 /* negative_slope 0.20, src/net_utils.py:15 */

/* Implicit-GEMM convolution descriptor: out[n,oy,ox,:] = sum_taps W[tap] . in[n, oy*stride-pad+ky, ox*stride-pad+kx, :]
 * where `in` is the channel concatenation [source1 | source2] (torch.cat([deconv, skip], 1), src/net_utils.py:565)
 * and source 1 may be gathered (nearest upsample / zero insert).  */
#define RCF_PREC_FP32 0
#define RCF_PREC_BF16 1
#define RCF_PREC_F16X2 2
/* Storage of the NHWC activation / gradient tensors a call reads and writes. */
#define RCF_STORE_FP32 0 /* fp32 tensors: the reference's configuration */
#define RCF_STORE_BF16 1 /* bf16 tensors in HBM (BASELINE.json configs 2-4): bf16 storage and MFMA operands, fp32 accumulation; weights,
                          * BatchNorm statistics / coefficients, loss sums and the optimizer stay fp32.  Requires RCF_PREC_BF16. */

typedef struct rcf_conv_desc {
    int n;              /* batch */
    int h_in, w_in;     /* logical input extent seen by the conv */
    int c1, c2;         /* channels of source 1 and source 2 (c2 == 0: single source) */
    int h_src1, w_src1; /* physical extent of source 1 (== h_in,w_in for RCF_GATHER_DIRECT) */
    int gather1;        /* RCF_GATHER_* */
    int h_out, w_out, c_out;
    int ksize;          /* 1, 2 (phase convs), 3, 4 (the stems on the space-to-depth image, bf16 tensors) or 7 */
    int stride;         /* 1 or 2 */
    int pad;            /* top pad; ksize/2 for the reference's Conv2d (src/net_utils.py:61); ksize-1-pad for its dgrad */
    int pad_x;          /* left pad (== pad except for the 2x2 phase convs) */
    int w_mode;         /* RCF_W_FORWARD / RCF_W_DGRAD */
    int w_o, w_i;       /* dims of the OIHW weight tensor this conv is packed from */
    int w_i_off;        /* RCF_W_DGRAD: first input channel of the dX slice produced (c_out channels) */
    int accumulate;     /* out += result instead of out = result */
    /* phase addressing: output pixel (oy,ox) lives at (oy*out_stride+out_off_y, ox*out_stride+out_off_x) of a tensor with
     * out_h_phys x out_w_phys pixels (out_stride 1, offsets 0, phys == h_out,w_out for an ordinary conv); rcf_conv2d_wgrad
     * reads dZ through the same map.  in_off_*: RCF_GATHER_STRIDED2 only. */
    int out_stride, out_off_y, out_off_x, out_h_phys, out_w_phys;
    int in_off_y, in_off_x;
    /* phase_sum != 0 (ksize 2, RCF_GATHER_STRIDED2): out = sum over the four input phases (a,b) of the 2x2 conv of phase
     * (a,b) with pad (a,b) -- the whole input gradient of an up-2x conv in ONE launch.  `packed` then holds the four
     * phases' packed weights back to back (each rcf_conv_info.packed_weight_floats long).  rcf_conv2d_wgrad on a phase_sum == 1
     * descriptor (in_off_* ignored): the four phase weight gradients of a 3x3 stride-2 convolution -- x gathered at (2y+a, 2x+b),
     * the same dZ -- in ONE launch, dw = [4][c_out][c_in][2][2] (what rcf_phase_wgrad_gather_s2 takes); split kernels only.
     * phase_sum == 2 (ksize 2, RCF_GATHER_DIRECT, out_stride 2, split / DMA kernels only): the four OUTPUT phases of an up-2x
     * forward conv3x3(nearest_up2x(x)) (src/net_utils.py:156-198) in ONE launch (phase (a,b): pad (1-a, 1-b), outputs at
     * (2y+a, 2x+b); pad / out_off_* of the descriptor are ignored): the 3x3-halo tile of x is staged once per channel chunk and all
     * four 2x2 phase convolutions run from it (bf16 tensors; fp32 tensors under RCF_PREC_F16X2), bitwise the four per-phase launches;
     * `packed` holds the four phases' packed weights back to back; the BatchNorm statistics cover all four phases.
     * rcf_conv2d_wgrad on such a descriptor: the four phases' weight gradients in ONE launch, dw = [4][c_out][c_in][2][2] (what
     * rcf_phase_wgrad_fold takes), bitwise the four per-phase calls; the four phases of a tile run on one XCD at the same time, so
     * x is fetched from HBM once and from that L2 three times.  Split weight-gradient kernels only (RCF_EUNSUPPORTED otherwise).
     * phase_sum == 3 (ksize 2, RCF_GATHER_DIRECT, pad 0, out_stride 2, h_out / w_out = ceil(out_*_phys / 2)): the four OUTPUT phases of
     * the input gradient of a 3x3 stride-2 convolution (the transposed convolution behind loss.backward() of the encoder's stride-2
     * layers) in ONE launch: in1 = dZ, out = dX (out_h_phys x out_w_phys = the forward input), phase (a,b) writes dX(2y+a, 2x+b) from
     * the taps (ty,tx) with ty <= a, tx <= b of its 2x2 phase weights (RCF_PHASE_S2_DGRAD; the other 7 of the 16 are zero by
     * construction and are not multiplied); `packed` = the four phases' packed weights back to back; accumulate allowed.  Bitwise
     * the four per-phase launches.  bf16 tensors and fp32 tensors under RCF_PREC_F16X2 only (RCF_EUNSUPPORTED otherwise: callers
     * keep the four launches there); no weight gradient (RCF_EUNSUPPORTED). */
    int phase_sum;
    /* RCF_PREC_FP32 (0): fp32 results (the reference's arithmetic; f32 MFMA or the exact 3-plane bf16 split).
     * RCF_PREC_BF16 (1): operands rounded to bf16 (nearest even), fp32 accumulate; honoured by the split kernels, every other
     * kernel keeps computing in fp32.
     * RCF_PREC_F16X2 (2), fp32 tensors only: each operand of the split kernels as TWO fp16 planes of x * s, s the power of two that
     * puts its tensor's max|x| into [2^14, 2^15) (22-23 significant bits, fp16 denormals honoured by the MFMA), the three products
     * a0*b0 + a0*b1 + a1*b0, fp32 accumulate, result * 1/(s_a s_b): the accuracy class of the three-plane bf16 split (measured against
     * fp64: within 2x of the f32 MFMA) at half its matrix work.  The per-tensor maxima arrive by device pointer (rcf_conv_scales; the
     * kernels that write the tensors accumulate them: rcf_*_amax below); a null maximum means scale 1 (data inside fp16's range).
     * Every other kernel keeps computing in fp32. */
    int precision;
    /* RCF_STORE_FP32 / RCF_STORE_BF16: element type of in1, in2, out, res and dz (`const void*` below).  The two 7x7 stem
     * convolutions (c1 <= 4) always read an fp32 input -- the network input is never rounded -- and write `storage`. */
    int storage;
} rcf_conv_desc;

typedef struct rcf_conv_info {
    size_t packed_weight_floats; /* size of the packed-weight buffer for rcf_conv2d_pack_weights */
    int n_partials;              /* rows of the per-workgroup BN-statistics partial buffer [n_partials][2][c_out] */
    size_t wgrad_workspace_floats; /* workspace for rcf_conv2d_wgrad on the same (forward) descriptor */
    int kernel_id;               /* which tile configuration was selected (for profiling/logs) */
    int wgrad_kernel_id;         /* same for the weight-gradient kernel of a forward descriptor (0: none) */
    int bn_on_load;              /* 1: rcf_conv2d_fwd_bn accepts this descriptor (raw conv outputs + BN coefficients as inputs) */
    int wgrad_bn_on_load;        /* 1: rcf_conv2d_wgrad_bn accepts it */
    int fwd_act;                 /* 1: rcf_conv2d_fwd_act accepts it (inference epilogue in the matrix kernel) */
    int bn_bwd_sums;             /* 1: rcf_conv2d_dgrad_bn_sums accepts it (BatchNorm-backward sums in the input-gradient kernel) */
} rcf_conv_info;

const char* rcf_version(void);
/* 1 when the library was built with the gfx950 code object this process can launch (a device is present). */
int rcf_device_ok(void);

int rcf_conv2d_query(const rcf_conv_desc* d, rcf_conv_info* info);

/* OIHW -> kernel layout [n-tile][k-chunk][tap][BN][CK].  Replaces nothing in the reference; it is the
 * price of keeping torch.nn.Conv2d.weight's layout at the boundary. */
int rcf_conv2d_pack_weights(const rcf_conv_desc* d, const float* w_oihw, float* packed, void* stream);
/* RCF_PREC_F16X2: the two fp16 planes hold w * s_w, s_w from *amax_w = max|w| over the tensor (device pointer, nullable = scale 1;
 * an upper bound is fine).  The same pointer goes to rcf_conv2d_fwd_scaled, whose epilogue divides s_w out again. */
int rcf_conv2d_pack_weights_scaled(const rcf_conv_desc* d, const float* w_oihw, float* packed, const float* amax_w, void* stream);
/* The same for n (descriptor, weight, destination) triples in ceil(n / 36) launches instead of n: a training step packs ~200
 * weights of a few thousand elements each, every one a 4-5 us launch (all weights are constant from the start of the forward
 * pass to the end of the backward pass, so the host can pack them all up front).  Results are identical to n single calls. */
typedef struct rcf_pack_item {
    const rcf_conv_desc* desc;
    const float* w_oihw;
    float* packed;
    const float* amax_w; /* RCF_PREC_F16X2 descriptors: as in rcf_conv2d_pack_weights_scaled; ignored otherwise */
} rcf_pack_item;
int rcf_conv2d_pack_weights_batch(const rcf_pack_item* items, int n, void* stream);

/* Per-tensor maxima for RCF_PREC_F16X2 (each a DEVICE pointer to one float holding max|x| of the tensor or an upper bound of it;
 * nullable = scale 1).  amax_in2 is read only when c2 > 0: the two sources of a concat share the scale of the larger maximum. */
typedef struct rcf_conv_scales {
    const float* amax_in1;
    const float* amax_in2;
    const float* amax_w;  /* forward / input gradient: the weight tensor `packed` was built from (rcf_conv2d_pack_weights_scaled) */
    const float* amax_dz; /* weight gradient: the output gradient */
} rcf_conv_scales;
/* max|x| over n floats, accumulated into *amax (amax = max(amax, max|x|): zero it first for a fresh maximum).  The network's own
 * activations and gradients get theirs from the kernels that write them (rcf_bn_act_fwd_amax, ...); this entry point serves the
 * network inputs, the weights and the tests. */
int rcf_amax(const float* x, long long n, float* amax, void* stream);
typedef struct rcf_amax_item {
    const float* x;
    long long n;
    float* amax;
} rcf_amax_item;
/* n of those in ceil(n / 128) launches (all weights of a training step up front). */
int rcf_amax_batch(const rcf_amax_item* items, int n, void* stream);

/* torch.nn.Conv2d.forward, bias=False (src/net_utils.py:85); with gather1=NEAREST also the F.interpolate of
 * UpConv2d.forward (src/net_utils.py:195-198); with c2>0 also the torch.cat of DecoderBlock.forward
 * (src/net_utils.py:564-569).  With w_mode=DGRAD it is the input gradient autograd computes for that conv.
 * stat_partials (nullable, fp64): per-workgroup sum and sum-of-squares of the outputs per channel, consumed by
 * rcf_bn_finalize -- the batch statistics of torch.nn.BatchNorm2d (src/net_utils.py:82,86). */
int rcf_conv2d_fwd(const rcf_conv_desc* d, const void* in1, const void* in2, const float* packed,
                   void* out, double* stat_partials, void* stream);
/* The same for RCF_PREC_F16X2 descriptors on the split kernels (rcf_conv_info.kernel_id >= 45000), with the operands' maxima. */
int rcf_conv2d_fwd_scaled(const rcf_conv_desc* d, const void* in1, const void* in2, const float* packed,
                          void* out, double* stat_partials, const rcf_conv_scales* scales, void* stream);
/* The input gradient of a convolution (w_mode = DGRAD, or the four-phase input gradient of an up-2x convolution) whose output dx
 * is dY of a BatchNorm + LeakyReLU block -- the gradient w.r.t. the activation net_utils.Conv2d.forward returns
 * (src/net_utils.py:84-91) -- and that is the ONLY writer of it: besides dx the kernel leaves, per workgroup, the two sums
 * torch.nn.BatchNorm2d's backward needs, sum g and sum g * xhat with g = dx * lrelu'(bn_z * coef[0] + coef[1]) and
 * xhat = (bn_z - coef[2]) * coef[3] (bn_z: that block's raw conv output, same shape as dx; bn_coef: its rcf_bn_finalize
 * coefficients [4][c_out]), in sum_partials [n_partials][2][c_out] fp64 -- what rcf_bn_act_bwd_reduce would produce from a second
 * pass over dx and bn_z, ready for rcf_bn_bwd_finalize.  scales as for rcf_conv2d_fwd_scaled (nullable members = scale 1).
 * Only where rcf_conv_info.bn_bwd_sums is set (3x3 stride 1 / 2x2 with a plain output tensor; fp32 tensors under RCF_PREC_F16X2, or
 * bf16 tensors -- storage RCF_STORE_BF16: dz, dx and bn_z hold bf16, scales is ignored and may be null); RCF_EUNSUPPORTED otherwise. */
int rcf_conv2d_dgrad_bn_sums(const rcf_conv_desc* d, const void* dz, const float* packed, void* dx, const void* bn_z,
                             const float* bn_coef, double* sum_partials, const rcf_conv_scales* scales, void* stream);
/* BatchNorm + LeakyReLU of the PRODUCING block applied while the operand is staged ("BN on load"): in1 / in2 are raw conv
 * outputs z and coef1 / coef2 (nullable, one per source) the rcf_bn_finalize coefficients [4][c] of the block that produced
 * them; the kernel uses y = lrelu(z * coef[0][c] + coef[1][c]).  That block's activation tensor (net_utils.Conv2d.forward,
 * src/net_utils.py:84-91) is then never written.  Only where rcf_conv_info.bn_on_load is set; RCF_EUNSUPPORTED otherwise. */
int rcf_conv2d_fwd_bn(const rcf_conv_desc* d, const void* in1, const float* coef1, const void* in2, const float* coef2,
                      const float* packed, void* out, double* stat_partials, void* stream);

/* Inference form of net_utils.Conv2d.forward with eval-mode BatchNorm (src/net_utils.py:84-91) and of ResNetBlock's tail
 * (src/net_utils.py:311-323) in ONE kernel: `packed` holds the weights already multiplied by the BatchNorm scale
 * gamma / sqrt(running_var + eps) per output channel (rcf_scale_channels, then rcf_conv2d_pack_weights), bias[c_out] is
 * beta - running_mean * scale, and the kernel stores out = lrelu(conv + bias), or lrelu(lrelu(conv + bias) + res) when res (same
 * shape as out, nullable) is given.  No z tensor, no BN pass.  Only where rcf_conv_info.fwd_act is set; RCF_EUNSUPPORTED otherwise. */
int rcf_conv2d_fwd_act(const rcf_conv_desc* d, const void* in1, const void* in2, const float* packed, const float* bias,
                       const void* res, void* out, void* stream);
/* out[o][i] = w[o][i] * scale[o]  (o < n_out, i < inner): folds a per-output-channel factor into an OIHW weight tensor. */
int rcf_scale_channels(const float* w, const float* scale, float* out, int n_out, int inner, void* stream);

/* Weight gradient of the conv described by the FORWARD descriptor d: dw[o][i][ky][kx] (OIHW, same layout as
 * the parameter) = sum over pixels of in[...] * dz[...].  Replaces autograd's conv weight backward behind
 * loss.backward() (src/fusionnet_main.py:398). */
int rcf_conv2d_wgrad(const rcf_conv_desc* d, const void* in1, const void* in2, const void* dz,
                     float* dw_oihw, float* workspace, void* stream);
int rcf_conv2d_wgrad_scaled(const rcf_conv_desc* d, const void* in1, const void* in2, const void* dz,
                            float* dw_oihw, float* workspace, const rcf_conv_scales* scales, void* stream);
int rcf_conv2d_wgrad_bn(const rcf_conv_desc* d, const void* in1, const float* coef1, const void* in2, const float* coef2,
                        const void* dz, float* dw_oihw, float* workspace, void* stream);

/* Phase weights.  w: the layer's OIHW 3x3 weight [o][i][3][3]; out: [4 phases (a*2+b)][O'][I'][2][2] with
 * (O',I') = (o,i) for RCF_PHASE_UP2X_FWD and (i,o) for the two DGRAD modes.  Each phase block is then packed with
 * rcf_conv2d_pack_weights (ksize 2, RCF_W_FORWARD).  rcf_phase_wgrad_fold maps the four 2x2 phase weight gradients
 * of an up-2x conv back to the 3x3 gradient: dw[o][i][ky][kx] = sum_{a,b} dwp[a][b][o][i][t(a,ky)][u(b,kx)]. */
int rcf_phase_weights(const float* w_oihw, float* out, int o, int i, int mode, void* stream);
/* n phase-weight transforms in ceil(n / 96) launches; identical results to n calls of rcf_phase_weights. */
typedef struct rcf_phase_item {
    const float* w_oihw;
    float* out;
    int o, i, mode;
} rcf_phase_item;
int rcf_phase_weights_batch(const rcf_phase_item* items, int n, void* stream);
int rcf_phase_wgrad_fold(const float* dwp, float* dw_oihw, int o, int i, void* stream);
/* Weight gradient of a 3x3 STRIDE-2 convolution from four 2x2 weight gradients: phase (a,b) is rcf_conv2d_wgrad of the descriptor
 * {ksize 2, pad 1, pad_x 1, gather1 RCF_GATHER_STRIDED2, in_off (a,b), source = the conv's input, h_in/w_in/h_out/w_out = the conv's
 * OUTPUT extent} against the same dZ, written to dwp[a*2+b][o][i][2][2]; this call picks the nine real taps:
 * dw[o][i][ky][kx] = dwp[a(ky)][b(kx)][o][i][t(ky)][u(kx)], (a,t)(0) = (1,0), (a,t)(1) = (0,1), (a,t)(2) = (1,1).  (16 taps computed
 * for 9 used, on the bf16 matrix pipe instead of the f32 one.) */
int rcf_phase_wgrad_gather_s2(const float* dwp, float* dw_oihw, int o, int i, void* stream);

/* BatchNorm2d batch statistics -> affine coefficients.  partials: [n_partials][2][c] from rcf_conv2d_fwd.
 * coef: [4][c] = scale (gamma*invstd), shift (beta-mean*scale), mean, invstd.
 * training != 0: batch statistics, and running_mean/var are updated with `momentum` (unbiased variance),
 * exactly torch.nn.BatchNorm2d(eps=1e-5, momentum=0.1) in train mode (src/net_utils.py:82).
 * training == 0: coefficients from the running statistics (eval mode); partials is ignored. */
int rcf_bn_finalize(const double* partials, int n_partials, int c, double count,
                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                    float momentum, float eps, int training, float* coef, void* stream);

/* out = act(z*scale+shift); with res != NULL: out = lrelu(act(z*scale+shift) + res), the tail of
 * ResNetBlock.forward (src/net_utils.py:309-323).  n_pix = N*H*W. */
int rcf_bn_act_fwd(const float* z, const float* coef, const float* res, float* out,
                   long long n_pix, int c, int act, void* stream);
/* ..._amax variants (fp32 tensors): the same kernel also accumulates max|value written| into *amax (device, zeroed by the caller;
 * see rcf_amax) -- the per-tensor maximum RCF_PREC_F16X2 consumers scale their fp16 planes by, at no extra pass over the tensor. */
int rcf_bn_act_fwd_amax(const float* z, const float* coef, const float* res, float* out,
                        long long n_pix, int c, int act, float* amax, void* stream);

/* skip = sigmoid(BN_w(zw)) * BN_p(zp) + img : FusionNetEncoder 'weight_and_project' fusion (src/networks.py:863-866). */
int rcf_fuse_fwd(const float* zw, const float* coef_w, const float* zp, const float* coef_p,
                 const float* img, float* out, long long n_pix, int c, void* stream);
int rcf_fuse_fwd_amax(const float* zw, const float* coef_w, const float* zp, const float* coef_p,
                      const float* img, float* out, long long n_pix, int c, float* amax, void* stream);

/* Backward of rcf_bn_act_fwd, two passes around a per-channel reduction (BatchNorm2d backward).
 * reduce: partials[n_blocks][2][c] (fp64: these sums cancel heavily, and PyTorch's CPU BatchNorm accumulates float
 * tensors in double) = (sum g, sum g*xhat), g = dout * act'(.) (* lrelu'(out) when has_res).
 * n_blocks for a given (n_pix, c) comes from rcf_ew_blocks. */
int rcf_ew_blocks(long long n_pix, int c);
int rcf_bn_act_bwd_reduce(const float* dout, const float* z, const float* coef, const float* out,
                          double* partials, long long n_pix, int c, int act, int has_res, void* stream);
/* bcoef[2][c] = (sum g / M, sum g*xhat / M); dgamma[c], dbeta[c] (accumulate == 0: overwrite). */
int rcf_bn_bwd_finalize(const double* partials, int n_blocks, int partial_stride, int c, double count,
                        float* bcoef, float* dgamma, float* dbeta, void* stream);
/* dz = scale * (g - bcoef0 - xhat*bcoef1); dres (nullable) = dout*lrelu'(out), accumulated when dres_accumulate. */
int rcf_bn_act_bwd_apply(const float* dout, const float* z, const float* coef, const float* out,
                         const float* bcoef, float* dz, float* dres, int dres_accumulate,
                         long long n_pix, int c, int act, int has_res, void* stream);
/* amax: max|dz| (the operand of the layer's input- and weight-gradient convolutions). */
int rcf_bn_act_bwd_apply_amax(const float* dout, const float* z, const float* coef, const float* out,
                              const float* bcoef, float* dz, float* dres, int dres_accumulate,
                              long long n_pix, int c, int act, int has_res, float* amax, void* stream);

/* BatchNorm + LeakyReLU backward of the layer that feeds the output head (MultiScaleDecoder deconv0.conv -> output0,
 * src/networks.py:1548-1555, :1649-1654), fused with rcf_head_bwd_dgrad: dout is recomputed from dlogit (N,H,W) and the head
 * weight in both passes, so the C-channel gradient of the head's input is never materialised.  c <= 64.  partials
 * [rcf_head_bn_blocks][2][c] feed rcf_bn_bwd_finalize like those of rcf_bn_act_bwd_reduce. */
int rcf_head_bn_blocks(int n, int h, int w, int c);
int rcf_head_bn_bwd_reduce(const float* dlogit, const float* w_head, const float* z, const float* coef, double* partials,
                           int n, int h, int w, int c, void* stream);
int rcf_head_bn_bwd_apply(const float* dlogit, const float* w_head, const float* z, const float* coef, const float* bcoef,
                          float* dz, int n, int h, int w, int c, void* stream);
int rcf_head_bn_bwd_apply_amax(const float* dlogit, const float* w_head, const float* z, const float* coef, const float* bcoef,
                               float* dz, int n, int h, int w, int c, float* amax, void* stream);

/* Backward of rcf_fuse_fwd. partials[n_blocks][4][c]: (sum gw, sum gw*xhat_w, sum gp, sum gp*xhat_p). */
int rcf_fuse_bwd_reduce(const float* dout, const float* zw, const float* coef_w, const float* zp,
                        const float* coef_p, double* partials, long long n_pix, int c, void* stream);
int rcf_fuse_bwd_apply(const float* dout, const float* zw, const float* coef_w, const float* zp,
                       const float* coef_p, const float* bcoef_w, const float* bcoef_p,
                       float* dzw, float* dzp, float* dimg, int dimg_accumulate,
                       long long n_pix, int c, void* stream);

/* torch.nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (src/networks.py:392-395, :875-876).
 * idx: one byte per output element = winning tap (ky*3+kx), first maximum in scan order as in PyTorch. */
int rcf_maxpool3x3s2_fwd(const float* in, float* out, unsigned char* idx, int n, int h, int w, int c, void* stream);
int rcf_maxpool3x3s2_bwd(const float* dout, const unsigned char* idx, float* din, int din_accumulate,
                         int n, int h, int w, int c, void* stream);

/* Backward of the nearest upsample folded into rcf_conv2d_fwd: dsrc[n,sy,sx,:] = sum of dup over the fan-out. */
int rcf_upsample_nearest_bwd(const float* dup, float* dsrc, int dsrc_accumulate,
                             int n, int h_up, int w_up, int h_src, int w_src, int c, void* stream);

/* output0 (3x3, C -> 1, no BN, no activation; src/networks.py:1548-1555) fused with the depth map
 * d = min/(sigmoid(o)+min/max) of FusionNetModel.forward (src/fusionnet_model.py:162-165).
 * w: [1][C][3][3] OIHW.  logit and depth: [N][H][W]. */
int rcf_head_fwd(const float* x, const float* w, float* logit, float* depth,
                 int n, int h, int w_, int c, float min_depth, float max_depth, void* stream);
/* Same with the previous block's BatchNorm + LeakyReLU applied on load: z is that block's RAW conv output and coef its
 * rcf_bn_finalize coefficients, so its activation tensor never has to be written (c <= 64). */
int rcf_head_fwd_bn(const float* z, const float* coef, const float* w, float* logit, float* depth, int n, int h, int w_, int c,
                    float min_depth, float max_depth, void* stream);
/* dlogit = ddepth * d(depth)/d(logit). */
int rcf_head_bwd_logit(const float* ddepth, const float* logit, float* dlogit, long long n_pix,
                       float min_depth, float max_depth, void* stream);
int rcf_head_bwd_dgrad(const float* dlogit, const float* w, float* dx, int n, int h, int w_, int c, void* stream);
size_t rcf_head_wgrad_workspace_floats(int n, int h, int w_, int c);
int rcf_head_bwd_wgrad(const float* x, const float* dlogit, float* dw, float* workspace,
                       int n, int h, int w_, int c, void* stream);
int rcf_head_bwd_wgrad_bn(const float* z, const float* coef, const float* dlogit, float* dw, float* workspace, int n, int h,
                          int w_, int c, void* stream);

/* Masked L1 of FusionNetModel.compute_loss, loss_func='l1' (src/fusionnet_model.py:209-253;
 * src/fusionnet_losses.py:19-32): ground truth is zeroed where lidar > 0, then
 * loss = mean|d-gt| over gt>0 + w_lidar * mean|d-lidar| over lidar>0.
 * sums (device, double[4]) = (sum|d-gt|, count gt, sum|d-lidar|, count lidar): local to this rank, so a
 * data-parallel caller can all-reduce it before the backward.  workspace: rcf_loss_workspace_floats. */
size_t rcf_loss_workspace_floats(long long n_pix);
int rcf_l1_loss_fwd(const float* depth, const float* gt, const float* lidar, float* workspace,
                    double* sums, long long n_pix, void* stream);
/* loss[0..2] = total, supervised, lidar from (possibly all-reduced) sums. */
int rcf_l1_loss_value(const double* sums, float w_lidar, float* loss, void* stream);
/* ddepth = upstream * (sign(d-gt)/count_gt [gt>0] + w_lidar*sign(d-lidar)/count_lidar [lidar>0]); upstream nullable (=1). */
int rcf_l1_loss_bwd(const float* depth, const float* gt, const float* lidar, const double* sums,
                    const float* upstream, float w_lidar, float* ddepth, long long n_pix, void* stream);

/* The same masked loss for loss_func 'l1' / 'l2' / 'smoothl1' (src/fusionnet_model.py:245-275: F.l1_loss, F.mse_loss, F.smooth_l1_loss
 * with reduction 'mean' over the valid pixels, src/fusionnet_losses.py:4-46): `kind` selects the per-pixel term (|e|, e^2, Huber with
 * beta 1) and its slope; sums / workspace / upstream as above; rcf_l1_loss_value turns the sums into the loss for every kind. */
#define RCF_LOSS_L1 0
#define RCF_LOSS_L2 1
#define RCF_LOSS_SMOOTH_L1 2
int rcf_masked_loss_fwd(const float* depth, const float* gt, const float* lidar, float* workspace,
                        double* sums, long long n_pix, int kind, void* stream);
int rcf_masked_loss_bwd(const float* depth, const float* gt, const float* lidar, const double* sums,
                        const float* upstream, float w_lidar, float* ddepth, long long n_pix, int kind, void* stream);

/* Local smoothness term of compute_loss (w_smoothness > 0, loss_smoothness_kernel_size <= 1: src/fusionnet_model.py:277-281 ->
 * losses.smoothness_loss_func, src/fusionnet_losses.py:48-72): edge-aware |dP/dx|, |dP/dy| weighted by exp(-mean_c |dI|).
 * image: N x C x H x W (the public NCHW tensor), depth: [N][H][W].  sums (device, double[4]) = (sum x, count x, sum y, count y), local
 * to this rank (all-reduce before the backward under data parallelism); value = sums[0]/sums[1] + sums[2]/sums[3].
 * _bwd ADDS upstream * w_smoothness * d(value)/d(depth) into ddepth (after rcf_masked_loss_bwd wrote it). */
int rcf_smoothness_loss_fwd(const float* image_nchw, const float* depth, float* workspace, double* sums, int n, int c, int h, int w,
                            void* stream);
int rcf_smoothness_loss_bwd(const float* image_nchw, const float* depth, const double* sums, const float* upstream, float w_smoothness,
                            float* ddepth, int n, int c, int h, int w, void* stream);

/* OutlierRemoval.remove_outliers (src/net_utils.py:591-638; called on the ground truth every training step,
 * src/fusionnet_main.py:377-378): a valid point (depth > 0) is zeroed when some valid point in its k x k window is more
 * than `threshold` metres closer.  depth, out: [N][H][W] (single channel).  scratch: one float (device), used for the
 * global maximum the reference fills invalid pixels with (10 * max(depth)); kernel_size odd, <= 15. */
int rcf_outlier_removal(const float* depth, float* out, float* scratch, int n, int h, int w, int kernel_size,
                        float threshold, void* stream);

/* torch.optim.Adam step (src/fusionnet_main.py:307-312, :399) over one flat parameter arena.
 * step is the 1-based step count; weight_decay is the L2 form Adam uses (added to the gradient). */
int rcf_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, void* stream);

/* The same update with the step count and the hyper-parameters in DEVICE memory, so a recorded launch sequence (hipGraph) performs
 * a correct optimizer.step() on every replay: state (device float[8]) = {step count so far, lr, beta1, beta2, eps, weight_decay,
 * scratch, scratch}.  The call advances state[0] by one, derives the bias corrections from it (in double, like the host path) and
 * applies torch.optim.Adam's update (src/fusionnet_main.py:399).  The host changes lr etc. by writing state[1..5]. */
int rcf_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, float* state, void* stream);

/* NCHW (the reference's public tensor layout) <-> NHWC (kernel layout). */
int rcf_nchw_to_nhwc(const float* in, float* out, int n, int c, int h, int w, void* stream);
int rcf_nhwc_to_nchw(const float* in, float* out, int n, int c, int h, int w, void* stream);

/* Radar point -> dense map scatter of radarnet_main.forward (src/radarnet_main.py:563-589):
 * crops [K][H][Wc] are sigmoid responses; values < 0.5 are zeroed, crop k is pasted at columns
 * [x_k - Wc/2, x_k + Wc/2) of a W-wide canvas, the maximum response and its argmax point are taken per
 * pixel, and the argmax is replaced by that point's depth z_k.  strict_reference != 0 reproduces the
 * reference's int64 in-place replacement chain (truncation and re-replacement quirk);
 * 0 gives depth = z[argmax].  points: [K][3] (x px, y px, z m).  depth/response: [H][W]. */
int rcf_radar_scatter(const float* crops, const float* points, int k, int h, int w, int wc,
                      int strict_reference, float* depth, float* response, void* stream);
/* The same from the correspondence LOGITS (what RadarNetModel.forward(return_logits=True) returns; src/radarnet_main.py:556-567 applies
 * torch.sigmoid and then zeroes responses < 0.5): the threshold is taken on the sign of the logit -- sigmoid(l) >= 0.5 exactly when
 * l >= 0 -- and response = 1 / (1 + expf(-l)) of the survivors, so the keep / drop decision of a pixel cannot depend on an ulp of this
 * device's sigmoid.  (The reference's fp32 CPU sigmoid rounds logits in (-6e-8, 0) up to 0.5 and keeps them; this entry drops them.) */
int rcf_radar_scatter_logits(const float* logits, const float* points, int k, int h, int w, int wc,
                             int strict_reference, float* depth, float* response, void* stream);

/* ---- RadarNet stage 1: ops FusionNet does not have --------------------------------------------------------- */

/* torchvision.ops.roi_pool as RadarNetV1Encoder.forward calls it (src/networks.py:1232-1247; torchvision 0.11.3 semantics
 * restated, see oracle/roi_pool_oracle.py): in (N,H,W,C) NHWC, rois (R,5) = (batch index, x1, y1, x2, y2) in input-image
 * coordinates scaled by spatial_scale; out (R,PH,PW,out_cstride) written at channel offset out_coff (so the pooled latent can be
 * placed next to the radar latent without a concat); argmax (R,PH,PW,C) int32 = y*W+x of each maximum, -1 for an empty bin. */
int rcf_roi_pool_fwd(const float* in, const float* rois, float* out, int* argmax, int n_roi, int n, int h, int w, int c,
                     int pooled_h, int pooled_w, float spatial_scale, int out_cstride, int out_coff, void* stream);
/* din (N,H,W,C) += scatter of dout through argmax (rois overlap: atomic adds; zero din first unless accumulating). */
int rcf_roi_pool_bwd(const float* dout, const int* argmax, const float* rois, float* din, int n_roi, int n, int h, int w, int c,
                     int pooled_h, int pooled_w, int dout_cstride, int dout_coff, void* stream);
/* The same gradient as a gather (one thread per input pixel walks the rois of its image and the bins containing the pixel): no
 * atomics, din (N,H,W,C) is written once -- overwritten, or added to when din_accumulate -- in a fixed summation order; needs the
 * forward's spatial_scale to rebuild the bin edges.  n_roi <= 1024 (RCF_EUNSUPPORTED beyond: use rcf_roi_pool_bwd).  The _b16 twin
 * takes dout AND din as bf16. */
int rcf_roi_pool_bwd_gather(const float* dout, const int* argmax, const float* rois, float* din, int din_accumulate, int n_roi, int n,
                            int h, int w, int c, int pooled_h, int pooled_w, float spatial_scale, int dout_cstride, int dout_coff,
                            void* stream);

/* net_utils.FullyConnected (src/net_utils.py:201-247): y = act(x W^T + b), x (M,n_in), W (n_out,n_in), any M >= 1
 * (rows are processed in blocks of 64 inside the call; dW and db are summed over all rows).
 * act: 0 linear, 1 LeakyReLU(0.2).  hw > 1: feature f = c*hw + p is stored at NHWC position m*(hw*cstride) + p*cstride + coff + c
 * (the .view(M, C, -1, W) + torch.cat of src/networks.py:1251-1255 done by addressing); hw <= 1: plain (M,n_out). */
int rcf_fc_fwd(const float* x, const float* w, const float* bias, float* y, int m_rows, int n_in, int n_out, int act,
               int hw, int cstride, int coff, void* stream);
size_t rcf_fc_bwd_workspace_floats(int m_rows, int n_in, int n_out);
/* dw (n_out,n_in), db (n_out) overwritten; dx (M,n_in) nullable (then workspace may be null too). y / dy use the forward layout. */
int rcf_fc_bwd(const float* x, const float* w, const float* y, const float* dy, float* dw, float* db, float* dx,
               float* workspace, int m_rows, int n_in, int n_out, int act, int hw, int cstride, int coff, void* stream);

/* RadarNetModel.compute_loss (src/radarnet_model.py:131-171): sum(valid * BCEWithLogits(logit, target, pos_weight)) / sum(valid).
 * sums[2] (fp64) = (weighted loss sum, valid count) feeds the backward; loss[1] fp32. */
size_t rcf_bce_workspace_doubles(void);
int rcf_bce_loss_fwd(const float* logit, const float* target, const float* valid, double* workspace, double* sums, float* loss,
                     long long n, float pos_weight, void* stream);
int rcf_bce_loss_bwd(const float* logit, const float* target, const float* valid, const double* sums, const float* upstream,
                     float* dlogit, long long n, float pos_weight, void* stream);

/* ---- input augmentation: fusionnet_transforms.Transforms.transform, src/fusionnet_transforms.py:46-178 ------- */

/* Per-sample brightness / contrast / saturation (torchvision.transforms.functional.adjust_* restated), float conversion, range
 * normalisation (norm_mode 0: [0,255], 1: [0,1], 2: [-1,1]) and flips of a batch of RGB images (N,3,H,W).  do_* are N bytes
 * (nullable: never), f_* N floats, all on the device; the 0..255-vs-0..1 decision of :81-83 is taken on the device. */
size_t rcf_transform_workspace_bytes(int n);
int rcf_transform_images(const float* img, float* out, int n, int h, int w, const unsigned char* do_brightness,
                         const float* f_brightness, const unsigned char* do_contrast, const float* f_contrast,
                         const unsigned char* do_saturation, const float* f_saturation, const unsigned char* do_hflip,
                         const unsigned char* do_vflip, int norm_mode, void* workspace, void* stream);
/* Per-sample horizontal / vertical flips of range maps (N,C,H,W) (src/fusionnet_transforms.py:139-163). */
int rcf_transform_flip(const float* in, float* out, int n, int c, int h, int w, const unsigned char* do_hflip,
                       const unsigned char* do_vflip, void* stream);

/* ---- on-disk sample formats finished on the device: src/data_utils.py:167-335, src/datasets.py:19-109 ------------ */

#define RCF_PIXEL_U8 0   /* 8-bit PNG */
#define RCF_PIXEL_U16 1  /* 16-bit PNG (PIL mode I;16): what save_depth / save_response write */
#define RCF_PIXEL_I32 2  /* PIL mode I */

/* load_image (src/data_utils.py:167-198) + the crop of datasets.random_crop (src/datasets.py:101-109) for a batch: RGB bytes
 * (n, src_h, src_w, 3) -> float32 (n, 3, h, w), rows/columns [y0, y0 + h) x [x0, x0 + w) with (y0, x0) = crop_yx[2b], crop_yx[2b+1]
 * (device ints; NULL: no offset); normalize != 0 divides by 255.0 like load_image(normalize=True). */
int rcf_decode_image_u8(const unsigned char* src, float* dst, int n, int src_h, int src_w, int h, int w, const int* crop_yx, int normalize,
                        void* stream);
/* load_depth / load_depth_with_validity_map / load_response (src/data_utils.py:200-269, :288-318) + crop for a batch of integer
 * maps (n, src_h, src_w) of type src_type -> float32 (n, 1, h, w): z = float(pixel) / multiplier; clamp_nonpositive != 0 applies
 * `z[z <= 0] = 0` (the depth loaders; load_response does not); validity (nullable) receives 1.0 where z > 0. */
int rcf_decode_map(const void* src, int src_type, float* dst, float* validity, int n, int src_h, int src_w, int h, int w,
                   const int* crop_yx, float multiplier, int clamp_nonpositive, void* stream);
/* save_depth / save_response (src/data_utils.py:271-286, :320-335): out = np.uint32(z * multiplier), the array PIL then writes. */
int rcf_encode_map_u32(const float* z, unsigned* out, long long count, float multiplier, void* stream);
/* points_to_depth_map (setup/setup_dataset_nuscenes_with_denseGT.py:814-840): depth_map[round(y_k), round(x_k)] = depth[k] in point
 * order (the last point of a pixel wins; np.round = half to even; negative indices wrap once like numpy's).  workspace:
 * rcf_points_to_depth_map_workspace_bytes(h, w); its last int holds the number of points that fell outside the image (numpy raises
 * IndexError for those; here they are skipped and counted). */
size_t rcf_points_to_depth_map_workspace_bytes(int h, int w);
int rcf_points_to_depth_map(const float* xs, const float* ys, const float* depth, int n_points, float* depth_map, int h, int w,
                            void* workspace, void* stream);

/* ---- bf16 NHWC tensors (BASELINE.json configs 2-4) ----------------------------------------------------------------------------
 * Every entry point above that reads or writes an NHWC activation / gradient tensor has a twin NAME_b16 with the SAME argument
 * list in which those tensors hold bf16 elements (2 bytes, round-to-nearest-even on store) instead of fp32; everything else --
 * weights, BatchNorm partial sums / coefficients, the single-channel logit / depth / dlogit maps, argmax and pool indices,
 * workspaces, the radar points and the fully connected activations -- keeps the type written in the fp32 declaration.  Arithmetic
 * is fp32 on the loaded values.  The convolutions take the storage in rcf_conv_desc.storage instead of a twin.
 * Which arguments are bf16 in the twin:
 *   rcf_bn_act_fwd_b16 z, res, out | rcf_fuse_fwd_b16 zw, zp, img, out | rcf_bn_act_bwd_reduce_b16 dout, z, out |
 *   rcf_bn_act_bwd_apply_b16 dout, z, out, dz, dres | rcf_fuse_bwd_reduce_b16 dout, zw, zp | rcf_fuse_bwd_apply_b16 dout, zw, zp,
 *   dzw, dzp, dimg | rcf_head_bn_bwd_reduce_b16 z | rcf_head_bn_bwd_apply_b16 z, dz | rcf_maxpool3x3s2_fwd_b16 in, out |
 *   rcf_maxpool3x3s2_bwd_b16 dout, din | rcf_upsample_nearest_bwd_b16 dup, dsrc | rcf_head_fwd_b16 x | rcf_head_fwd_bn_b16 z |
 *   rcf_head_bwd_dgrad_b16 dx | rcf_head_bwd_wgrad_b16 x | rcf_head_bwd_wgrad_bn_b16 z | rcf_roi_pool_fwd_b16 in, out |
 *   rcf_roi_pool_bwd_b16 dout (din stays fp32: the scatter uses fp32 atomics; add it into a bf16 gradient with rcf_convert) |
 *   rcf_roi_pool_bwd_gather_b16 dout, din |
 *   rcf_fc_fwd_b16 y | rcf_fc_bwd_b16 y, dy. */
int rcf_bn_act_fwd_b16(const float* z, const float* coef, const float* res, float* out, long long n_pix, int c, int act, void* stream);
int rcf_fuse_fwd_b16(const float* zw, const float* coef_w, const float* zp, const float* coef_p, const float* img, float* out,
                     long long n_pix, int c, void* stream);
int rcf_bn_act_bwd_reduce_b16(const float* dout, const float* z, const float* coef, const float* out, double* partials, long long n_pix,
                              int c, int act, int has_res, void* stream);
int rcf_bn_act_bwd_apply_b16(const float* dout, const float* z, const float* coef, const float* out, const float* bcoef, float* dz,
                             float* dres, int dres_accumulate, long long n_pix, int c, int act, int has_res, void* stream);
/* Inference form of the 'weight_and_project' fusion (src/networks.py:863-866 with the BatchNorms in eval mode) on bf16 tensors, one
 * pass:  out = sigmoid(scale_w * (W1 d) + shift_w) * (scale_p * (W2 d) + shift_p) + img.   d [n_pix][c_d], img / out [n_pix][c_i] bf16
 * NHWC; w1, w2: the 1x1 convolutions' fp32 OIHW weights [c_i][c_d]; coef_w, coef_p: rcf_bn_finalize coefficient tables [4][c_i] (rows
 * scale, shift used).  Replaces two rcf_conv2d_fwd + rcf_fuse_fwd_b16 when no batch statistics are needed.  c_d in {16, 32, 64, 128},
 * c_i even (rcf_fuse_wp_infer_supported returns 1); anything else: RCF_EUNSUPPORTED. */
int rcf_fuse_wp_infer_supported(int c_d, int c_i);
int rcf_fuse_wp_infer_b16(const float* d, const float* w1, const float* coef_w, const float* w2, const float* coef_p, const float* img,
                          float* out, long long n_pix, int c_d, int c_i, void* stream);
int rcf_fuse_bwd_reduce_b16(const float* dout, const float* zw, const float* coef_w, const float* zp, const float* coef_p,
                            double* partials, long long n_pix, int c, void* stream);
int rcf_fuse_bwd_apply_b16(const float* dout, const float* zw, const float* coef_w, const float* zp, const float* coef_p,
                           const float* bcoef_w, const float* bcoef_p, float* dzw, float* dzp, float* dimg, int dimg_accumulate,
                           long long n_pix, int c, void* stream);
int rcf_head_bn_bwd_reduce_b16(const float* dlogit, const float* w_head, const float* z, const float* coef, double* partials, int n,
                               int h, int w, int c, void* stream);
int rcf_head_bn_bwd_apply_b16(const float* dlogit, const float* w_head, const float* z, const float* coef, const float* bcoef, float* dz,
                              int n, int h, int w, int c, void* stream);
int rcf_maxpool3x3s2_fwd_b16(const float* in, float* out, unsigned char* idx, int n, int h, int w, int c, void* stream);
int rcf_maxpool3x3s2_bwd_b16(const float* dout, const unsigned char* idx, float* din, int din_accumulate, int n, int h, int w, int c,
                             void* stream);
int rcf_upsample_nearest_bwd_b16(const float* dup, float* dsrc, int dsrc_accumulate, int n, int h_up, int w_up, int h_src, int w_src,
                                 int c, void* stream);
int rcf_head_fwd_b16(const float* x, const float* w, float* logit, float* depth, int n, int h, int w_, int c, float min_depth,
                     float max_depth, void* stream);
int rcf_head_fwd_bn_b16(const float* z, const float* coef, const float* w, float* logit, float* depth, int n, int h, int w_, int c,
                        float min_depth, float max_depth, void* stream);
int rcf_head_bwd_dgrad_b16(const float* dlogit, const float* w, float* dx, int n, int h, int w_, int c, void* stream);
int rcf_head_bwd_wgrad_b16(const float* x, const float* dlogit, float* dw, float* workspace, int n, int h, int w_, int c, void* stream);
int rcf_head_bwd_wgrad_bn_b16(const float* z, const float* coef, const float* dlogit, float* dw, float* workspace, int n, int h, int w_,
                              int c, void* stream);
int rcf_roi_pool_bwd_gather_b16(const float* dout, const int* argmax, const float* rois, float* din, int din_accumulate, int n_roi, int n,
                                int h, int w, int c, int pooled_h, int pooled_w, float spatial_scale, int dout_cstride, int dout_coff,
                                void* stream);
int rcf_roi_pool_fwd_b16(const float* in, const float* rois, float* out, int* argmax, int n_roi, int n, int h, int w, int c,
                         int pooled_h, int pooled_w, float spatial_scale, int out_cstride, int out_coff, void* stream);
int rcf_roi_pool_bwd_b16(const float* dout, const int* argmax, const float* rois, float* din, int n_roi, int n, int h, int w, int c,
                         int pooled_h, int pooled_w, int dout_cstride, int dout_coff, void* stream);
int rcf_fc_fwd_b16(const float* x, const float* w, const float* bias, float* y, int m_rows, int n_in, int n_out, int act, int hw,
                   int cstride, int coff, void* stream);
int rcf_fc_bwd_b16(const float* x, const float* w, const float* y, const float* dy, float* dw, float* db, float* dx, float* workspace,
                   int m_rows, int n_in, int n_out, int act, int hw, int cstride, int coff, void* stream);
/* The two 7x7 stride-2 stem convolutions (FusionNetEncoder conv1_image / conv1_depth, src/networks.py:332-352, :854-855; ResNetEncoder
 * conv1, :70-78) with bf16 tensors: a 7-tap kernel at stride 2 is a 4-tap kernel at stride 1 on the space-to-depth image whose 16
 * channels are the 2 x 2 pixel phases x (up to) 4 input channels.  rcf_s2d_image_b16 builds that image straight from the reference's
 * NCHW fp32 input (replacing rcf_nchw_to_nhwc for the forward pass): out[n][y][x][a*8 + b*4 + c] = bf16(img[n][c][2y + a][2x + b]),
 * (ceil(h/2), ceil(w/2)) pixels, zero where the source ends.  rcf_stem_weights_s2d rewrites the OIHW 7x7 weight as an OIHW 4x4 weight
 * over those 16 channels.  The convolution is then rcf_conv2d_fwd with {ksize 4, stride 1, pad 2, pad_x 2, c1 16, storage BF16, h_in /
 * w_in = the space-to-depth extent, h_out / w_out = the stem's output extent}.  Its weight gradient is taken on the 7x7 form. */
int rcf_s2d_image_b16(const float* img_nchw, void* out, int n, int c, int h, int w, void* stream);
/* The same image in fp32, [n][ceil(h/2)][ceil(w/2)][16]: the stems of the fp32 configuration on the two-plane fp16 arithmetic
 * (rcf_conv2d_fwd_scaled with {ksize 4, ..., precision RCF_PREC_F16X2, storage RCF_STORE_FP32}).  amax (nullable): a zeroed device
 * scalar that receives max|pixel| (see rcf_amax). */
int rcf_s2d_image_f32(const float* img_nchw, float* out, int n, int c, int h, int w, float* amax, void* stream);
int rcf_stem_weights_s2d(const float* w7_oihw, float* w4_oihw, int c_out, int c_in, void* stream);

/* dst[i] = (accumulate ? dst[i] : 0) + src[i] for n elements, each side RCF_STORE_FP32 or RCF_STORE_BF16 (torch's .to(dtype) of the
 * reference-side glue; also how an fp32 scatter result joins a bf16 gradient). */
int rcf_convert(const void* src, int src_storage, void* dst, int dst_storage, long long n, int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif
