/*
 * Host-side sanitizer harness for the C ABI of librcf_hip.so ("compile-time -fsanitize=address host harness
 * for the C ABI").  Plain C: it also proves include/rcf_hip.h is a C header.  Built and run by tests/test_host_logic.py with
 *     gcc -std=c99 -fsanitize=address,undefined -I include tests/abi/abi_harness.c -ldl -o build/abi_harness
 * It needs no GPU: every call below must be REJECTED by argument validation (RCF_EINVAL / RCF_EUNSUPPORTED) before anything is
 * enqueued, must not read through the (null or poisoned) pointers it is given, and must leave the caller's descriptors untouched.
 * Exit code 0 = all checks passed; ASan / UBSan abort otherwise.
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rcf_hip.h"

static int failures = 0;
#define EXPECT(expr, want)                                                                      \
    do {                                                                                        \
        long long got_ = (long long)(expr);                                                     \
        if (got_ != (long long)(want)) {                                                        \
            fprintf(stderr, "FAIL %s:%d: %s = %lld, want %lld\n", __FILE__, __LINE__, #expr, got_, (long long)(want)); \
            ++failures;                                                                         \
        }                                                                                       \
    } while (0)

#define LOAD(name) \
    name##_t p_##name = (name##_t)dlsym(lib, #name); \
    if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }

typedef const char* (*rcf_version_t)(void);
typedef int (*rcf_conv2d_query_t)(const rcf_conv_desc*, rcf_conv_info*);
typedef int (*rcf_conv2d_fwd_t)(const rcf_conv_desc*, const float*, const float*, const float*, float*, double*, void*);
typedef int (*rcf_conv2d_wgrad_t)(const rcf_conv_desc*, const float*, const float*, const float*, float*, float*, void*);
typedef int (*rcf_conv2d_pack_weights_t)(const rcf_conv_desc*, const float*, float*, void*);
typedef int (*rcf_bn_act_fwd_t)(const float*, const float*, const float*, float*, long long, int, int, void*);
typedef int (*rcf_maxpool3x3s2_fwd_t)(const float*, float*, unsigned char*, int, int, int, int, void*);
typedef int (*rcf_roi_pool_fwd_t)(const float*, const float*, float*, int*, int, int, int, int, int, int, int, float, int, int, void*);
typedef int (*rcf_fc_fwd_t)(const float*, const float*, const float*, float*, int, int, int, int, int, int, int, void*);
typedef size_t (*rcf_fc_bwd_workspace_floats_t)(int, int, int);
typedef int (*rcf_outlier_removal_t)(const float*, float*, float*, int, int, int, int, float, void*);
typedef int (*rcf_adam_step_t)(float*, const float*, float*, float*, long long, float, float, float, float, float, int, void*);
typedef int (*rcf_radar_scatter_t)(const float*, const float*, int, int, int, int, int, float*, float*, void*);
typedef int (*rcf_head_fwd_t)(const float*, const float*, float*, float*, int, int, int, int, float, float, void*);
typedef int (*rcf_ew_blocks_t)(long long, int);
typedef size_t (*rcf_loss_workspace_floats_t)(long long);

static rcf_conv_desc good_desc(void) {
    rcf_conv_desc d;
    memset(&d, 0, sizeof d);
    d.n = 1; d.h_in = 32; d.w_in = 32; d.c1 = 16; d.c2 = 0; d.h_src1 = 32; d.w_src1 = 32; d.gather1 = RCF_GATHER_DIRECT;
    d.h_out = 32; d.w_out = 32; d.c_out = 32; d.ksize = 3; d.stride = 1; d.pad = 1; d.pad_x = 1; d.w_mode = RCF_W_FORWARD;
    d.w_o = 32; d.w_i = 16; d.out_stride = 1; d.out_h_phys = 32; d.out_w_phys = 32; d.precision = RCF_PREC_FP32;
    return d;
}

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "radar-camera-fusion-depth_amd/librcf_hip.so";
    void* lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen(%s): %s\n", path, dlerror()); return 2; }
    LOAD(rcf_version) LOAD(rcf_conv2d_query) LOAD(rcf_conv2d_fwd) LOAD(rcf_conv2d_wgrad) LOAD(rcf_conv2d_pack_weights)
    LOAD(rcf_bn_act_fwd) LOAD(rcf_maxpool3x3s2_fwd) LOAD(rcf_roi_pool_fwd) LOAD(rcf_fc_fwd) LOAD(rcf_fc_bwd_workspace_floats)
    LOAD(rcf_outlier_removal) LOAD(rcf_adam_step) LOAD(rcf_radar_scatter) LOAD(rcf_head_fwd) LOAD(rcf_ew_blocks)
    LOAD(rcf_loss_workspace_floats)

    if (strncmp(p_rcf_version(), "rcf_hip", 7) != 0) { fprintf(stderr, "bad version string\n"); ++failures; }

    /* a heap block with ASan redzones: any read or write through it by a call that should have been rejected is reported */
    float* poison = (float*)malloc(4);
    rcf_conv_info info;
    rcf_conv_desc d = good_desc(), keep = d;

    /* null descriptor / null info / null buffers */
    EXPECT(p_rcf_conv2d_query(NULL, &info), RCF_EINVAL);
    EXPECT(p_rcf_conv2d_query(&d, NULL), RCF_EINVAL);
    EXPECT(p_rcf_conv2d_fwd(&d, NULL, NULL, poison, poison, NULL, NULL), RCF_EINVAL);
    EXPECT(p_rcf_conv2d_fwd(NULL, poison, NULL, poison, poison, NULL, NULL), RCF_EINVAL);
    EXPECT(p_rcf_conv2d_wgrad(&d, poison, NULL, NULL, poison, poison, NULL), RCF_EINVAL);
    EXPECT(p_rcf_conv2d_pack_weights(&d, NULL, poison, NULL), RCF_EINVAL);
    /* inconsistent descriptors */
    d = good_desc(); d.ksize = 5;
    EXPECT(p_rcf_conv2d_query(&d, &info), RCF_EINVAL);
    d = good_desc(); d.h_out = 31;                         /* does not follow from h_in, pad, stride */
    EXPECT(p_rcf_conv2d_query(&d, &info), RCF_EINVAL);
    d = good_desc(); d.w_i = 17;                           /* weight tensor does not match c1 + c2 */
    EXPECT(p_rcf_conv2d_query(&d, &info), RCF_EINVAL);
    d = good_desc(); d.n = -3;
    EXPECT(p_rcf_conv2d_fwd(&d, poison, NULL, poison, poison, NULL, NULL), RCF_EINVAL);
    d = good_desc(); d.precision = 7;
    EXPECT(p_rcf_conv2d_query(&d, &info), RCF_EINVAL);
    /* valid but outside what the kernels implement */
    d = good_desc(); d.c_out = 5; d.w_o = 5;
    EXPECT(p_rcf_conv2d_query(&d, &info), RCF_EUNSUPPORTED);
    EXPECT(p_rcf_conv2d_fwd(&d, poison, NULL, poison, poison, NULL, NULL), RCF_EUNSUPPORTED);
    d = good_desc();
    if (memcmp(&d, &keep, sizeof d) != 0) { fprintf(stderr, "descriptor modified\n"); ++failures; }

    EXPECT(p_rcf_bn_act_fwd(poison, poison, NULL, poison, 0, 32, RCF_ACT_LEAKY_RELU, NULL), RCF_EINVAL);
    EXPECT(p_rcf_bn_act_fwd(NULL, poison, NULL, poison, 10, 32, RCF_ACT_LEAKY_RELU, NULL), RCF_EINVAL);
    EXPECT(p_rcf_ew_blocks(1000, 3) <= 0, 1);              /* channel count the elementwise kernels do not cover */
    EXPECT(p_rcf_maxpool3x3s2_fwd(poison, poison, NULL, 1, 8, 8, 4, NULL), RCF_EINVAL);
    EXPECT(p_rcf_maxpool3x3s2_fwd(poison, poison, (unsigned char*)poison, 1, 0, 8, 4, NULL), RCF_EINVAL);
    EXPECT(p_rcf_roi_pool_fwd(poison, poison, poison, (int*)poison, 1, 1, 8, 8, 3, 2, 2, 0.5f, 4, 0, NULL), RCF_EUNSUPPORTED);
    EXPECT(p_rcf_roi_pool_fwd(poison, NULL, poison, (int*)poison, 1, 1, 8, 8, 4, 2, 2, 0.5f, 4, 0, NULL), RCF_EINVAL);
    EXPECT(p_rcf_fc_fwd(poison, poison, poison, poison, 0, 3, 32, 1, 1, 0, 0, NULL), RCF_EINVAL);
    EXPECT(p_rcf_fc_fwd(poison, poison, poison, poison, 4, 3, 30, 1, 7, 64, 0, NULL), RCF_EUNSUPPORTED);   /* 30 features do not tile hw = 7 */
    EXPECT(p_rcf_fc_bwd_workspace_floats(-1, 3, 32), 0);
    EXPECT(p_rcf_fc_bwd_workspace_floats(100, 3, 300), (size_t)5 * 100 * 3);   /* one partial [m][k] per 64 output features */
    EXPECT(p_rcf_outlier_removal(poison, poison, poison, 1, 8, 8, 6, 1.5f, NULL) != RCF_OK, 1);              /* even window */
    EXPECT(p_rcf_outlier_removal(poison, NULL, poison, 1, 8, 8, 7, 1.5f, NULL), RCF_EINVAL);
    EXPECT(p_rcf_adam_step(poison, poison, poison, poison, 0, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.f, 1, NULL), RCF_EINVAL);
    EXPECT(p_rcf_adam_step(poison, NULL, poison, poison, 16, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.f, 1, NULL), RCF_EINVAL);
    EXPECT(p_rcf_radar_scatter(poison, poison, 0, 8, 8, 4, 1, poison, poison, NULL), RCF_EINVAL);
    EXPECT(p_rcf_head_fwd(poison, poison, poison, NULL, 1, 8, 8, 32, 1.f, 100.f, NULL), RCF_EINVAL);
    EXPECT(p_rcf_loss_workspace_floats(1000) > 0, 1);

    free(poison);
    if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
    printf("abi_harness: all checks passed (%s)\n", p_rcf_version());
    return 0;
}
