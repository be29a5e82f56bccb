// bf16 NHWC tensors (rcf_conv_desc.storage == RCF_STORE_BF16): the same kernels instantiated on 2-byte activations; reached
// through the public entry points of rcf_conv.hip (see rcf_conv_impl.h)
#define RCF_CONV_B16 1
#include "rcf_conv_impl.h"
