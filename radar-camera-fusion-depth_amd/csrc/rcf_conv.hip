// fp32 NHWC tensors: the public extern "C" entry points of the convolution family (see rcf_conv_impl.h)
#define RCF_CONV_B16 0
#include "rcf_conv_impl.h"
