// fp32 instantiation of the normalisation / activation kernels (BASELINE config C4's arithmetic type): norm_act.hip with
// 4-byte elements, exported as danet_bn_forward_f32, danet_bn_backward_f32, danet_bn_forward_multi_f32,
// danet_bn_backward_multi_f32, danet_sum_relu_forward_f32, danet_sum_relu_backward_f32, danet_sum_relu_backward_all_f32
// (same arguments as the bf16 entry points; activations fp32 NHWC).  The one-pass backward stays bf16-only.
#define NA_F32 1
#include "norm_act.hip"
