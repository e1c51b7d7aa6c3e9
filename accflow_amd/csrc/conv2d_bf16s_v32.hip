#include "conv2d_bf16s_kernel.h"
// instantiation group of the im2col kernel: 6 x 32 output channels x 2 x 64 pixels per workgroup (all arithmetic modes)
int accflow_launch_conv_bf16s_32(const accflow_conv_desc& d, hipStream_t st) { return launch_conv_bf16s<3, 2>(d, st); }
