#include "conv2d_bf16s_kernel.h"
// instantiation group of the im2col kernel: 4 x 32 output channels x 1 x 64 pixels per workgroup (all arithmetic modes)
int accflow_launch_conv_bf16s_21(const accflow_conv_desc& d, hipStream_t st) { return launch_conv_bf16s<2, 1>(d, st); }
