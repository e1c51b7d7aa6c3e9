#include "conv2d_f32_kernel.h"
// instantiation group of the fp32-MFMA kernel (tile shapes WC, WP, TC, TP)
int accflow_launch_conv_f32_2212(const accflow_conv_desc& d, hipStream_t st) { return launch_conv<2, 2, 1, 2>(d, st); }
int accflow_launch_conv_f32_2211(const accflow_conv_desc& d, hipStream_t st) { return launch_conv<2, 2, 1, 1>(d, st); }
