// fp32-input MFMA implicit-GEMM convolution: dispatcher over the tile shapes (kernel template: conv2d_f32_kernel.h; instantiation
// groups: this unit and conv2d_f32_v1 / _v2 / _v3.hip).
#include "conv2d_f32_kernel.h"

int accflow_launch_conv_f32_1412(const accflow_conv_desc& d, hipStream_t st);
int accflow_launch_conv_f32_1411(const accflow_conv_desc& d, hipStream_t st);
int accflow_launch_conv_f32_2212(const accflow_conv_desc& d, hipStream_t st);
int accflow_launch_conv_f32_2211(const accflow_conv_desc& d, hipStream_t st);
int accflow_launch_conv_f32_1431(const accflow_conv_desc& d, hipStream_t st);
int accflow_launch_conv_f32_2221(const accflow_conv_desc& d, hipStream_t st);

int accflow_launch_conv_f32(const accflow_conv_desc& d, int wc, int wp, int tc, int tp, hipStream_t st) {
  const int key = wc * 1000 + wp * 100 + tc * 10 + tp;
  switch (key) {
    case 1412: return accflow_launch_conv_f32_1412(d, st);   // 32 ch x 256 px
    case 1411: return accflow_launch_conv_f32_1411(d, st);   // 32 ch x 128 px
    case 2212: return accflow_launch_conv_f32_2212(d, st);   // 64 ch x 128 px
    case 2211: return accflow_launch_conv_f32_2211(d, st);   // 64 ch x 64 px
    case 1431: return accflow_launch_conv_f32_1431(d, st);   // 96 ch x 128 px
    case 2222: return launch_conv<2, 2, 2, 2>(d, st);        // 128 ch x 128 px (this translation unit)
    case 2221: return accflow_launch_conv_f32_2221(d, st);   // 128 ch x 64 px
  }
  return 1;
}
