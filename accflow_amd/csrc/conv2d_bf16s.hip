#include "conv2d_bf16s_kernel.h"

int accflow_launch_conv_bf16s(const accflow_conv_desc& d, int tc, int tp, hipStream_t st) {
  switch (tc * 10 + tp) {
    case 11: return accflow_launch_conv_bf16s_11(d, st);   //  64 ch x  64 px
    case 12: return accflow_launch_conv_bf16s_12(d, st);   //  64 ch x 128 px
    case 21: return accflow_launch_conv_bf16s_21(d, st);   // 128 ch x  64 px
    case 22: return launch_conv_bf16s<2, 2>(d, st);        // 128 ch x 128 px (this translation unit)
  }
  return 1;
}

// level 0 of the displaced correlation pyramid: 128 x 128 tiles over (query pixel, target pixel) of one pair
int accflow_launch_corr_disp_bf16s(const accflow_conv_desc& d, hipStream_t st) {
  const int P = d.OH * d.OW;
  dim3 grid(cdiv(P, 128), cdiv(P, 128));
  if (d.mode == ACCFLOW_CONV_BF16X6) hipLaunchKernelGGL((conv2d_bf16s_kernel<2, 2, 3, 16, true>), grid, dim3(256), 0, st, d);
  else hipLaunchKernelGGL((conv2d_bf16s_kernel<2, 2, 2, 32, true>), grid, dim3(256), 0, st, d);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
