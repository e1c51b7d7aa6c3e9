// (-DACCFLOW_KPROF builds compile this body inside conv2d_direct.hip instead: the in-kernel stamp buffer is per translation unit)
#if !defined(ACCFLOW_KPROF) || defined(ACCFLOW_DIRECT_UNITY)
#include "conv2d_direct_kernel.h"
// instantiation group: bf16 split, two terms (bf16x3), fp32 sources
int accflow_direct_launch_bf16x3(const accflow_conv_desc& d, int tc, bool w4, dim3 grid, hipStream_t st) {
  if (tc == 2 && w4) hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<2, 2, false, true>), grid, dim3(256), 0, st, d);
  else if (tc == 2) hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<2, 2>), grid, dim3(256), 0, st, d);
  else hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<1, 2>), grid, dim3(256), 0, st, d);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
#endif
