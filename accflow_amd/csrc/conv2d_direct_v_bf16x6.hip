// (-DACCFLOW_KPROF builds compile this body inside conv2d_direct.hip instead: the in-kernel stamp buffer is per translation unit)
#if !defined(ACCFLOW_KPROF) || defined(ACCFLOW_DIRECT_UNITY)
#include "conv2d_direct_kernel.h"
// instantiation group: bf16 split, three terms (bf16x6), fp32 sources - the other term count lives in conv2d_direct_v_bf16x3.hip
int accflow_direct_launch_bf16x3(const accflow_conv_desc& d, int tc, bool w4, dim3 grid, hipStream_t st);
int accflow_direct_launch_bf16(const accflow_conv_desc& d, int tc, int nt, bool w4, dim3 grid, hipStream_t st) {
  if (nt == 2) return accflow_direct_launch_bf16x3(d, tc, w4, grid, st);
  if (tc == 2 && w4) hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<2, 3, false, true>), grid, dim3(256), 0, st, d);
  else if (tc == 2) hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<2, 3>), grid, dim3(256), 0, st, d);
  else hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<1, 3>), grid, dim3(256), 0, st, d);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
#endif
