// (-DACCFLOW_KPROF builds compile this body inside conv2d_direct.hip instead: the in-kernel stamp buffer is per translation unit)
#if !defined(ACCFLOW_KPROF) || defined(ACCFLOW_DIRECT_UNITY)
#include "conv2d_direct_kernel.h"
// instantiation group: bf16 split with normalise-on-load
int accflow_direct_launch_bf16_norm(const accflow_conv_desc& d, int tc, int nt, dim3 grid, hipStream_t st) {
  if (tc == 2 && nt == 2) hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<2, 2, false, true, true>), grid, dim3(256), 0, st, d);
  else if (tc == 2) hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<2, 3, false, true, true>), grid, dim3(256), 0, st, d);
  else if (nt == 2) hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<1, 2, false, false, true>), grid, dim3(256), 0, st, d);
  else hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<1, 3, false, false, true>), grid, dim3(256), 0, st, d);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
#endif
