// (-DACCFLOW_KPROF builds compile this body inside conv2d_direct.hip instead: the in-kernel stamp buffer is per translation unit)
#if !defined(ACCFLOW_KPROF) || defined(ACCFLOW_DIRECT_UNITY)
#include "conv2d_direct_kernel.h"
// instantiation group: S16 sources (fp16 split, DMA loader)
int accflow_direct_launch_s16(const accflow_conv_desc& d, int tc, dim3 grid, hipStream_t st) {
  if (tc == 2) hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<2, 2, true, true, false, true>), grid, dim3(256), 0, st, d);
  else hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<1, 2, true, false, false, true>), grid, dim3(256), 0, st, d);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
#endif
