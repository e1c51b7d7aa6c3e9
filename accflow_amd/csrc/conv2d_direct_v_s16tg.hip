// (-DACCFLOW_KPROF builds compile this body inside conv2d_direct.hip instead: the in-kernel stamp buffer is per translation unit)
#if !defined(ACCFLOW_KPROF) || defined(ACCFLOW_DIRECT_UNITY)
#include "conv2d_direct_kernel.h"
// instantiation group: S16 sources, ACCFLOW_EPI_TAPGEMM epilogue (the flow head's two convolutions in one launch)
int accflow_direct_launch_s16tg(const accflow_conv_desc& d, dim3 grid, hipStream_t st) {
  hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<2, 2, true, true, false, true, 7, true>), grid, dim3(256), 0, st, d);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
#endif
