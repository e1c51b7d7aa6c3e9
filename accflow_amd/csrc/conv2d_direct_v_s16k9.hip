// (-DACCFLOW_KPROF builds compile this body inside conv2d_direct.hip instead: the in-kernel stamp buffer is per translation unit)
#if !defined(ACCFLOW_KPROF) || defined(ACCFLOW_DIRECT_UNITY)
#include "conv2d_direct_kernel.h"
// instantiation group (round 6): S16 sources, tap-specialised K loop for 3x3 convolutions (9 taps, kernel width 3 a compile-time
// constant: every LDS fragment address is one base register + an immediate): the 128-channel kernel (convc2's 128 channels,
// conv), its ACCFLOW_EPI_TAPGEMM form (the flow head) and the 64-channel kernel (convf2, convc2's last 64 channels).
// Same-box A/B of the 128-channel form alone: 23.27-23.31 vs 23.44-23.54 ms per step (profiles/r06_ab_kt9.txt).
int accflow_direct_launch_s16k9(const accflow_conv_desc& d, int tc, bool tapgemm, dim3 grid, hipStream_t st) {
  if (d.KW != 3 || d.KH != 3) return 1;
  if (tapgemm) {
    if (tc != 2) return 1;
    hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<2, 2, true, true, false, true, 7, true, 9, 3>), grid, dim3(256), 0, st, d);
  } else if (tc == 2) {
    hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<2, 2, true, true, false, true, 7, false, 9, 3>), grid, dim3(256), 0, st, d);
  } else {
    hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<1, 2, true, false, false, true, 7, false, 9, 3>), grid, dim3(256), 0, st, d);
  }
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
#endif
