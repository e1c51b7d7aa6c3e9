// (-DACCFLOW_KPROF builds compile this body inside conv2d_direct.hip instead: the in-kernel stamp buffer is per translation unit)
#if !defined(ACCFLOW_KPROF) || defined(ACCFLOW_DIRECT_UNITY)
#include "conv2d_direct_kernel.h"
// instantiation group: S16 sources, 128-channel kernel, tap-specialised K loop for the GRU's 1x5 / 5x1 convolutions (5 taps).
// A 9-tap instantiation (3x3) was built and measured too: 192 VGPRs (2 waves per SIMD) or 168 + 12 spilled registers whose reloads
// wait for every outstanding load - 24.4-24.9 ms per step against 23.9-24.3 with the generic loop on the same boxes; not kept.
int accflow_direct_launch_s16k(const accflow_conv_desc& d, int kt, dim3 grid, hipStream_t st) {
  if (kt != 5) return 1;
  hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<2, 2, true, true, false, true, 7, false, 5>), grid, dim3(256), 0, st, d);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
#endif
