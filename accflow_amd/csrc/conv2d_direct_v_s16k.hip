// (-DACCFLOW_KPROF builds compile this body inside conv2d_direct.hip instead: the in-kernel stamp buffer is per translation unit)
#if !defined(ACCFLOW_KPROF) || defined(ACCFLOW_DIRECT_UNITY)
#include "conv2d_direct_kernel.h"
// instantiation group: S16 sources, 128-channel kernel, tap-specialised K loop for the GRU's 1x5 / 5x1 convolutions (5 taps).
// (Round 5's 9-tap instantiation needed 192 VGPRs - run-time tap offsets made hipcc keep one LDS address register per (tile, tap,
// stage) - and was dropped; with immediate offsets it fits the 5-tap kernel's 137: conv2d_direct_v_s16k9.hip.)
// Round 6: one instantiation per kernel SHAPE (1x5 and 5x1): the tap offsets inside the LDS patch are immediates of the fragment
// reads, one base register per stage (conv2d_direct_kernel.h, KWC).
int accflow_direct_launch_s16k(const accflow_conv_desc& d, int kt, dim3 grid, hipStream_t st) {
  if (kt == 9) return accflow_direct_launch_s16k9(d, 2, false, grid, st);
  if (kt != 5) return 1;
  if (d.KW == 5 && d.KH == 1)
    hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<2, 2, true, true, false, true, 7, false, 5, 5>), grid, dim3(256), 0, st, d);
  else if (d.KW == 1 && d.KH == 5)
    hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<2, 2, true, true, false, true, 7, false, 5, 1>), grid, dim3(256), 0, st, d);
  else
    return 1;
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
#endif
