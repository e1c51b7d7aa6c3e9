// (-DACCFLOW_KPROF builds compile this body inside conv2d_direct.hip instead: the in-kernel stamp buffer is per translation unit)
#if !defined(ACCFLOW_KPROF) || defined(ACCFLOW_DIRECT_UNITY)
#include "conv2d_direct_kernel.h"
// instantiation group: fp16 split with normalise-on-load
// (round 6: the 3x3 shapes - conv2 of the feature encoder's residual blocks, extractor.py:56-58 - on the tap-specialised 9-tap loop
// with the gather loader: the patch of the next chunk is gathered at tap 0 and normalised / split / stored behind tap 8)
int accflow_direct_launch_f16_norm(const accflow_conv_desc& d, int tc, dim3 grid, hipStream_t st) {
  static const bool kt9_on = [] { const char* e = getenv("ACCFLOW_DIRECT_KT9"); return !e || atoi(e) != 0; }();
  // (128-channel kernel only: 0.354 -> 0.314 and 0.161 -> 0.146 ms per step on the 96- / 128-channel layers; the 64-channel kernel
  // needs 129 registers in this form - 3 instead of 4 waves per SIMD - and was 4 % slower: profiles/r06_ab_kt9_norm.txt)
  if (kt9_on && tc == 2 && d.KH == 3 && d.KW == 3 && d.C0 % 16 == 0) {
    hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<2, 2, true, true, true, false, 7, false, 9, 3>), grid, dim3(256), 0, st, d);
    ACCFLOW_RETURN_LAUNCH_STATUS();
  }
  if (tc == 2) hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<2, 2, true, true, true>), grid, dim3(256), 0, st, d);
  else hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<1, 2, true, false, true>), grid, dim3(256), 0, st, d);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
#endif
