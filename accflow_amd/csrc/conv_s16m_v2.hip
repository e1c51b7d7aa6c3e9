#include "conv_s16m_kernel.h"
// instantiation unit: wave layout 2 of the multi-source S16 kernel
int accflow_s16m_launch_2(const accflow_conv_desc& d, dim3 grid, hipStream_t st, bool) {
  hipLaunchKernelGGL((conv_s16m_kernel<2>), grid, dim3(256), 0, st, d);   // (the small-grid 64-channel layout: generic loop only)
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
#ifdef ACCFLOW_KPROF
extern "C" int accflow_debug_kprof_s16m_2(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kprof), 4096 * 16 * 8);
  if (reset) { void* p; hipGetSymbolAddress(&p, HIP_SYMBOL(g_kprof)); hipMemset(p, 0, 4096 * 16 * 8); }
  return (int)hipGetLastError();
}
#endif
