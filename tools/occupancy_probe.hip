#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int MODE>
__global__ __launch_bounds__(256) void spin(unsigned long long cycles, int* sink) {
  extern __shared__ int sm[];
  sm[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (MODE == 1) { asm volatile("v_mov_b32 v139, 0\n v_accvgpr_write_b32 a63, 0" ::: "v139", "a63"); }
  if (MODE == 2) { asm volatile("v_mov_b32 v250, 0" ::: "v250"); }
  if (MODE == 3) { asm volatile("v_mov_b32 v120, 0\n v_accvgpr_write_b32 a120, 0" ::: "v120", "a120"); }
  if (MODE == 4) { asm volatile("v_mov_b32 v160, 0" ::: "v160"); }
  if (MODE == 5) { asm volatile("v_mov_b32 v100, 0\n v_accvgpr_write_b32 a63, 0" ::: "v100", "a63"); }
  const unsigned long long t0 = __builtin_readcyclecounter();
  int acc = 0;
  while (__builtin_readcyclecounter() - t0 < cycles) acc += sm[(threadIdx.x + acc) & 255];
  if (acc == 123456789) sink[0] = acc;
}
template <int MODE>
void run(const char* what, int* sink) {
  hipFuncSetAttribute((const void*)spin<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int kb : {8, 72}) for (int mult : {1, 2, 3, 4}) {
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spin<MODE>, 256, (size_t)kb * 1024);
    hipLaunchKernelGGL(spin<MODE>, dim3(256 * mult), dim3(256), kb * 1024, 0, 200000ull, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(spin<MODE>, dim3(256 * mult), dim3(256), kb * 1024, 0, 200000ull, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s lds %3d KB  blocks/CU launched %d  api-occupancy %d  time %.1f us\n", what, kb, mult, nb, ms * 1e3);
  }
}
int main() {
  int* sink; (void)hipMalloc(&sink, 4);
  run<0>("few regs", sink);
  run<1>("v139+a63 (total 204)", sink);
  run<5>("v100+a63 (total ~168)", sink);
  run<4>("v160", sink);
  run<2>("v250", sink);
  run<3>("v120+a120", sink);
  return 0;
}
