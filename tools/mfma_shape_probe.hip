// Which 16-bit MFMA shape holds the higher clock in a loop shaped like the direct conv kernel's step?
// (MI355X_MICROARCH.md, DVFS give-back item 7: on random data the chip can hold a higher clock on 16x16x32 than on
// 32x32x16; cdna_hip_programming.md rule 28: "build both at the same output tile per wave and keep the faster by wall".)
//
// Both variants: 256-thread workgroups, 3 per CU, each wave owns a 32-channel x 128-pixel fp32 accumulator tile
// (64 registers), per 32 of K it loads 4 x 16 B per lane of "weight" fragments from an L2-resident buffer (two terms),
// reads 16 x 16 B per lane of "activation" fragments from LDS (two terms) and issues the 3-product fp16 split:
//   shape 0: 2 x 12 v_mfma_f32_32x32x16_f16      shape 1: 48 v_mfma_f32_16x16x32_f16      (same flop)
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int A_ELEMS = 1 << 16;   // 1 MiB of u32x4: stays in L2
constexpr int LDS_ELEMS = 2048;    // 32 KiB per workgroup are read ...
constexpr int LDS_ALLOC = 3328;    // ... of 52 KiB allocated: 3 workgroups per CU, like the conv kernel

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void probe(const u32x4* __restrict__ A, float* __restrict__ out, int iters,
                                                unsigned long long* __restrict__ clk) {
  __shared__ u32x4 lds[LDS_ALLOC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < LDS_ALLOC; i += 256) lds[i] = A[(i * 13 + blockIdx.x * 7) & (A_ELEMS - 1)];
  __syncthreads();
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(A), 0, A_ELEMS * 16, 0x00020000);
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  float total = 0.0f;
  if constexpr (SHAPE == 0) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    f16x8 a[2][2], an[2][2];  // [substep][term]
    unsigned voff = (unsigned)((wave * 64 + lane) * 16);
    for (int s = 0; s < 2; ++s)
      for (int t = 0; t < 2; ++t)
        a[s][t] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)voff, (s * 2 + t) * 4096, 0));
    for (int it = 0; it < iters; ++it) {
      const int so = ((it + 1) * 16384) & (A_ELEMS * 16 - 1);
      for (int s = 0; s < 2; ++s)
        for (int t = 0; t < 2; ++t)
          an[s][t] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)voff, so + (s * 2 + t) * 4096, 0));
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f16x8 b[2][4];
        const int base = ((it * 2 + s) * 97) & 1023;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int p = 0; p < 4; ++p) b[t][p] = __builtin_bit_cast(f16x8, lds[(base + t * 512 + p * 128 + lane) & (LDS_ELEMS - 1)]);
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][1], b[0][p], acc[p], 0, 0, 0);
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][0], b[1][p], acc[p], 0, 0, 0);
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][0], b[0][p], acc[p], 0, 0, 0);
      }
      for (int s = 0; s < 2; ++s)
        for (int t = 0; t < 2; ++t) a[s][t] = an[s][t];
    }
    for (int t = 0; t < 4; ++t)
      for (int r = 0; r < 16; ++r) total += acc[t][r];
  } else {
    f32x4 acc[2][8];
    for (int m = 0; m < 2; ++m)
      for (int p = 0; p < 8; ++p)
        for (int r = 0; r < 4; ++r) acc[m][p][r] = 0.0f;
    f16x8 a[2][2], an[2][2];  // [row tile][term]
    unsigned voff = (unsigned)((wave * 64 + lane) * 16);
    for (int m = 0; m < 2; ++m)
      for (int t = 0; t < 2; ++t)
        a[m][t] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)voff, (m * 2 + t) * 4096, 0));
    for (int it = 0; it < iters; ++it) {
      const int so = ((it + 1) * 16384) & (A_ELEMS * 16 - 1);
      for (int m = 0; m < 2; ++m)
        for (int t = 0; t < 2; ++t)
          an[m][t] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)voff, so + (m * 2 + t) * 4096, 0));
      const int base = (it * 2 * 97) & 1023;
#pragma unroll
      for (int h = 0; h < 2; ++h) {  // two halves of the 8 pixel tiles: 8 reads in flight like the 32x32 variant
        f16x8 b[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int p = 0; p < 4; ++p)
            b[t][p] = __builtin_bit_cast(f16x8, lds[(base + h * 61 + t * 512 + p * 128 + lane) & (LDS_ELEMS - 1)]);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
          for (int p = 0; p < 4; ++p) acc[m][h * 4 + p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m][1], b[0][p], acc[m][h * 4 + p], 0, 0, 0);
#pragma unroll
          for (int p = 0; p < 4; ++p) acc[m][h * 4 + p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m][0], b[1][p], acc[m][h * 4 + p], 0, 0, 0);
#pragma unroll
          for (int p = 0; p < 4; ++p) acc[m][h * 4 + p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m][0], b[0][p], acc[m][h * 4 + p], 0, 0, 0);
        }
      }
      for (int m = 0; m < 2; ++m)
        for (int t = 0; t < 2; ++t) a[m][t] = an[m][t];
    }
    for (int m = 0; m < 2; ++m)
      for (int p = 0; p < 8; ++p)
        for (int r = 0; r < 4; ++r) total += acc[m][p][r];
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + tid] = total;
  if (tid == 0) {
    clk[blockIdx.x * 2] = t1 - t0;
    clk[blockIdx.x * 2 + 1] = r1 - r0;
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 3000;
  const int nwg = 768 * 2;  // two rounds of 3 workgroups per CU
  std::vector<unsigned short> h(A_ELEMS * 8);
  srand(1);
  for (auto& v : h) {  // random fp16 in about [-2, 2): sign, exponent 12..15, random mantissa
    const unsigned r = rand();
    v = (unsigned short)(((r & 1) << 15) | ((12 + ((r >> 1) & 3)) << 10) | ((r >> 3) & 1023));
  }
  u32x4* dA; float* dO; unsigned long long* dC;
  hipMalloc(&dA, A_ELEMS * 16); hipMalloc(&dO, nwg * 256 * 4); hipMalloc(&dC, nwg * 16);
  hipMemcpy(dA, h.data(), A_ELEMS * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int zero = 0; zero < 2; ++zero) {
    if (zero) hipMemset(dA, 0, A_ELEMS * 16);
    for (int shape = 0; shape < 2; ++shape) {
      for (int rep = 0; rep < 3; ++rep) {  // the last repetition is reported (clocks settle under sustained load)
        hipEventRecord(e0);
        for (int k = 0; k < 6; ++k) {
          if (shape == 0) hipLaunchKernelGGL(probe<0>, dim3(nwg), dim3(256), 0, 0, dA, dO, iters, dC);
          else hipLaunchKernelGGL(probe<1>, dim3(nwg), dim3(256), 0, 0, dA, dO, iters, dC);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep < 2) continue;
        std::vector<unsigned long long> c(nwg * 2);
        hipMemcpy(c.data(), dC, nwg * 16, hipMemcpyDeviceToHost);
        double cyc = 0, rt = 0;
        for (int i = 0; i < nwg; ++i) { cyc += c[2 * i]; rt += c[2 * i + 1]; }
        const double flop = 6.0 * nwg * 4.0 * iters * 24.0 * 32768.0;  // per wave and iteration: 24 x 32x32x16 (or 48 x 16x16x32)
        printf("%s data, %s: %.3f ms per launch, %.1f TFLOP/s of MFMA, in-kernel clock %.3f GHz, %.0f cycles per 32-deep step and wave\n",
               zero ? "zero  " : "random", shape ? "16x16x32" : "32x32x16", ms / 6, flop / (ms * 1e-3) / 1e12, cyc / rt * 0.1,
               cyc / nwg / iters);
      }
    }
  }
  return 0;
}
