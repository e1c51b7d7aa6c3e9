// Is the displaced correlation volume's WRITE PATTERN what limits corr_disp_ring_kernel?  (VERDICT r03 #4 / r04 #3.)
// Level 0 of one pair at 60 x 128: E[pb 60][dy 60][dx 128][128 floats] = 236 MB.  Every pattern writes each byte exactly
// once with dword stores, 64 KB per workgroup, nothing else in the kernel:
//   A  (today's tile: 128 query pixels x 2 rows x 64 target columns): for each of 2 dy and 128 dx a 256-byte piece that
//      starts at float (xc*64 - dx) mod 128 of the 512-byte row and WRAPS - two workgroups (xc = 0, 1) complete a row;
//   B  (64 query pixels x 2 full rows): for each of 2 dy and 128 dx the ALIGNED half row [ph*64, ph*64 + 64);
//   C  (128 query pixels x 1 full row): whole 512-byte rows, 64 KB contiguous per workgroup.
// Workgroup order as in the kernel (XCD-aware: the 8 workgroups of a round-robin group belong to 8 query blocks).
// build: hipcc -O3 --offload-arch=gfx950 tools/write_pattern_probe.hip -o tools/bin/write_pattern_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int H8 = 60, W8 = 128, P = H8 * W8, NPB = P / 128;
template <int PAT>
__global__ __launch_bounds__(256) void wr(float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  constexpr int percol = (NPB + 7) / 8;
  const int pb = xcd + 8 * (seq % percol), qt = seq / percol;
  if (pb >= NPB) return;
  const int y1 = pb;                                // (W8 = 128: a query block is one image row)
  if (PAT == 0) {
    const int yo = qt >> 1, xc = qt & 1;
    for (int k = wave * 64; k < wave * 64 + 64; ++k) {
      const int h = k >> 7, dx = k & 127;
      int dy = 2 * yo + h - y1; if (dy < 0) dy += H8;
      const int p = (xc * 64 - dx + lane) & 127;
      out[(((long long)pb * H8 + dy) * W8 + dx) * 128 + p] = (float)k;
    }
  } else if (PAT == 1) {
    const int yo = qt >> 1, ph = qt & 1;
    for (int k = wave * 64; k < wave * 64 + 64; ++k) {
      const int h = k >> 7, dx = k & 127;
      int dy = 2 * yo + h - y1; if (dy < 0) dy += H8;
      out[(((long long)pb * H8 + dy) * W8 + dx) * 128 + ph * 64 + lane] = (float)k;
    }
  } else {
    const int y2 = qt;                              // 60 rows -> gridDim covers qt < 60 with 2x the 64-KB... see main
    int dy = y2 - y1; if (dy < 0) dy += H8;
    for (int k = wave * 64; k < wave * 64 + 64; ++k) {
      const int dx = k >> 1, half = k & 1;
      out[(((long long)pb * H8 + dy) * W8 + dx) * 128 + half * 64 + lane] = (float)k;
    }
  }
}
template <int PAT>
void run(const char* what, float* buf, int nqt) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const dim3 grid(8 * ((NPB + 7) / 8) * nqt);
  // (8 pair-sized buffers in rotation: 1.9 GB, far beyond the 256 MB Infinity Cache - as the 11 pairs of a sequence are)
  const size_t pair = (size_t)P * P;
  for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(wr<PAT>, grid, dim3(256), 0, 0, buf + (i % 8) * pair);
  hipDeviceSynchronize();
  hipEventRecord(a);
  const int n = 16;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(wr<PAT>, grid, dim3(256), 0, 0, buf + (i % 8) * pair);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)P * P * 4;
  printf("%-64s %7.1f us per 236 MB = %.2f TB/s\n", what, ms * 1e3 / n, bytes / (ms * 1e-3 / n) / 1e12);
}
int main() {
  float* buf; (void)hipMalloc(&buf, (size_t)P * P * 4 * 8);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("A: 256-B pieces at arbitrary offsets, wrapping (today)", buf, 60);
    run<1>("B: aligned 256-B half rows (64 queries x 2 full rows)", buf, 60);
    run<2>("C: whole 512-B rows, 64 KB contiguous per workgroup", buf, 60);
  }
  return 0;
}
