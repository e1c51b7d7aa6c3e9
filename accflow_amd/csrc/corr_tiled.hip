// Tiled correlation pyramid: the hot-path layout of the all-pairs volume (the row-major variant in
// corr_volume.hip / corr_lookup.hip keeps the reference's CorrBlock.corr_pyramid layout for API users).
//
// Why: every query pixel owns a private (Hl x Wl) plane and a lookup reads a 10x10 window of it per level.
// Row-major, that is 10 row segments of 40 B, each dragging in one or two 128-B lines: PMC showed ~500 MB of
// HBM traffic per launch against 245 MB algorithmic.  Here each plane is cut into tiles of 4 rows x 8 columns
// (128 B = one line), stored as two 64-B sectors of 4 rows x 4 columns; a 10x10 window then touches ~3.25 x
// 3.25 sectors (~680 B) instead of ~13 lines (~1.7 KB), and every 16-B chunk (one row of a sector) is a
// naturally aligned dwordx4.
//
//   tiled(y, x) = ((y/4) * TX + x/8) * 32 + ((x%8)/4) * 16 + (y%4) * 4 + x%4,   TX = ceil(W/8), planes padded
//   to multiples of 4 rows / 8 columns (padding holds zeros and is never selected: the lookup masks by the true
//   plane size, exactly reproducing grid_sample's zero padding).
//
// Level 0 comes straight out of the GEMM: fmap2's pixel axis is permuted into tiled order first, so the GEMM's
// row i IS query pixel i's tiled plane.  Levels 1..3 are pooled per plane through LDS in the reference's
// pool-of-pool order.
#include "common.h"

int accflow_gemm_atb_f32(const float* A, const float* Bm, float* C, int M, int N, int K, long long a_bs,
                         long long b_bs, long long c_bs, int batch, float scale, hipStream_t st);

namespace {

__host__ __device__ __forceinline__ int tiled_index(int y, int x, int TX) {
  return ((y >> 2) * TX + (x >> 3)) * 32 + ((x >> 2) & 1) * 16 + (y & 3) * 4 + (x & 3);
}
__host__ __device__ __forceinline__ int pad4(int h) { return (h + 3) & ~3; }
__host__ __device__ __forceinline__ int pad8(int w) { return (w + 7) & ~7; }

// f2t[b][c][t] = f2[b][c][y*W + x] for t = tiled(y, x); zeros in the padding
__global__ void tile_permute_kernel(const float* __restrict__ f2, float* __restrict__ f2t, int BC, int H, int W) {
  const int Hp = pad4(H), Wp = pad8(W), TX = Wp >> 3, Pt = Hp * Wp;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (long long)BC * Pt) return;
  const int bc = (int)(g / Pt), t = (int)(g - (long long)bc * Pt);
  const int tile = t >> 5, in = t & 31;
  const int y = (tile / TX) * 4 + ((in >> 2) & 3), x = (tile % TX) * 8 + (in >> 4) * 4 + (in & 3);
  f2t[g] = (y < H && x < W) ? f2[(long long)bc * H * W + y * W + x] : 0.0f;
}

// one workgroup per query plane: tiled level 0 -> tiled levels 1..3 (F.avg_pool2d(2,2): floor sizes,
// ((a+b)+c)+d then * 0.25, pool of pool)
__global__ __launch_bounds__(256) void corr_pool_tiled_kernel(const float* __restrict__ l0, float* __restrict__ l1,
                                                              float* __restrict__ l2, float* __restrict__ l3, int H0,
                                                              int W0) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int H1 = H0 >> 1, W1 = W0 >> 1, H2 = H1 >> 1, W2 = W1 >> 1, H3 = H2 >> 1, W3 = W2 >> 1;
  const int P0 = pad4(H0) * pad8(W0), P1 = pad4(H1) * pad8(W1), P2 = pad4(H2) * pad8(W2), P3 = pad4(H3) * pad8(W3);
  const int TX0 = pad8(W0) >> 3, TX1 = pad8(W1) >> 3, TX2 = pad8(W2) >> 3, TX3 = pad8(W3) >> 3;
  float* s0 = sm;
  float* s1 = s0 + P0;
  float* s2 = s1 + P1;
  const long long plane = blockIdx.x;
  const float4* src = reinterpret_cast<const float4*>(l0 + plane * P0);
  for (int i = threadIdx.x; i < P0 / 4; i += blockDim.x) reinterpret_cast<float4*>(s0)[i] = src[i];
  __syncthreads();
  auto pool = [&](const float* in, int TXi, float* outs, float* outg, int Ho, int Wo, int Po, int TXo) {
    for (int t = threadIdx.x; t < Po; t += blockDim.x) {
      const int tile = t >> 5, r = t & 31;
      const int y = (tile / TXo) * 4 + ((r >> 2) & 3), x = (tile % TXo) * 8 + (r >> 4) * 4 + (r & 3);
      float v = 0.0f;
      if (y < Ho && x < Wo) {
        const float a = in[tiled_index(2 * y, 2 * x, TXi)], b = in[tiled_index(2 * y, 2 * x + 1, TXi)];
        const float c = in[tiled_index(2 * y + 1, 2 * x, TXi)], d = in[tiled_index(2 * y + 1, 2 * x + 1, TXi)];
        v = (((a + b) + c) + d) * 0.25f;
      }
      if (outs) outs[t] = v;
      outg[t] = v;
    }
  };
  pool(s0, TX0, s1, l1 + plane * P1, H1, W1, P1, TX1);
  __syncthreads();
  pool(s1, TX1, s2, l2 + plane * P2, H2, W2, P2, TX2);
  __syncthreads();
  pool(s2, TX2, nullptr, l3 + plane * P3, H3, W3, P3, TX3);
}

constexpr int R = 4, WIN = 2 * R + 2;

// r = mask[lane] ? b : a
__device__ __forceinline__ float lane_select(float a, float b, unsigned long long mask) {
  float r;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
  return r;
}

// One window row from the tiled plane: the 16 floats of the 4 aligned column groups starting at group g0,
// then a per-lane funnel shift by sh = (xs mod 4) picks the 10 wanted columns.  Elements outside the true plane
// (rows / columns of the zero padding of grid_sample) are zeroed.
__device__ __forceinline__ void load_row_tiled(const float* __restrict__ plane, int Hl, int Wl, int TX, int ngrp,
                                               int yy, int xs, int g0, int sh, float (&row)[WIN]) {
  float d[16];
  const bool yok = (unsigned)yy < (unsigned)Hl;
  const int ybase = (yy >> 2) * TX * 32 + (yy & 3) * 4;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int gg = g0 + g;
    const bool ok = yok && (unsigned)gg < (unsigned)ngrp;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) v = *reinterpret_cast<const f32x4*>(plane + ybase + (gg >> 1) * 32 + (gg & 1) * 16);
    d[4 * g + 0] = v[0]; d[4 * g + 1] = v[1]; d[4 * g + 2] = v[2]; d[4 * g + 3] = v[3];
  }
  // per-lane funnel shift by sh dwords as two v_cndmask stages.  Written with the instruction itself: given
  // `cond ? d[q+2] : d[q]` hipcc turns d[] into a dynamically indexed stack array (scratch traffic).
  float t[12];
  const unsigned long long m2 = __ballot((sh & 2) != 0), m1 = __ballot((sh & 1) != 0);
#pragma unroll
  for (int q = 0; q < 12; ++q) t[q] = lane_select(d[q], d[q + 2], m2);
#pragma unroll
  for (int q = 0; q < WIN; ++q) {
    const float v = lane_select(t[q], t[q + 1], m1);
    row[q] = ((unsigned)(xs + q) < (unsigned)Wl) ? v : 0.0f;  // right/left edge inside a partially valid group
  }
}

__global__ __launch_bounds__(256) void corr_lookup_tiled_kernel(const float* __restrict__ l0, const float* __restrict__ l1,
                                                                const float* __restrict__ l2, const float* __restrict__ l3,
                                                                const float* __restrict__ coords, float* __restrict__ out,
                                                                long long out_bs, int B, int H8, int W8) {
  const int P = H8 * W8;
  const int lane = threadIdx.x & 63;
  const int lvl = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long gp = (long long)blockIdx.x * 64 + lane;
  if (gp >= (long long)B * P) return;
  const int b = (int)(gp / P);
  const int pix = (int)(gp - (long long)b * P);
  const float* vol = lvl == 0 ? l0 : lvl == 1 ? l1 : lvl == 2 ? l2 : l3;
  const int Hl = H8 >> lvl, Wl = W8 >> lvl;
  const int TX = pad8(Wl) >> 3, ngrp = TX * 2;
  const float* plane = vol + gp * (long long)(pad4(Hl) * pad8(Wl));

  const float inv = 1.0f / (float)(1 << lvl);
  float cx = coords[((long long)b * 2 + 0) * P + pix] * inv;
  float cy = coords[((long long)b * 2 + 1) * P + pix] * inv;
  cx = fminf(fmaxf(cx, -1.0e6f), 1.0e6f);
  cy = fminf(fmaxf(cy, -1.0e6f), 1.0e6f);
  const float fx0 = floorf(cx), fy0 = floorf(cy);
  const float ax = cx - fx0, ay = cy - fy0;
  const int xs = (int)fx0 - R, ys = (int)fy0 - R;
  const int g0 = xs >> 2, sh = xs & 3;  // arithmetic shift: floor division also for negative xs
  const float w00 = (1.0f - ax) * (1.0f - ay), w01 = ax * (1.0f - ay), w10 = (1.0f - ax) * ay, w11 = ax * ay;

  float* o = out + (long long)b * out_bs + (long long)(lvl * 81) * P + pix;
  float r0[WIN], r1[WIN];
  load_row_tiled(plane, Hl, Wl, TX, ngrp, ys, xs, g0, sh, r0);
#pragma unroll
  for (int j = 0; j < 2 * R + 1; ++j) {
    load_row_tiled(plane, Hl, Wl, TX, ngrp, ys + j + 1, xs, g0, sh, r1);
#pragma unroll
    for (int i = 0; i < 2 * R + 1; ++i)
      o[(long long)(i * 9 + j) * P] = r0[i] * w00 + r0[i + 1] * w01 + r1[i] * w10 + r1[i + 1] * w11;
#pragma unroll
    for (int q = 0; q < WIN; ++q) r0[q] = r1[q];
  }
}

}  // namespace

extern "C" long long accflow_corr_tiled_plane_elems(int Hl, int Wl) { return (long long)pad4(Hl) * pad8(Wl); }

extern "C" int accflow_corr_volume_tiled_f32(const float* fmap1, const float* fmap2, float* f2t_ws, float* lvl0,
                                             float* lvl1, float* lvl2, float* lvl3, int B, int C, int H8, int W8,
                                             void* stream) {
  if (!fmap1 || !fmap2 || !f2t_ws || !lvl0 || !lvl1 || !lvl2 || !lvl3 || B <= 0 || C <= 0 || H8 < 8 || W8 < 8) return 1;
  hipStream_t st = as_stream(stream);
  const int P = H8 * W8, Pt = pad4(H8) * pad8(W8);
  const long long n = (long long)B * C * Pt;
  hipLaunchKernelGGL(tile_permute_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, fmap2, f2t_ws, B * C, H8, W8);
  const float scale = 1.0f / sqrtf((float)C);
  int rc = accflow_gemm_atb_f32(fmap1, f2t_ws, lvl0, P, Pt, C, (long long)C * P, (long long)C * Pt, (long long)P * Pt, B,
                                scale, st);
  if (rc) return rc;
  const int H1 = H8 >> 1, W1 = W8 >> 1, H2 = H1 >> 1, W2 = W1 >> 1;
  const size_t smem = (size_t)(Pt + pad4(H1) * pad8(W1) + pad4(H2) * pad8(W2)) * sizeof(float);
  if (smem > 64 * 1024) return 1;
  hipLaunchKernelGGL(corr_pool_tiled_kernel, dim3((unsigned)((long long)B * P)), dim3(256), smem, st, lvl0, lvl1, lvl2,
                     lvl3, H8, W8);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_corr_lookup_tiled_f32(const float* lvl0, const float* lvl1, const float* lvl2,
                                             const float* lvl3, const float* coords, float* out, long long out_bs,
                                             int B, int H8, int W8, void* stream) {
  if (!lvl0 || !lvl1 || !lvl2 || !lvl3 || !coords || !out || B <= 0 || H8 < 8 || W8 < 8) return 1;
  const long long np = (long long)B * H8 * W8;
  hipLaunchKernelGGL(corr_lookup_tiled_kernel, dim3(cdiv(np, 64)), dim3(256), 0, as_stream(stream), lvl0, lvl1, lvl2,
                     lvl3, coords, out, out_bs, B, H8, W8);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
