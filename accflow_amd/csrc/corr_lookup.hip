// CorrBlock.__call__ : radius-4 lookup in the 4-level correlation pyramid
// (raft/corr.py:24-45 + bilinear_sampler raft/utils/utils.py:66-80; gma/corr.py:25-48).
//
// out[b, l*81 + i*9 + j, y, x] = bilinear_zeros(V_l[b,y,x], cx/2^l + (i-4), cy/2^l + (j-4)):
// i walks along x and j along y (the reference adds meshgrid(dy,dx) to an (x,y) centroid, so the
// window is transposed w.r.t. row-major).  All 81 taps of a level share one fractional offset, so a
// 10x10 integer window W[r][q] = V_l[y0-4+r][x0-4+q] (zero outside the plane) is enough:
//   out(i,j) = W[j][i]*(1-fx)(1-fy) + W[j][i+1]*fx(1-fy) + W[j+1][i]*(1-fx)fy + W[j+1][i+1]*fx*fy.
//
// Work split: one 256-thread workgroup = 64 consecutive query pixels; wave l handles pyramid level l,
// lane = pixel.  Each lane streams the 10 rows of its private window (40 B each) with dword-aligned
// 16-byte loads where the whole row is inside the plane, keeps two rows in registers and emits the 9
// j-rows x 9 i-columns; every store instruction writes 64 consecutive pixels of one output channel
// (256 B, coalesced NCHW).
#include "common.h"

namespace {

constexpr int R = 4, WIN = 2 * R + 2;  // 10x10 integer window

__device__ __forceinline__ void load_row(const float* __restrict__ plane, int Hl, int Wl, int yy, int xs,
                                         float (&row)[WIN]) {
  if ((unsigned)yy >= (unsigned)Hl) {
#pragma unroll
    for (int q = 0; q < WIN; ++q) row[q] = 0.0f;
    return;
  }
  const float* p = plane + yy * Wl + xs;
  if (xs >= 0 && xs + WIN <= Wl) {
    const f4u a = *reinterpret_cast<const f4u*>(p);
    const f4u b = *reinterpret_cast<const f4u*>(p + 4);
    const f2u c = *reinterpret_cast<const f2u*>(p + 8);
    row[0] = a.x; row[1] = a.y; row[2] = a.z; row[3] = a.w;
    row[4] = b.x; row[5] = b.y; row[6] = b.z; row[7] = b.w;
    row[8] = c.x; row[9] = c.y;
  } else {
#pragma unroll
    for (int q = 0; q < WIN; ++q) row[q] = ((unsigned)(xs + q) < (unsigned)Wl) ? p[q] : 0.0f;
  }
}

__global__ __launch_bounds__(256) void corr_lookup_kernel(const float* __restrict__ l0, const float* __restrict__ l1,
                                                          const float* __restrict__ l2, const float* __restrict__ l3,
                                                          const float* __restrict__ coords, float* __restrict__ out,
                                                          long long out_bs, int B, int H8, int W8) {
  const int P = H8 * W8;
  const int lane = threadIdx.x & 63;
  const int lvl = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long gp = (long long)blockIdx.x * 64 + lane;  // global query pixel (b, y, x)
  if (gp >= (long long)B * P) return;
  const int b = (int)(gp / P);
  const int pix = (int)(gp - (long long)b * P);

  const float* vol = lvl == 0 ? l0 : lvl == 1 ? l1 : lvl == 2 ? l2 : l3;
  const int Hl = H8 >> lvl, Wl = W8 >> lvl;
  const float* plane = vol + gp * (long long)(Hl * Wl);

  const float inv = 1.0f / (float)(1 << lvl);
  float cx = coords[((long long)b * 2 + 0) * P + pix] * inv;
  float cy = coords[((long long)b * 2 + 1) * P + pix] * inv;
  cx = fminf(fmaxf(cx, -1.0e6f), 1.0e6f);
  cy = fminf(fmaxf(cy, -1.0e6f), 1.0e6f);
  const float fx0 = floorf(cx), fy0 = floorf(cy);
  const float ax = cx - fx0, ay = cy - fy0;
  const int xs = (int)fx0 - R, ys = (int)fy0 - R;
  const float w00 = (1.0f - ax) * (1.0f - ay), w01 = ax * (1.0f - ay), w10 = (1.0f - ax) * ay, w11 = ax * ay;

  float* o = out + (long long)b * out_bs + (long long)(lvl * 81) * P + pix;
  float r0[WIN], r1[WIN];
  load_row(plane, Hl, Wl, ys, xs, r0);
#pragma unroll
  for (int j = 0; j < 2 * R + 1; ++j) {
    load_row(plane, Hl, Wl, ys + j + 1, xs, r1);
#pragma unroll
    for (int i = 0; i < 2 * R + 1; ++i) {
      // (the same explicit fma chain as corr_disp.hip: the two layouts return identical bits)
      const float v = __builtin_fmaf(r1[i + 1], w11, __builtin_fmaf(r1[i], w10, __builtin_fmaf(r0[i + 1], w01, r0[i] * w00)));
      o[(long long)(i * 9 + j) * P] = v;
    }
#pragma unroll
    for (int q = 0; q < WIN; ++q) r0[q] = r1[q];
  }
}

}  // namespace

extern "C" int accflow_corr_lookup_f32(const float* lvl0, const float* lvl1, const float* lvl2, const float* lvl3,
                                       const float* coords, float* out, long long out_bs, int B, int H8, int W8,
                                       void* stream) {
  if (!lvl0 || !lvl1 || !lvl2 || !lvl3 || !coords || !out || B <= 0 || H8 < 8 || W8 < 8) return 1;
  const long long np = (long long)B * H8 * W8;
  hipLaunchKernelGGL(corr_lookup_kernel, dim3(cdiv(np, 64)), dim3(256), 0, as_stream(stream), lvl0, lvl1, lvl2,
                     lvl3, coords, out, out_bs, B, H8, W8);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
