// Shared device helpers for libaccflow_hip (gfx950 only: wave64, fp32-input MFMA 32x32x2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "accflow_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// 4-byte aligned 16-byte vector: lets hipcc emit global_load_dwordx4 on dword-aligned addresses
// (gfx950 global memory runs in unaligned-access mode).
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) f2u { float x, y; };

#define ACCFLOW_RETURN_LAUNCH_STATUS() return (int)hipGetLastError()

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

constexpr int MMA_BK = 16;  // granularity of the packed-weight K padding (and default slab depth)

// One BK-deep slab of D[ch x px] += W[k x ch]^T X[k x px] on v_mfma_f32_32x32x2_f32.
// Ws: [BK][LDW] (k-major, output rows contiguous), Xs: [BK][LDX].  A-operand lane map: lane l holds
// A[i = l&31][k = l>>5]; B: B[k = l>>5][j = l&31]; so both fragments are one conflict-free
// ds_read_b32 per lane (two 32-lane groups read two different k rows).
template <int TC, int TP, int LDW, int LDX, int BKS = MMA_BK>
__device__ __forceinline__ void mma_slab(const float* __restrict__ Ws, const float* __restrict__ Xs,
                                         f32x16 (&acc)[TC][TP], int wrow0, int xcol0, int lane) {
  const int l31 = lane & 31, kh = lane >> 5;
  const float* __restrict__ wp = Ws + kh * LDW + wrow0 + l31;
  const float* __restrict__ xp = Xs + kh * LDX + xcol0 + l31;
  // fragments are double-buffered in registers: the ds_reads of k-step kk+1 are in flight while the
  // MFMAs of k-step kk issue (the compiler turns the dependencies into counted lgkmcnt waits).
  float a[2][TC], b[2][TP];
#pragma unroll
  for (int tc = 0; tc < TC; ++tc) a[0][tc] = wp[tc * 32];
#pragma unroll
  for (int tp = 0; tp < TP; ++tp) b[0][tp] = xp[tp * 32];
#pragma unroll
  for (int kk = 0; kk < BKS / 2; ++kk) {
    const int cur = kk & 1, nxt = cur ^ 1;
    if (kk + 1 < BKS / 2) {
#pragma unroll
      for (int tc = 0; tc < TC; ++tc) a[nxt][tc] = wp[(kk + 1) * 2 * LDW + tc * 32];
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) b[nxt][tp] = xp[(kk + 1) * 2 * LDX + tp * 32];
    }
#pragma unroll
    for (int tc = 0; tc < TC; ++tc)
#pragma unroll
      for (int tp = 0; tp < TP; ++tp)
        acc[tc][tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][tc], b[cur][tp], acc[tc][tp], 0, 0, 0);
  }
}

// C/D map of the 32x32 accumulator: register r of lane l is row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31.
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACCFLOW_ACT_RELU: return fmaxf(v, 0.0f);
    case ACCFLOW_ACT_SIGMOID: return sigmoidf_(v);
    case ACCFLOW_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// One 16-byte chunk pair of the "S16" activation format (include/accflow_hip.h): 8 channel values of one pixel ->
// hi = fp16(x 2^ASHIFT), lo = fp16(x 2^ASHIFT - hi); `bad` collects the fp16 range check (-> accflow_conv_desc.guard).
typedef unsigned mu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void s16_split8(const float (&x)[8], mu32x4& hi, mu32x4& lo, bool& bad) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  unsigned h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = x[2 * j] * (float)(1 << ACCFLOW_F16_ASHIFT), b = x[2 * j + 1] * (float)(1 << ACCFLOW_F16_ASHIFT);
    bad |= !(fabsf(a) < 65520.0f) | !(fabsf(b) < 65520.0f);
    const f2 v = {a, b};
    const h2 hq = __builtin_convertvector(v, h2);
    const f2 back = __builtin_convertvector(hq, f2);
    const f2 r = {a - back[0], b - back[1]};
    const h2 lq = __builtin_convertvector(r, h2);
    h[j] = __builtin_bit_cast(unsigned, hq);
    l[j] = __builtin_bit_cast(unsigned, lq);
  }
  hi = mu32x4{h[0], h[1], h[2], h[3]};
  lo = mu32x4{l[0], l[1], l[2], l[3]};
}
