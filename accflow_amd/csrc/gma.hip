// GMA attention (gma/modules.py:54-76, content-only branch, heads = 1) and aggregation
// (gma/modules.py:102-115).  The P x P attention matrix is materialised once per image1 (fp32, as the
// reference does) and re-read every GRU iteration by the aggregation GEMM, which is the HBM-bound part.
#include "common.h"

int accflow_gemm_atb_f32(const float* A, const float* Bm, float* C, int M, int N, int K, long long a_bs,
                         long long b_bs, long long c_bs, int batch, float scale, hipStream_t st);

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum2(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// in-place softmax over each row of a (rows, n) matrix; one workgroup per row.
__global__ __launch_bounds__(256) void row_softmax_kernel(float* __restrict__ a, int n) {
  __shared__ float red[4];
  float* row = a + (long long)blockIdx.x * n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, row[i]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.0f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float e = expf(row[i] - mx);
    row[i] = e;
    s += e;
  }
  s = wave_sum2(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  s = (red[0] + red[1]) + (red[2] + red[3]);
  for (int i = threadIdx.x; i < n; i += 256) row[i] = row[i] / s;
}

// out[b][d][i] = fmap[b][d][i] + gamma * sum_j V[b][d][j] * attn[b][i][j]   (both operands k-contiguous)
// Tile: all 128 rows d  x  128 columns i; operands are transposed into the k-major LDS layout of
// mma_slab while staging (global float4 along k, four conflict-free ds_write_b32).
__global__ __launch_bounds__(256) void gma_aggregate_kernel(const float* __restrict__ attn, const float* __restrict__ v,
                                                            const float* __restrict__ fmap,
                                                            const float* __restrict__ gamma, float* __restrict__ out,
                                                            long long out_bs, int D, int P) {
  constexpr int BC = 128, BP = 128, BK = MMA_BK;
  __shared__ __attribute__((aligned(16))) float Vs[2][BK * BC];
  __shared__ __attribute__((aligned(16))) float As[2][BK * BP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave >> 1, wp = wave & 1;
  const int b = blockIdx.z, i0 = blockIdx.x * BP, d0 = blockIdx.y * BC;
  const float* vb = v + (long long)b * D * P;
  const float* ab = attn + (long long)b * P * P;
  const int row = tid & 127, kq = tid >> 7;  // each thread: one row, 8 consecutive k
  const bool vec = (P & 3) == 0;
  float vr[8], ar[8];
  auto load = [&](int kbase) {
    const int k0 = kbase + kq * 8;
    const int d = d0 + row, i = i0 + row;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = k0 + h * 4;
      if (vec && k + 3 < P) {
        const float4 t = d < D ? *reinterpret_cast<const float4*>(vb + (long long)d * P + k) : make_float4(0, 0, 0, 0);
        const float4 u = i < P ? *reinterpret_cast<const float4*>(ab + (long long)i * P + k) : make_float4(0, 0, 0, 0);
        vr[h * 4 + 0] = t.x; vr[h * 4 + 1] = t.y; vr[h * 4 + 2] = t.z; vr[h * 4 + 3] = t.w;
        ar[h * 4 + 0] = u.x; ar[h * 4 + 1] = u.y; ar[h * 4 + 2] = u.z; ar[h * 4 + 3] = u.w;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          vr[h * 4 + q] = (d < D && k + q < P) ? vb[(long long)d * P + k + q] : 0.0f;
          ar[h * 4 + q] = (i < P && k + q < P) ? ab[(long long)i * P + k + q] : 0.0f;
        }
      }
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      Vs[buf][(kq * 8 + q) * BC + row] = vr[q];
      As[buf][(kq * 8 + q) * BP + row] = ar[q];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int tc = 0; tc < 2; ++tc)
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tc][tp][r] = 0.0f;
  const int nslab = (P + BK - 1) / BK;
  load(0);
  store(0);
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const int cur = s & 1;
    if (s + 1 < nslab) load((s + 1) * BK);
    mma_slab<2, 2, BC, BP>(Vs[cur], As[cur], acc, wc * 64, wp * 64, lane);
    if (s + 1 < nslab) store(cur ^ 1);
    __syncthreads();
  }
  const float g = gamma[0];
  const int l31 = lane & 31;
#pragma unroll
  for (int tp = 0; tp < 2; ++tp) {
    const int i = i0 + wp * 64 + tp * 32 + l31;
    if (i >= P) continue;
#pragma unroll
    for (int tc = 0; tc < 2; ++tc)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = d0 + wc * 64 + tc * 32 + acc_row(r, lane);
        if (d < D) out[b * out_bs + (long long)d * P + i] = fmap[((long long)b * D + d) * P + i] + g * acc[tc][tp][r];
      }
  }
}

// In-place softmax over the ROWS index j of a (P x P) matrix stored j-major.  (One thread per column walking all P rows
// left 342 workgroups on the chip at 720x1280: 10.6 ms per call.)
// Column softmax with the rows dealt to 16 waves per 64 columns (wave w takes rows w, w + 16, ...), TWO sweeps: an online
// (max, sum of exponentials) pass - the 16 partial pairs of a column meet in LDS and are combined in a fixed order -
// and one read-normalise-write pass.  2 reads + 1 write of the matrix instead of the 3 + 2 of max / exp / scale sweeps.
__global__ __launch_bounds__(1024) void col_softmax16_kernel(float* __restrict__ a, int P) {
  __shared__ float redm[16][64], reds[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const bool live = i < P;
  float* col = a + (long long)blockIdx.y * P * P + (live ? i : 0);
  constexpr int U = 8;
  float mx = -INFINITY, s = 0.0f;
  for (int j = wave; j < P; j += 16 * U) {
    float x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = (live && j + 16 * u < P) ? col[(long long)(j + 16 * u) * P] : -INFINITY;
    float m2 = mx;
#pragma unroll
    for (int u = 0; u < U; ++u) m2 = fmaxf(m2, x[u]);
    if (m2 > -INFINITY) {   // (a dead lane or an all -inf column keeps s = 0)
      float t = 0.0f;
#pragma unroll
      for (int u = 0; u < U; ++u) t += expf(x[u] - m2);
      s = s * expf(mx - m2) + t;
      mx = m2;
    }
  }
  redm[wave][lane] = mx;
  reds[wave][lane] = s;
  __syncthreads();
  float m = -INFINITY;
#pragma unroll
  for (int w = 0; w < 16; ++w) m = fmaxf(m, redm[w][lane]);
  float tot = 0.0f;
#pragma unroll
  for (int w = 0; w < 16; ++w) tot += redm[w][lane] > -INFINITY ? reds[w][lane] * expf(redm[w][lane] - m) : 0.0f;
  const float inv = 1.0f / tot;
  for (int j = wave; j < P; j += 16 * U) {
    float x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = (live && j + 16 * u < P) ? col[(long long)(j + 16 * u) * P] : -INFINITY;
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (live && j + 16 * u < P) col[(long long)(j + 16 * u) * P] = expf(x[u] - m) * inv;
  }
}

}  // namespace

int accflow_gma_aggregate_conv(const float* attnT, const float* v, const float* fmap, const float* gamma, float* out,
                               long long out_bs, void* ws, int mode, int* guard, int B, int D, int H, int W, hipStream_t st);

// Transposed-attention pair used by the estimator's hot path (same numbers, j-major storage):
//   attnT[b][j][i] = softmax_j(scale * <q_i, k_j>)
int accflow_corr_level0_bf16s(const float* fmap1, const float* fmap2, float* lvl0, void* ws, int B, int C, int H8,
                              int W8, int mode, int disp, int* guard, hipStream_t st, float* lvl1, int* lvl1_done);

extern "C" long long accflow_gma_attention_ws_bytes(int D, int H, int W) { return accflow_corr_volume_ws_bytes(D, H, W); }

extern "C" int accflow_gma_attention_t_f32(const float* qk, float* attnT, void* ws, int mode, int B, int D, int H, int W,
                                           float scale, void* stream) {
  if (!qk || !attnT || B <= 0 || D <= 0 || H <= 0 || W <= 0) return 1;
  const int P = H * W;
  hipStream_t st = as_stream(stream);
  static const bool split_ok = [] { const char* e = getenv("ACCFLOW_GMA_QK_SPLIT"); return !e || atoi(e) != 0; }();
  // C[j][i] = scale * sum_d k[d][j] * q[d][i]: with scale = D^-1/2 (GMA's, modules.py:42) this is the correlation
  // volume of (k, q), so the split modes reuse its matrix-core GEMM (bf16x6: fp32-equivalent) pair by pair
  if (split_ok && mode != ACCFLOW_CONV_F32 && ws && fabsf(scale * sqrtf((float)D) - 1.0f) < 1e-6f) {
    for (int b = 0; b < B; ++b) {
      const float* q = qk + (long long)b * 2 * D * P;
      const int rc = accflow_corr_level0_bf16s(q + (long long)D * P, q, attnT + (long long)b * P * P, ws, 1, D, H, W,
                                               ACCFLOW_CONV_BF16X6, 0, nullptr, st, nullptr, nullptr);
      if (rc) return rc;
    }
  } else {
    const int rc = accflow_gemm_atb_f32(qk + (long long)D * P, qk, attnT, P, P, D, 2LL * D * P, 2LL * D * P,
                                        (long long)P * P, B, scale, st);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(col_softmax16_kernel, dim3(cdiv(P, 64), B), dim3(1024), 0, st, attnT, P);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" long long accflow_gma_aggregate_ws_bytes(int B, int D, int P) {
  const long long Kpad = (P + 31) / 32 * 32, CoutPad = (D + 127) / 128 * 128;
  // im2col route: split v and row scales per item, one k-table, and (single-item calls) up to 8 split-K partial outputs
  const long long a = B * (3 * Kpad * CoutPad * 2 + CoutPad * 4) + Kpad * 16 + 16 + (B == 1 ? 8LL * D * P * 4 : 0);
  // LDS-patch route (B == 1, f16x3): v in the patch layout, row scales, split-K partials
  const long long b = B == 1 ? accflow_conv_patch_elems(D, P, 1, 1) * 2 + CoutPad * 4 + 8LL * D * P * 4 : 0;
  return a > b ? a : b;
}

extern "C" int accflow_gma_aggregate_t_f32(const float* attnT, const float* v, const float* fmap, const float* gamma,
                                           float* out, long long out_bs, void* ws, int mode, int* guard, int B, int D,
                                           int H, int W, void* stream) {
  if (!attnT || !v || !fmap || !gamma || !out || !ws || B <= 0 || D <= 0 || H <= 0 || W <= 0) return 1;
  if (mode == ACCFLOW_CONV_F32) return 1;  // the fp32 path is accflow_gma_aggregate_f32 on the i-major attention
  return accflow_gma_aggregate_conv(attnT, v, fmap, gamma, out, out_bs, ws, mode, guard, B, D, H, W, as_stream(stream));
}

int accflow_gma_aggregate_s16_impl(const void* attn16, const float* v, const float* fmap, long long fmap_bs, const float* gamma,
                                   float* out, long long out_bs, void* out16, long long out16_bs, void* ws, int* guard, int n,
                                   int D, int H, int W, hipStream_t st);

extern "C" long long accflow_gma_aggregate_s16_ws_bytes(int n, int D, int P) {
  const long long CoutPad = ((long long)n * D + 127) / 128 * 128;
  return accflow_conv_patch_elems(n * D, P, 1, 1) * 2 + CoutPad * 4 + 8LL * n * D * P * 4;
}

extern "C" int accflow_gma_aggregate_s16(const void* attn16, const float* v, const float* fmap, long long fmap_bs,
                                         const float* gamma, float* out, long long out_bs, void* out16, long long out16_bs,
                                         void* ws, int* guard, int n, int D, int H, int W, void* stream) {
  if (!attn16 || !v || !fmap || !gamma || (!out && !out16) || !ws || n <= 0 || D <= 0 || (D & 31) || H <= 0 || W <= 0) return 1;
  return accflow_gma_aggregate_s16_impl(attn16, v, fmap, fmap_bs, gamma, out, out_bs, out16, out16_bs, ws, guard, n, D, H, W,
                                        as_stream(stream));
}

extern "C" int accflow_gma_attention_f32(const float* qk, float* attn, int B, int D, int P, float scale,
                                         void* stream) {
  if (!qk || !attn || B <= 0 || D <= 0 || P <= 0) return 1;
  hipStream_t st = as_stream(stream);
  // sim[b][i][j] = scale * <q[b][:, i], k[b][:, j]>, q = qk[:, :D], k = qk[:, D:]
  int rc = accflow_gemm_atb_f32(qk, qk + (long long)D * P, attn, P, P, D, 2LL * D * P, 2LL * D * P, (long long)P * P, B,
                                scale, st);
  if (rc) return rc;
  hipLaunchKernelGGL(row_softmax_kernel, dim3((unsigned)((long long)B * P)), dim3(256), 0, st, attn, P);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_gma_aggregate_f32(const float* attn, const float* v, const float* fmap, const float* gamma,
                                         float* out, long long out_bs, int B, int D, int P, void* stream) {
  if (!attn || !v || !fmap || !gamma || !out || B <= 0 || D <= 0 || P <= 0) return 1;
  dim3 grid(cdiv(P, 128), cdiv(D, 128), B);
  hipLaunchKernelGGL(gma_aggregate_kernel, grid, dim3(256), 0, as_stream(stream), attn, v, fmap, gamma, out, out_bs, D,
                     P);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
