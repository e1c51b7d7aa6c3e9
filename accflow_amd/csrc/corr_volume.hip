// All-pairs correlation volume + 4-level pyramid (raft/corr.py:8-22, 47-55; gma/corr.py:50-58).
//
// Level 0 is a batched A^T B GEMM on the fp32-input MFMA: corr[b][i][j] = sum_c f1[b][c][i] *
// f2[b][c][j] / sqrt(C).  Both feature maps are already k-major ([c][pixel]) in NCHW, which is exactly
// the LDS layout the 32x32x2 fragments want, so the global->LDS staging is plain coalesced float4 rows
// and the store is coalesced along j.  Levels 1..3 are produced by one streaming kernel per query
// plane that pools hierarchically through LDS with the reference's pool-of-pool rounding order.
#include "common.h"

namespace {

// global (row-major [k][ld]) -> registers: NV float4 per thread of a [BK][TILEW] tile, zero filled
// outside (K, ld).
template <int NV, int TILEW>
__device__ __forceinline__ void g2r(const float* __restrict__ src, int ld, int K, int col0, int kbase, bool vec,
                                    int tid, float4 (&reg)[NV]) {
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int v = tid + j * 256;
    const int krow = v / (TILEW / 4), c4 = (v % (TILEW / 4)) * 4;
    const int k = kbase + krow, col = col0 + c4;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < K) {
      const float* p = src + (long long)k * ld + col;
      if (vec && col + 3 < ld) {
        r = *reinterpret_cast<const float4*>(p);
      } else {
        if (col < ld) r.x = p[0];
        if (col + 1 < ld) r.y = p[1];
        if (col + 2 < ld) r.z = p[2];
        if (col + 3 < ld) r.w = p[3];
      }
    }
    reg[j] = r;
  }
}
template <int NV, int TILEW>
__device__ __forceinline__ void r2s(float* dst, int tid, const float4 (&reg)[NV]) {
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int v = tid + j * 256;
    const int krow = v / (TILEW / 4), c4 = (v % (TILEW / 4)) * 4;
    *reinterpret_cast<float4*>(&dst[krow * TILEW + c4]) = reg[j];
  }
}

// C[b][i][j] = scale * sum_k A[b][k][i] * B[b][k][j];  A: (K, M) row-major, B: (K, N) row-major.
template <int TC, int TP>
__global__ __launch_bounds__(256) void gemm_atb_f32_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                           float* __restrict__ C, int M, int N, int K,
                                                           long long a_bs, long long b_bs, long long c_bs,
                                                           float scale) {
  constexpr int BC = 2 * TC * 32, BP = 2 * TP * 32, BK = MMA_BK;
  __shared__ __attribute__((aligned(16))) float As[2][BK * BC];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * BP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave >> 1, wp = wave & 1;
  const int i0 = blockIdx.y * BC, j0 = blockIdx.x * BP;
  A += (long long)blockIdx.z * a_bs;
  Bm += (long long)blockIdx.z * b_bs;
  C += (long long)blockIdx.z * c_bs;
  const bool vecA = (M & 3) == 0, vecB = (N & 3) == 0;
  constexpr int AV = BK * BC / 4 / 256, BV = BK * BP / 4 / 256;
  float4 ar[AV], br[BV];

  f32x16 acc[TC][TP];
#pragma unroll
  for (int tc = 0; tc < TC; ++tc)
#pragma unroll
    for (int tp = 0; tp < TP; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tc][tp][r] = 0.0f;

  const int nslab = (K + BK - 1) / BK;
  g2r<AV, BC>(A, M, K, i0, 0, vecA, tid, ar);
  g2r<BV, BP>(Bm, N, K, j0, 0, vecB, tid, br);
  r2s<AV, BC>(As[0], tid, ar);
  r2s<BV, BP>(Bs[0], tid, br);
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const int cur = s & 1;
    if (s + 1 < nslab) {
      g2r<AV, BC>(A, M, K, i0, (s + 1) * BK, vecA, tid, ar);
      g2r<BV, BP>(Bm, N, K, j0, (s + 1) * BK, vecB, tid, br);
    }
    mma_slab<TC, TP, BC, BP>(As[cur], Bs[cur], acc, wc * TC * 32, wp * TP * 32, lane);
    if (s + 1 < nslab) {
      r2s<AV, BC>(As[cur ^ 1], tid, ar);
      r2s<BV, BP>(Bs[cur ^ 1], tid, br);
    }
    __syncthreads();
  }

  const int l31 = lane & 31;
#pragma unroll
  for (int tp = 0; tp < TP; ++tp) {
    const int j = j0 + wp * TP * 32 + tp * 32 + l31;
    if (j >= N) continue;
#pragma unroll
    for (int tc = 0; tc < TC; ++tc)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + wc * TC * 32 + tc * 32 + acc_row(r, lane);
        if (i < M) C[(long long)i * N + j] = acc[tc][tp][r] * scale;
      }
  }
}

// One workgroup per query plane: level0 (H0 x W0) -> level1..3 with F.avg_pool2d(2, stride 2)
// semantics (floor on odd sizes; sum of the 4 taps in row-major order, then * 0.25).
__global__ __launch_bounds__(256) void corr_pool_kernel(const float* __restrict__ l0, float* __restrict__ l1,
                                                        float* __restrict__ l2, float* __restrict__ l3, int H0,
                                                        int W0) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int H1 = H0 >> 1, W1 = W0 >> 1, H2 = H1 >> 1, W2 = W1 >> 1, H3 = H2 >> 1, W3 = W2 >> 1;
  float* s1 = sm;
  float* s2 = sm + H1 * W1;
  const long long plane = blockIdx.x;
  const float* p0 = l0 + plane * H0 * W0;
  float* o1 = l1 + plane * H1 * W1;
  float* o2 = l2 + plane * H2 * W2;
  float* o3 = l3 + plane * H3 * W3;
  const bool even = (W0 & 1) == 0;
  for (int idx = threadIdx.x; idx < H1 * W1; idx += blockDim.x) {
    const int y = idx / W1, x = idx - y * W1;
    const float* r0 = p0 + (2 * y) * W0 + 2 * x;
    float a, b, c, d;
    if (even) {
      const float2 t0 = *reinterpret_cast<const float2*>(r0);
      const float2 t1 = *reinterpret_cast<const float2*>(r0 + W0);
      a = t0.x; b = t0.y; c = t1.x; d = t1.y;
    } else {
      a = r0[0]; b = r0[1]; c = r0[W0]; d = r0[W0 + 1];
    }
    const float v = (((a + b) + c) + d) * 0.25f;
    s1[idx] = v;
    o1[idx] = v;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < H2 * W2; idx += blockDim.x) {
    const int y = idx / W2, x = idx - y * W2;
    const float* r0 = s1 + (2 * y) * W1 + 2 * x;
    const float v = (((r0[0] + r0[1]) + r0[W1]) + r0[W1 + 1]) * 0.25f;
    s2[idx] = v;
    o2[idx] = v;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < H3 * W3; idx += blockDim.x) {
    const int y = idx / W3, x = idx - y * W3;
    const float* r0 = s2 + (2 * y) * W2 + 2 * x;
    o3[idx] = (((r0[0] + r0[1]) + r0[W2]) + r0[W2 + 1]) * 0.25f;
  }
}

}  // namespace

int accflow_gemm_atb_f32(const float* A, const float* Bm, float* C, int M, int N, int K, long long a_bs,
                         long long b_bs, long long c_bs, int batch, float scale, hipStream_t st) {
  dim3 grid(cdiv(N, 128), cdiv(M, 128), batch);
  hipLaunchKernelGGL((gemm_atb_f32_kernel<2, 2>), grid, dim3(256), 0, st, A, Bm, C, M, N, K, a_bs, b_bs, c_bs,
                     scale);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

int accflow_corr_level0_bf16s(const float* fmap1, const float* fmap2, float* lvl0, void* ws, int B, int C, int H8,
                              int W8, int mode, int disp, int* guard, hipStream_t st, float* lvl1, int* lvl1_done);
extern "C" int accflow_corr_disp_supported(int H8, int W8);
extern "C" int accflow_corr_disp_pool_f32(const float* lvl0, float* lvl1, float* lvl2, float* lvl3, int B, int H8,
                                          int W8, void* stream);
int accflow_corr_disp_pool_from(const float* lvl0, float* lvl1, float* lvl2, float* lvl3, int B, int H8, int W8, int first,
                                hipStream_t st);

static int corr_volume_impl(const float* fmap1, const float* fmap2, float* lvl0, float* lvl1, float* lvl2, float* lvl3,
                            void* ws, int mode, int B, int C, int H8, int W8, void* stream) {
  if (mode == ACCFLOW_CONV_F16X3) mode = ACCFLOW_CONV_BF16X6;
  if (!fmap1 || !fmap2 || !lvl0 || !lvl1 || !lvl2 || !lvl3 || B <= 0 || C <= 0 || H8 < 8 || W8 < 8) return 1;
  hipStream_t st = as_stream(stream);
  const int P = H8 * W8;
  int rc;
  if (ws && mode != ACCFLOW_CONV_F32) {
    rc = accflow_corr_level0_bf16s(fmap1, fmap2, lvl0, ws, B, C, H8, W8, mode, 0, nullptr, st, nullptr, nullptr);
  } else {
    // corr / torch.sqrt(torch.tensor(dim).float())  (raft/corr.py:55)
    const float scale = 1.0f / sqrtf((float)C);
    rc = accflow_gemm_atb_f32(fmap1, fmap2, lvl0, P, P, C, (long long)C * P, (long long)C * P, (long long)P * P, B, scale,
                              st);
  }
  if (rc) return rc;
  const int H1 = H8 >> 1, W1 = W8 >> 1, H2 = H1 >> 1, W2 = W1 >> 1;
  const size_t smem = (size_t)(H1 * W1 + H2 * W2) * sizeof(float);
  if (smem > 64 * 1024) return 1;
  hipLaunchKernelGGL(corr_pool_kernel, dim3((unsigned)((long long)B * P)), dim3(256), smem, st, lvl0, lvl1, lvl2,
                     lvl3, H8, W8);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_corr_volume_f32(const float* fmap1, const float* fmap2, float* lvl0, float* lvl1,
                                       float* lvl2, float* lvl3, int B, int C, int H8, int W8, void* stream) {
  return corr_volume_impl(fmap1, fmap2, lvl0, lvl1, lvl2, lvl3, nullptr, ACCFLOW_CONV_F32, B, C, H8, W8, stream);
}

extern "C" long long accflow_corr_volume_ws_bytes(int C, int H8, int W8) {
  const long long Kpad = (C + 31) / 32 * 32, CoutPad = ((long long)H8 * W8 + 127) / 128 * 128;
  return 2 * (3 * Kpad * CoutPad * 2) + Kpad * 16;  // split packs of fmap1 and (displaced layout) fmap2, k-table
}

extern "C" int accflow_corr_volume_split_f32(const float* fmap1, const float* fmap2, float* lvl0, float* lvl1,
                                             float* lvl2, float* lvl3, void* ws, int mode, int B, int C, int H8,
                                             int W8, void* stream) {
  if (!ws) return 1;
  return corr_volume_impl(fmap1, fmap2, lvl0, lvl1, lvl2, lvl3, ws, mode, B, C, H8, W8, stream);
}

// The same pyramid in the displacement-indexed layout of corr_disp.hip (split-bf16 modes only: level 0 is written
// by the matrix-core kernel's displaced epilogue).  Level l holds B * (H8>>l) * (W8>>l) * H8*W8 floats.
extern "C" int accflow_corr_volume_disp_f32(const float* fmap1, const float* fmap2, float* lvl0, float* lvl1,
                                            float* lvl2, float* lvl3, void* ws, int mode, int* guard, int B, int C,
                                            int H8, int W8, void* stream) {
  if (!fmap1 || !fmap2 || !lvl0 || !lvl1 || !lvl2 || !lvl3 || !ws || B <= 0 || C <= 0) return 1;
  if (mode != ACCFLOW_CONV_BF16X3 && mode != ACCFLOW_CONV_BF16X6 && mode != ACCFLOW_CONV_F16X3) return 1;
  if (!accflow_corr_disp_supported(H8, W8)) return 1;
  int lvl1_done = 0;
  const int rc = accflow_corr_level0_bf16s(fmap1, fmap2, lvl0, ws, B, C, H8, W8, mode, 1, guard, as_stream(stream), lvl1,
                                           &lvl1_done);
  if (rc) return rc;
  // the register-only GEMM wrote level 1 with level 0 (bit-identical to pooling level 0): pool from level 1 on
  return accflow_corr_disp_pool_from(lvl0, lvl1, lvl2, lvl3, B, H8, W8, lvl1_done ? 1 : 0, as_stream(stream));
}
