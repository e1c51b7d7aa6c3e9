// fp32-input MFMA implicit-GEMM convolution (conv mode "f32", deformable convolution, Cout <= 32 in every mode): the kernel
// template.  Included by conv2d_f32.hip (the dispatcher + one instantiation group) and conv2d_f32_v*.hip (one group of tile
// shapes each), so that the 14 instantiations compile in parallel - in one translation unit they took 6.5 minutes and bounded
// a from-scratch build; the unit carries no hot-path time.
#pragma once
#include "conv_common.h"

namespace {

// ---- staging helpers (free functions with array references: lambdas capturing register arrays made
// hipcc spill the weight tile to scratch) --------------------------------------------------------------

// weight tile [BK][BC] <- wpack rows kbase..kbase+BK, columns cblk0..cblk0+BC, as float4 per thread
template <int BC, int WPT, int BK>
__device__ __forceinline__ void load_w(const float* __restrict__ wpack, int CoutPad, int kbase, int cblk0, int tid,
                                       f32x4 (&wr)[WPT]) {
  constexpr int WV = BK * BC / 4;
#pragma unroll
  for (int j = 0; j < WPT; ++j) {
    const int v = tid + j * 256;
    if ((j + 1) * 256 <= WV || v < WV) {
      const int krow = v / (BC / 4), c4 = v % (BC / 4);
      wr[j] = *reinterpret_cast<const f32x4*>(wpack + (long long)(kbase + krow) * CoutPad + cblk0 + c4 * 4);
    }
  }
}
template <int BC, int WPT, int BK>
__device__ __forceinline__ void store_w(float* __restrict__ Ws, int tid, const f32x4 (&wr)[WPT]) {
  constexpr int WV = BK * BC / 4;
#pragma unroll
  for (int j = 0; j < WPT; ++j) {
    const int v = tid + j * 256;
    if ((j + 1) * 256 <= WV || v < WV) {
      const int krow = v / (BC / 4), c4 = v % (BC / 4);
      *reinterpret_cast<f32x4*>(&Ws[krow * BC + c4 * 4]) = wr[j];
    }
  }
}

// im2col gather of XPT consecutive k rows for this thread's pixel.  All table entries are fetched with
// scalar loads first; each element is then ONE buffer_load_dword whose per-lane byte offset is forced to
// 0xFFFFFFFF when the tap falls into the zero padding (or the pixel is past the end): the buffer
// descriptor's range check returns 0 for it, so there is no branch, no select, and the loads stay in
// flight under the MFMAs of the current slab until the registers are written to LDS.
// The k-table is read through the CONSTANT address space: it is never written while a conv runs, and that is
// what lets hipcc keep these wave-uniform loads on the scalar unit (s_load_dwordx8/x16) even inside loops
// that contain barriers and global stores - as plain global loads they turn into VMEM loads whose
// `s_waitcnt vmcnt` drains the prefetched gathers.
template <int WC, int WP, int TC, int TP, bool DEFORM, int BK = MMA_BK, int MINW = 1>
__global__ __launch_bounds__(256, MINW) void conv2d_f32_kernel(const accflow_conv_desc d) {
  constexpr int BC = WC * TC * 32, BP = WP * TP * 32;
  static_assert(WC * WP == 4, "4 waves per workgroup");
  static_assert(BP == 64 || BP == 128 || BP == 256, "pixel tile");
  constexpr int KG = 256 / BP;   // thread groups along k for the activation tile
  constexpr int XPT = BK / KG;   // activation elements per thread per slab
  constexpr int WV = BK * BC / 4;  // float4s of the weight tile
  constexpr int WPT = (WV + 255) / 256;
  __shared__ __attribute__((aligned(16))) float Ws[2][BK * BC];
  __shared__ __attribute__((aligned(16))) float Xs[2][BK * BP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave / WP, wp = wave % WP;
  const int cblk0 = blockIdx.y * BC;
  const int OHW = d.OH * d.OW;
  const int Ptot = d.B * OHW;

  // --- this thread's pixel of the activation tile ---
  const int px_local = tid % BP, kg = tid / BP;
  XLoaderCtx cx;
  {
    const int p = blockIdx.x * BP + px_local;
    cx.pvalid = p < Ptot;
    const int pb = cx.pvalid ? p / OHW : 0;
    const int prem = cx.pvalid ? p - pb * OHW : 0;
    const int oy = prem / d.OW, ox = prem - oy * d.OW;
    // pixels past the end of the tensor get an iy0 no tap can bring back in range: no per-element test
    cx.iy0 = cx.pvalid ? oy * d.stride - d.padH : -(1 << 28);
    cx.ix0 = ox * d.stride - d.padW;
    cx.H = d.H; cx.W = d.W; cx.HW = d.H * d.W;
    cx.pixbyte0 = (unsigned)(((long long)pb * d.in0_bs + cx.iy0 * d.W + cx.ix0) * 4);
    cx.pixbyte1 = (unsigned)(((long long)pb * d.in1_bs + cx.iy0 * d.W + cx.ix0) * 4);
    cx.base0 = d.in0 + (long long)pb * d.in0_bs;
    cx.base1 = d.in1 ? d.in1 + (long long)pb * d.in1_bs : cx.base0;
    cx.OHW = OHW; cx.KW = d.KW;
    if constexpr (DEFORM) {
      cx.off = d.offset + (long long)pb * d.offset_bs + prem;
      cx.dmk = d.dmask + (long long)pb * d.dmask_bs + prem;
    } else {
      cx.off = nullptr; cx.dmk = nullptr;
    }
  }
  const ktab_ptr ktab = as_ktab(d.ktab);
  const float* __restrict__ wpack = d.wpack;
  const int kthr = __builtin_amdgcn_readfirstlane(kg * XPT);  // wave-uniform first k row of this thread

  // buffer descriptors over the two sources (wave-uniform: built from kernel arguments only)
  const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.in0), 0, (int)(unsigned)((((long long)(d.B - 1)) * d.in0_bs + (long long)d.C0 * cx.HW) * 4),
      0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.in1 ? d.in1 : d.in0), 0,
      (int)(unsigned)(d.in1 ? (((long long)(d.B - 1)) * d.in1_bs + (long long)d.C1 * cx.HW) * 4 : 0), 0x00020000);

  float xr[XPT];
  f32x4 wr[WPT];

#define ACCFLOW_LOAD_SLAB(KBASE)                                                            \
  do {                                                                                      \
    if constexpr (DEFORM) {                                                                 \
      _Pragma("unroll") for (int i = 0; i < XPT; ++i)                                       \
          { const i32x4 ee = ktab[(KBASE) + kthr + i];                                      \
            xr[i] = load_x_deform(cx, make_int4(ee.x, ee.y, ee.z, ee.w)); }                 \
    } else {                                                                                \
      gather_x<XPT>(cx, ktab, (KBASE) + kthr, rsrc0, rsrc1, xr);                            \
    }                                                                                       \
    load_w<BC, WPT, BK>(wpack, d.CoutPad, (KBASE), cblk0, tid, wr);                             \
  } while (0)
#define ACCFLOW_STORE_SLAB(BUF)                                                             \
  do {                                                                                      \
    _Pragma("unroll") for (int i = 0; i < XPT; ++i)                                         \
        Xs[BUF][(kg * XPT + i) * BP + px_local] = xr[i];                                    \
    store_w<BC, WPT, BK>(Ws[BUF], tid, wr);                                                     \
  } while (0)

  f32x16 acc[TC][TP];
#pragma unroll
  for (int tc = 0; tc < TC; ++tc)
#pragma unroll
    for (int tp = 0; tp < TP; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tc][tp][r] = 0.0f;

  const int nslab = d.Kpad / BK;  // Kpad is a multiple of 32
  ACCFLOW_LOAD_SLAB(0);
  ACCFLOW_STORE_SLAB(0);
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const int cur = s & 1;
    const bool more = s + 1 < nslab;
    if (more) ACCFLOW_LOAD_SLAB((s + 1) * BK);
    mma_slab<TC, TP, BC, BP, BK>(Ws[cur], Xs[cur], acc, wc * TC * 32, wp * TP * 32, lane);
    if (more) ACCFLOW_STORE_SLAB(cur ^ 1);
    __syncthreads();
  }
#undef ACCFLOW_LOAD_SLAB
#undef ACCFLOW_STORE_SLAB

  conv_epilogue<WC, WP, TC, TP>(d, acc, cblk0, wc, wp, lane, OHW, Ptot, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 variant: the same implicit GEMM on the bf16 matrix cores (16x the fp32-MFMA rate) with every
// fp32 operand split on the fly into NT round-to-nearest bf16 terms, x = x0 + x1 (+ x2), and the product
// expanded to the leading cross terms with fp32 accumulation:
//   NT = 2 ("bf16x3"): w0x0 + w0x1 + w1x0                 3 MFMAs, |error| <~ 3 * 2^-16 per product
//   NT = 3 ("bf16x6"): + w1x1 + w0x2 + w2x0               6 MFMAs, |error| <~ 2^-23 per product
// bf16 keeps fp32's exponent range, so there is no overflow / subnormal hazard (unlike an fp16 hi/lo split).
// Weights are pre-split at pack time into [term][k/8][channel][8] (a lane's 8 consecutive k are one 16-B
// chunk = its MFMA fragment); activations are gathered as fp32 exactly like the fp32 kernel - each thread
// owns 8 consecutive k of one pixel, i.e. exactly one B-operand fragment - split in registers and written
// to LDS as 16-B chunks [term][k/8][pixel].  Fragment reads are conflict-free ds_read_b128.
template <int WC, int WP, int TC, int TP>
int launch_conv(const accflow_conv_desc& d, hipStream_t st) {
  constexpr int BC = WC * TC * 32, BP = WP * TP * 32;
  const long long Ptot = (long long)d.B * d.OH * d.OW;
  dim3 grid(cdiv(Ptot, BP), cdiv(d.Cout, BC));
  if (d.offset) {
    hipLaunchKernelGGL((conv2d_f32_kernel<WC, WP, TC, TP, true>), grid, dim3(256), 0, st, d);
  } else if (TC * TP == 4) {
    // 64 accumulator registers per lane: cap the rest so that 4 waves/SIMD stay resident (measured
    // 106 -> 112 TFLOP/s on the 128x128 tile; BK = 32 at 2 waves/SIMD measured 96)
    hipLaunchKernelGGL((conv2d_f32_kernel<WC, WP, TC, TP, false, MMA_BK, 4>), grid, dim3(256), 0, st, d);
  } else {
    hipLaunchKernelGGL((conv2d_f32_kernel<WC, WP, TC, TP, false>), grid, dim3(256), 0, st, d);
  }
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

}  // namespace

