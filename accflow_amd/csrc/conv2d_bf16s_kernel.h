// The im2col split-operand kernel (template) and its per-tile-shape launcher - included by the translation units that
// instantiate one tile shape each (conv2d_bf16s.hip: 128 x 128 + the displaced-store form; conv2d_bf16s_v*.hip: the other
// shapes), so that the ~20 instantiations compile in parallel instead of in one 15-minute translation unit.
#pragma once
#include "conv_common.h"

namespace {

template <bool F16>
__device__ __forceinline__ f32x16 s_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// F16: operands as fp16 hi + lo (NT = 2) of the scaled values - weights from accflow_conv_pack_split16, activations
// times 2^ACCFLOW_F16_ASHIFT with the range guard (see include/accflow_hip.h); the epilogue undoes the scales.
template <int TC, int TP, int NT, int BK, bool DISP = false, bool F16 = false>
__global__ __launch_bounds__(256) void conv2d_bf16s_kernel(const accflow_conv_desc d) {
  static_assert(!F16 || (NT == 2 && !DISP), "the fp16 split has two terms");
  constexpr int WC = 2, WP = 2;
  constexpr int BC = WC * TC * 32, BP = WP * TP * 32;
  constexpr int OCT = BK / 8;            // 8-deep k chunks per slab
  constexpr int KG = 256 / BP;           // thread groups along k
  constexpr int OPT = OCT / KG;          // octets gathered per thread per slab
  constexpr int XPT = OPT * 8;
  constexpr int WCH = NT * OCT * BC;     // 16-B weight chunks per slab
  constexpr int WPT = (WCH + 255) / 256;
  static_assert(OPT == 1 || OPT == 2, "tile / slab shape");
  constexpr int MAIN_BYTES = 2 * NT * OCT * (BC + BP) * 16;
  constexpr int LDS_BYTES = DISP && DISP_LDS_BYTES > MAIN_BYTES ? DISP_LDS_BYTES : MAIN_BYTES;
  static_assert(!DISP || (TC == 2 && TP == 2), "the displaced store is written for the 128 x 128 tile");
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  u32x4 (&Ws)[2][NT][OCT][BC] = *reinterpret_cast<u32x4 (*)[2][NT][OCT][BC]>(smem);
  u32x4 (&Xs)[2][NT][OCT][BP] = *reinterpret_cast<u32x4 (*)[2][NT][OCT][BP]>(smem + 2 * NT * OCT * BC * 16);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave / WP, wp = wave % WP;
  // Workgroups are handed to the 8 XCDs round-robin in launch order.  With several channel tiles per pixel tile, the
  // launch order is remapped so that the channel tiles of ONE pixel tile are consecutive on ONE XCD: they gather the
  // same activations (for the GMA aggregation: the same 3.7 MB column block of the attention) through that XCD's L2.
  // (K-parts, gridDim.z > 1, are the slowest index of the same order.)
  int bx = blockIdx.x, by = blockIdx.y, bz = 0;
  if (!DISP && (gridDim.y > 1 || gridDim.z > 1)) {
    const int plane = gridDim.x * gridDim.y, n = plane * gridDim.z;
    const int l = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int q = n >> 3, r = n & 7, xcd = l & 7, slot = l >> 3;
    const int logical = xcd < r ? xcd * (q + 1) + slot : r * (q + 1) + (xcd - r) * q + slot;
    bz = logical / plane;
    const int rest = logical - bz * plane;
    bx = rest / (int)gridDim.y;
    by = rest - bx * (int)gridDim.y;
  }
  const int cblk0 = by * BC;
  const int OHW = d.OH * d.OW;
  const int Ptot = d.B * OHW;
  const int px_local = tid % BP, kg = tid / BP;
  XLoaderCtx cx;
  {
    const int p = bx * BP + px_local;
    cx.pvalid = p < Ptot;
    const int pb = cx.pvalid ? p / OHW : 0;
    const int prem = cx.pvalid ? p - pb * OHW : 0;
    const int oy = prem / d.OW, ox = prem - oy * d.OW;
    cx.iy0 = cx.pvalid ? oy * d.stride - d.padH : -(1 << 28);
    cx.ix0 = ox * d.stride - d.padW;
    cx.H = d.H; cx.W = d.W; cx.HW = d.H * d.W;
    cx.pixbyte0 = (unsigned)(((long long)pb * d.in0_bs + cx.iy0 * d.W + cx.ix0) * 4);
    cx.pixbyte1 = (unsigned)(((long long)pb * d.in1_bs + cx.iy0 * d.W + cx.ix0) * 4);
    cx.OHW = OHW; cx.KW = d.KW; cx.off = nullptr; cx.dmk = nullptr;
  }
  const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.in0), 0, (int)(unsigned)((((long long)(d.B - 1)) * d.in0_bs + (long long)d.C0 * cx.HW) * 4),
      0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.in1 ? d.in1 : d.in0), 0,
      (int)(unsigned)(d.in1 ? (((long long)(d.B - 1)) * d.in1_bs + (long long)d.C1 * cx.HW) * 4 : 0), 0x00020000);
  const ktab_ptr ktab = as_ktab(d.ktab);
  // per-batch-item weights (GMA aggregation: v[b] is the weight matrix of pair b): the launcher guarantees that a
  // pixel tile never straddles two batch items, so the item is workgroup-uniform
  const u32x4* __restrict__ wsplit = reinterpret_cast<const u32x4*>(
      reinterpret_cast<const char*>(F16 ? d.wsplit16 : d.wsplit) + (d.wsplit_bs ? (long long)((bx * BP) / OHW) * d.wsplit_bs : 0));
  const int kthr = __builtin_amdgcn_readfirstlane(kg * XPT);
  const int K8 = d.Kpad / 8;

  float xr[XPT];
  u32x4 wr[WPT];
  bool bad = false;
  constexpr float ASC = (float)(1 << ACCFLOW_F16_ASHIFT);
  f32x16 acc[TC][TP];
#pragma unroll
  for (int tc = 0; tc < TC; ++tc)
#pragma unroll
    for (int tp = 0; tp < TP; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tc][tp][r] = 0.0f;

#define BF_LOAD_SLAB(KBASE)                                                                       \
  do {                                                                                            \
    gather_x<XPT>(cx, ktab, (KBASE) + kthr, rsrc0, rsrc1, xr);                                    \
    _Pragma("unroll") for (int j = 0; j < WPT; ++j) {                                             \
      const int v = tid + j * 256;                                                                \
      const int ch = v % BC, o = (v / BC) % OCT, t = v / (BC * OCT);                              \
      if ((j + 1) * 256 <= WCH || v < WCH)                                                        \
        wr[j] = wsplit[((long long)t * K8 + (KBASE) / 8 + o) * d.CoutPad + cblk0 + ch];           \
    }                                                                                             \
  } while (0)
#define BF_STORE_SLAB(BUF)                                                                        \
  do {                                                                                            \
    {                                                                                             \
      u32x4 terms[NT];                                                                            \
      if constexpr (F16) split8_f16<0>(xr, terms, bad, ASC);                                      \
      else split8_bf16<NT, 0>(xr, terms);                                                         \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) Xs[BUF][t][kg * OPT][px_local] = terms[t];   \
      if constexpr (OPT == 2) {                                                                   \
        if constexpr (F16) split8_f16<8 * (OPT - 1)>(xr, terms, bad, ASC);                        \
        else split8_bf16<NT, 8 * (OPT - 1)>(xr, terms);                                           \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) Xs[BUF][t][kg * OPT + 1][px_local] = terms[t]; \
      }                                                                                           \
    }                                                                                             \
    _Pragma("unroll") for (int j = 0; j < WPT; ++j) {                                             \
      const int v = tid + j * 256;                                                                \
      const int ch = v % BC, o = (v / BC) % OCT, t = v / (BC * OCT);                              \
      if ((j + 1) * 256 <= WCH || v < WCH) Ws[BUF][t][o][ch] = wr[j];                             \
    }                                                                                             \
  } while (0)

  // split-K (gridDim.z parts of the slab range; raw partial sums, finished by conv_ksplit_reduce_kernel)
  const int nslab_all = d.Kpad / BK;
  const int slab0 = (int)((long long)nslab_all * bz / (int)gridDim.z);
  const int nslab = (int)((long long)nslab_all * (bz + 1) / (int)gridDim.z) - slab0;
  const int l31 = lane & 31, kh = lane >> 5;
  BF_LOAD_SLAB(slab0 * BK);
  BF_STORE_SLAB(0);
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const int cur = s & 1;
    const bool more = s + 1 < nslab;
    if (more) BF_LOAD_SLAB((slab0 + s + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 a[NT][TC], b[NT][TP];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int tc = 0; tc < TC; ++tc)
          a[t][tc] = __builtin_bit_cast(bf16x8, Ws[cur][t][2 * ks + kh][wc * TC * 32 + tc * 32 + l31]);
#pragma unroll
        for (int tp = 0; tp < TP; ++tp)
          b[t][tp] = __builtin_bit_cast(bf16x8, Xs[cur][t][2 * ks + kh][wp * TP * 32 + tp * 32 + l31]);
      }
#pragma unroll
      for (int tc = 0; tc < TC; ++tc)
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) {
          f32x16 c = acc[tc][tp];
          if constexpr (NT == 3) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][tc], b[0][tp], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][tc], b[2][tp], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][tc], b[1][tp], c, 0, 0, 0);
          }
          c = s_mfma<F16>(a[NT == 3 ? 1 : 1][tc], b[0][tp], c);
          c = s_mfma<F16>(a[0][tc], b[1][tp], c);
          c = s_mfma<F16>(a[0][tc], b[0][tp], c);
          acc[tc][tp] = c;
        }
    }
    if (more) BF_STORE_SLAB(cur ^ 1);
    __syncthreads();
  }
#undef BF_LOAD_SLAB
#undef BF_STORE_SLAB
  if constexpr (F16) {
    if (bad && d.guard) atomicOr(d.guard, 1);
  }
  if constexpr (DISP) {
    corr_disp_store(d, acc, reinterpret_cast<float*>(smem), reinterpret_cast<int*>(smem) + 64 * DISP_PITCH, cblk0, wc, wp,
                    lane, wave, tid, [&](int j) {
                      const int q = bx * 128 + j;
                      return q < d.OH * d.OW ? q : -1;
                    });
  } else if (gridDim.z > 1) {
    accflow_conv_desc e = d;
    e.out = d.kws + (long long)bz * d.B * d.Cout * OHW;
    e.out_bs = (long long)d.Cout * OHW;
    e.bias = nullptr; e.wscale16 = nullptr; e.pre = nullptr;  // (applied by the reduce kernel)
    conv_epilogue_impl<ACCFLOW_EPI_STORE, ACCFLOW_ACT_NONE, WC, WP, TC, TP>(e, acc, cblk0, wc, wp, lane, OHW, [&](int j, int& b) {
      const int p = bx * BP + j;
      if (p >= Ptot) return -1;
      b = p / OHW;
      return p - b * OHW;
    });
  } else {
    conv_epilogue<WC, WP, TC, TP>(d, acc, cblk0, wc, wp, lane, OHW, Ptot, bx);
  }
}

// Two restructurings of this kernel were built, verified and measured slower on MI355X (bf16x6, B = 11 update-block
// shapes; this kernel: 134-148 TFLOP/s): (1) wave specialisation - 4 MFMA-only consumer waves + 4 staging producer
// waves per 512-thread workgroup, 2-stage LDS ring, 3 register sets of prefetch: 107-112 (one workgroup per CU, and
// hipcc's waitcnt insertion falls back to vmcnt(0) across the rotating sets); (2) in-wave software pipelining -
// weights by LDS-DMA into a 3-stage ring, split/gather of the next slabs pinned between the MFMAs with
// sched_group_barrier, raw s_barrier + counted vmcnt: 101-126.  Both are in the git history (round 1).  The LDS-patch
// kernel below (tap-major K, ~8x fewer staging instructions per MFMA) lands at the SAME throughput, and so does a
// variant of it that prefetches the next step's fragments into a second register set behind a 4-stage weight ring
// (387 vs 380 us on 3x3 128->256, B=11).  PMC for that shape: matrix pipe 40 % busy, LDS array 16 % busy (a third
// of it bank conflicts of the patch reads), 2.1 GHz; compile-time ablation: MFMA + barrier only 211 us, + fragment
// reads 299 us, + staging 380 us.  None of the latency-hiding restructurings moved the total, i.e. the limiter is
// not a latency that more overlap inside a wave removes; open question for the next round.

template <int TC, int TP>
int launch_conv_bf16s_grid(const accflow_conv_desc& d, dim3 grid, hipStream_t st);

template <int TC, int TP>
int launch_conv_bf16s(const accflow_conv_desc& d, hipStream_t st) {
  constexpr int BC = 2 * TC * 32, BP = 2 * TP * 32;
  const long long Ptot = (long long)d.B * d.OH * d.OW;
  // statistics: a pixel tile must not straddle two batch items; 2 waves along the pixels = 2 slots per tile
  const int OHW = d.OH * d.OW;
  ACCFLOW_DRY_RUN(OHW % BP == 0 ? OHW / BP * 2 : 0);
  if (d.stats && OHW % BP) return 1;
  dim3 grid(cdiv(Ptot, BP), cdiv(d.Cout, BC));
  // split-K for deep reductions on grids that do not fill the 768 workgroup slots (3 per CU): the GMA aggregation
  // (K = h*w = 14 400 at 720x1280, 225 pixel tiles per item).  The part count with the best fill of whole rounds wins.
  int Z = 1;
  const long long nout = Ptot * d.Cout;
  if (d.kws && !d.stats && d.Kpad >= 2048 && (long long)grid.x * grid.y < 700) {
    static const int zenv = [] { const char* e = getenv("ACCFLOW_IM2COL_KSPLIT"); return e ? atoi(e) : 0; }();
    const long long nb = (long long)grid.x * grid.y;
    double best = 0.0;
    for (int z = 1; z <= 8; ++z) {
      const double fill = (double)(nb * z) / (double)(cdiv(nb * z, 768) * 768);
      if (fill > best + 0.03) { best = fill; Z = z; }
    }
    if (zenv > 0) Z = zenv;
    if ((long long)Z * nout > d.kws_elems) Z = (int)(d.kws_elems / nout);
    if (Z < 1) Z = 1;
    grid.z = Z;
  }
  const int rc = launch_conv_bf16s_grid<TC, TP>(d, grid, st);
  if (rc || Z == 1) return rc;
  return conv_ksplit_reduce_launch(d, Z, st);
}

template <int TC, int TP>
int launch_conv_bf16s_grid(const accflow_conv_desc& d, dim3 grid, hipStream_t st) {
  if (d.mode == ACCFLOW_CONV_F16X3 && d.wsplit16) {
    // slab depth (measured, one box, us per launch at working size, 16 / 32): 7x7 s2 stem (K = 147) 173 / 190,
    // 3x3 s2 64->96 (K = 576) 239 / 216, 3x3 s2 96->128 88 / 82: shallow reductions take 16-deep slabs
    static const int bkenv = [] { const char* e = getenv("ACCFLOW_IM2COL_F16_BK"); return e ? atoi(e) : 0; }();
    const bool bk32 = bkenv ? bkenv == 32 : (d.C0 + d.C1) * d.KH * d.KW >= 256;
    if constexpr (TP == 2) {
      if (bk32) hipLaunchKernelGGL((conv2d_bf16s_kernel<TC, TP, 2, 32, false, true>), grid, dim3(256), 0, st, d);
      else hipLaunchKernelGGL((conv2d_bf16s_kernel<TC, TP, 2, 16, false, true>), grid, dim3(256), 0, st, d);
    } else {
      hipLaunchKernelGGL((conv2d_bf16s_kernel<TC, TP, 2, 32, false, true>), grid, dim3(256), 0, st, d);
    }
  } else if (d.mode == ACCFLOW_CONV_BF16X6) {
    if constexpr (TP == 2) hipLaunchKernelGGL((conv2d_bf16s_kernel<TC, TP, 3, 16>), grid, dim3(256), 0, st, d);
    else hipLaunchKernelGGL((conv2d_bf16s_kernel<TC, TP, 3, 32>), grid, dim3(256), 0, st, d);
  } else {
    hipLaunchKernelGGL((conv2d_bf16s_kernel<TC, TP, 2, 32>), grid, dim3(256), 0, st, d);
  }
  ACCFLOW_RETURN_LAUNCH_STATUS();
}


}  // namespace

int accflow_launch_conv_bf16s_11(const accflow_conv_desc& d, hipStream_t st);   //  64 ch x  64 px
int accflow_launch_conv_bf16s_12(const accflow_conv_desc& d, hipStream_t st);   //  64 ch x 128 px
int accflow_launch_conv_bf16s_21(const accflow_conv_desc& d, hipStream_t st);   // 128 ch x  64 px
