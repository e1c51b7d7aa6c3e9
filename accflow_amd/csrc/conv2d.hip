// Direct (implicit-GEMM) fp32 convolution on the gfx950 fp32-input matrix cores.
//
// Replaces every cuDNN conv the reference's inference path reaches through nn.Conv2d
// (raft/update.py:9-10,37-43,83-87,122-125; raft/extractor.py:9-15,52,140-158; AccFlow_.py:16-25,
// 51-53,71-95,115-120; gma/update.py:117-125) and torchvision.ops.deform_conv2d (AccFlow_.py:83,104).
//
// Formulation: D[ch x px] = Wp[k x ch]^T * X[k x px], k = (c, ky, kx) flattened, px = (b, oy, ox)
// flattened.  Output channels ride the MFMA A operand and pixels the B operand, so accumulator
// column = lane&31 = pixel and every store instruction writes two 128-B runs of consecutive pixels of
// an NCHW plane (coalesced without an LDS transpose).  Both operands are staged k-major in LDS
// ([k][ch], [k][px]): with the 32x32x2 fp32 MFMA each lane needs ONE float of each per instruction,
// read by conflict-free ds_read_b32.  fp32-in / fp32-acc MFMA is a bitwise fmaf chain, so the result
// matches an fp32 CPU convolution to summation-order rounding.
//
// Staging: the activation tile is an im2col gather.  Each thread owns one pixel of the tile for the
// whole kernel (its (b, oy, ox) is decoded once); the k row it loads is wave-uniform, so the per-k
// descriptor {channel, ky, kx, source} comes from a scalar load of a small table and the per-element
// work is two adds, two unsigned compares and one predicated dword load, coalesced along x.
// Registers double-buffer the next slab while the current one feeds the MFMAs (one barrier per slab).
#include "common.h"
#include <stdlib.h>

namespace {

struct XLoaderCtx {
  const float* base0;
  const float* base1;
  int iy0, ix0, H, W, HW;
  unsigned pixbyte0, pixbyte1;  // byte offset of (b, iy0, ix0) inside source 0 / 1 (mod 2^32)
  bool pvalid;
  // deformable mode
  const float* off;   // offset + b*offset_bs + prem
  const float* dmk;   // dmask  + b*dmask_bs  + prem
  int OHW, KW;
};

// torchvision deform_conv2d (modulated): sample (y + dy_t, x + dx_t), dy first; whole sample is 0 when
// h <= -1 || h >= H || w <= -1 || w >= W; per-corner zeros otherwise.
__device__ __forceinline__ float load_x_deform(const XLoaderCtx& c, const int4 e) {  // e: {channel, ky, kx, source}
  const float* src = e.w ? c.base1 : c.base0;
  if (!c.pvalid || e.y >= (1 << 19)) return 0.0f;
  const int tap = e.y * c.KW + e.z;
  const float dy = c.off[(2 * tap) * c.OHW], dx = c.off[(2 * tap + 1) * c.OHW];
  const float m = c.dmk[tap * c.OHW];
  const float h = (float)(c.iy0 + e.y) + dy, w = (float)(c.ix0 + e.z) + dx;
  if (!(h > -1.0f && h < (float)c.H && w > -1.0f && w < (float)c.W)) return 0.0f;
  const float* plane = src + e.x * c.HW;
  const float fh = floorf(h), fw = floorf(w);
  const int hl = (int)fh, wl = (int)fw, hh = hl + 1, wh = wl + 1;
  const float lh = h - fh, lw = w - fw, uh = 1.0f - lh, uw = 1.0f - lw;
  const float v1 = (hl >= 0 && wl >= 0) ? plane[hl * c.W + wl] : 0.0f;
  const float v2 = (hl >= 0 && wh <= c.W - 1) ? plane[hl * c.W + wh] : 0.0f;
  const float v3 = (hh <= c.H - 1 && wl >= 0) ? plane[hh * c.W + wl] : 0.0f;
  const float v4 = (hh <= c.H - 1 && wh <= c.W - 1) ? plane[hh * c.W + wh] : 0.0f;
  return m * (uh * uw * v1 + uh * lw * v2 + lh * uw * v3 + lh * lw * v4);
}

// Epilogue shared by the fp32 and the split-bf16 kernels (same accumulator layout: 32x32 tiles, row = channel,
// column = pixel): bias, activation, fused GRU / residual math, NCHW stores of 32 consecutive pixels per half-wave.
// PixMap: (local pixel index in [0, BP)) -> batch index b and offset `rem` inside one (OH, OW) plane, or rem < 0.
//
// History, from in-kernel timestamps (ACCFLOW_KPROF): the first form - one fully unrolled generic loop with the
// epi / act switches, 64-bit address arithmetic and a load -> wait -> store round trip per element - was 13 000+
// instructions of straight-line code per kernel and took 15-22 % of a workgroup's lifetime, 15 us of 100 even with
// the stores removed.  This form keeps an element at ~10 instructions:
//   * every tensor is addressed through a range-checked buffer descriptor with a per-lane 32-bit pixel offset
//     (0xFFFFFFFF = masked: outside the image, or a channel >= Cout) plus a wave-uniform SCALAR channel offset, so
//     there is no per-element vector address arithmetic and no exec-mask branch;
//   * the activation is a template parameter (4 copies of the element code instead of an inlined expf / tanhf
//     chain per element);
//   * gfx950 counts loads and stores in ONE in-order vmcnt, so a load issued after a store cannot be waited for
//     without waiting for that store's acknowledgement: all bias values are loaded before the first store and the
//     e0 / e1 operands of group g+1 are requested before the stores of group g (counted waits only).
// d.out may alias d.e0 / d.e1 element for element (in-place GRU state): a group's operands are read before any
// store of that group or a later one.
template <int EPI, int ACT, int WC, int WP, int TC, int TP, class PixMap>
__device__ __forceinline__ void conv_epilogue_impl(const accflow_conv_desc& d, f32x16 (&acc)[TC][TP], int cblk0, int wc,
                                                   int wp, int lane, int OHW, PixMap pixmap) {
  constexpr unsigned MASKED = 0xFFFFFFFFu;
  const int l31 = lane & 31, lh4 = (lane >> 5) * 4;
  const int epi = EPI >= 0 ? EPI : d.epi;  // EPI < 0: read from the descriptor (combinations the estimators do not use)
  const int half = d.Cout >> 1;
  const bool has_h = epi != ACCFLOW_EPI_STORE, has_z = epi == ACCFLOW_EPI_GRU_Q, zr = epi == ACCFLOW_EPI_GRU_ZR;
  const int nout = zr ? half : d.Cout;  // channels of d.out
  auto span = [&](long long bs, int nch) { return (int)(unsigned)((((long long)(d.B - 1)) * bs + (long long)nch * OHW) * 4); };
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(d.out, 0, span(d.out_bs, nout), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_o2 =
      __builtin_amdgcn_make_buffer_rsrc(zr ? d.out2 : d.out, 0, zr ? span(d.out2_bs, half) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_e0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(has_h ? d.e0 : d.out), 0, has_h ? span(d.e0_bs, zr ? half : d.Cout) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_e1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(has_z ? d.e1 : d.out), 0, has_z ? span(d.e1_bs, d.Cout) : 0, 0x00020000);

  // per-lane byte offsets of (batch item, pixel, + the 4-row step of the upper half-wave) in each tensor
  unsigned vo_out[TP], vo_o2[TP], vo_e0[TP], vo_e1[TP];
#pragma unroll
  for (int tp = 0; tp < TP; ++tp) {
    int b;
    const int rem = pixmap(wp * TP * 32 + tp * 32 + l31, b);
    const bool ok = rem >= 0;
    const long long lp = (long long)rem + (long long)lh4 * OHW;
    vo_out[tp] = ok ? (unsigned)((b * d.out_bs + lp) * 4) : MASKED;
    vo_o2[tp] = ok && zr ? (unsigned)((b * d.out2_bs + lp) * 4) : MASKED;
    vo_e0[tp] = ok && has_h ? (unsigned)((b * d.e0_bs + lp) * 4) : MASKED;
    vo_e1[tp] = ok && has_z ? (unsigned)((b * d.e1_bs + lp) * 4) : MASKED;
  }
  const int rowbase = cblk0 + wc * TC * 32;  // first channel of this wave's rows (wave-uniform)
  const int OHW4 = OHW * 4;
  // bias through SCALAR loads (lgkmcnt: independent of the stores' vmcnt), requested one group ahead - waiting for
  // them inside their own group cost a full SMEM latency per group, 13 us of a 100 us workgroup lifetime
  typedef const __attribute__((address_space(4))) float* cfloat_ptr;
  const cfloat_ptr sbias = (cfloat_ptr)(unsigned long long)d.bias;
  float sb0[2], sb1[2];
  // One GROUP = accumulator row r of tile tc for both pixel tiles: channel chu = rowbase + tc*32 + (r&3) + 8*(r>>2)
  // in the lower half-wave, chu + 4 in the upper one.  Operands of group g+1 are requested before the stores of g.
  float h[2][TP], z[2][TP];
#define EPI_CHU(G) (rowbase + ((G) / 16) * 32 + ((G) & 3) + 8 * (((G) & 15) >> 2))
#define EPI_FETCH(G, HH, ZZ)                                                                                     \
  do {                                                                                                           \
    if (has_h) {                                                                                                 \
      const int chu_ = EPI_CHU(G);                                                                               \
      const int che_ = zr ? chu_ - half : chu_;                                                                  \
      const bool live_ = che_ >= 0;                                                                              \
      const bool in_ = chu_ + lh4 < d.Cout;                                                                      \
      _Pragma("unroll") for (int tp = 0; tp < TP; ++tp) {                                                        \
        HH[tp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(                                 \
            r_e0, (int)((live_ && in_) ? vo_e0[tp] : MASKED), live_ ? che_ * OHW4 : 0, 0));                      \
        if (has_z) ZZ[tp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(                      \
            r_e1, (int)(in_ ? vo_e1[tp] : MASKED), chu_ * OHW4, 0));                                             \
      }                                                                                                          \
    }                                                                                                            \
  } while (0)
#define EPI_BIAS(G, S)                                                                                           \
  do {                                                                                                           \
    const int chu_ = EPI_CHU(G);                                                                                 \
    sb0[S] = d.bias ? sbias[min(chu_, d.Cout - 1)] : 0.0f;                                                       \
    sb1[S] = d.bias ? sbias[min(chu_ + 4, d.Cout - 1)] : 0.0f;                                                   \
  } while (0)
  EPI_FETCH(0, h[0], z[0]);
  EPI_BIAS(0, 0);
#pragma unroll
  for (int g = 0; g < TC * 16; ++g) {
    __builtin_amdgcn_sched_barrier(0);
    const float bv = lh4 ? sb1[g & 1] : sb0[g & 1];
    __builtin_amdgcn_sched_barrier(0);
    if (g + 1 < TC * 16) {
      EPI_FETCH(g + 1, h[(g + 1) & 1], z[(g + 1) & 1]);
      EPI_BIAS(g + 1, (g + 1) & 1);
    }
    const int tc = g / 16, r = g & 15;
    const int chu = EPI_CHU(g);
    const bool in = chu + lh4 < d.Cout;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) {
      const float v = apply_act(acc[tc][tp][r] + bv, ACT);
#ifdef ACCFLOW_KPROF_NOSTORE
      if (v != 12345.678f) continue;
#endif
      const float hh = h[g & 1][tp], zz = z[g & 1][tp];
      float o = v;
      if (epi == ACCFLOW_EPI_RES_RELU) o = fmaxf(hh + v, 0.0f);
      else if (epi == ACCFLOW_EPI_GRU_Q) o = (1.0f - zz) * hh + zz * v;
      else if (epi == ACCFLOW_EPI_ACCUM) o = hh + v;
      if (zr && chu >= half) {  // r gate rows (Cout % 16 == 0: both half-waves on the same side): r * h into out2
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v * hh), r_o2, (int)(in ? vo_o2[tp] : MASKED),
                                              (chu - half) * OHW4, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), r_out, (int)(in ? vo_out[tp] : MASKED),
                                              chu * OHW4, 0);
      }
    }
  }
#undef EPI_BIAS
#undef EPI_FETCH
#undef EPI_CHU
}

template <int WC, int WP, int TC, int TP, class PixMap>
__device__ __forceinline__ void conv_epilogue_px(const accflow_conv_desc& d, f32x16 (&acc)[TC][TP], int cblk0, int wc,
                                                 int wp, int lane, int OHW, PixMap pixmap) {
  // the (epilogue, activation) pairs the estimators use are compiled as straight-line code (update.py, extractor.py,
  // AccFlow_.py mirrors); any other pair takes the descriptor-driven copy
#define ACCFLOW_EPI_CASE(E, A)                                                                          \
  case (E) * 8 + (A): conv_epilogue_impl<E, A, WC, WP, TC, TP>(d, acc, cblk0, wc, wp, lane, OHW, pixmap); break;
  switch (d.epi * 8 + d.act) {
    ACCFLOW_EPI_CASE(ACCFLOW_EPI_STORE, ACCFLOW_ACT_NONE)
    ACCFLOW_EPI_CASE(ACCFLOW_EPI_STORE, ACCFLOW_ACT_RELU)
    ACCFLOW_EPI_CASE(ACCFLOW_EPI_STORE, ACCFLOW_ACT_SIGMOID)
    ACCFLOW_EPI_CASE(ACCFLOW_EPI_RES_RELU, ACCFLOW_ACT_RELU)
    ACCFLOW_EPI_CASE(ACCFLOW_EPI_GRU_ZR, ACCFLOW_ACT_SIGMOID)
    ACCFLOW_EPI_CASE(ACCFLOW_EPI_GRU_Q, ACCFLOW_ACT_TANH)
    ACCFLOW_EPI_CASE(ACCFLOW_EPI_ACCUM, ACCFLOW_ACT_NONE)
    default:
      switch (d.act) {
        case ACCFLOW_ACT_RELU: conv_epilogue_impl<-1, ACCFLOW_ACT_RELU, WC, WP, TC, TP>(d, acc, cblk0, wc, wp, lane, OHW, pixmap); break;
        case ACCFLOW_ACT_SIGMOID: conv_epilogue_impl<-1, ACCFLOW_ACT_SIGMOID, WC, WP, TC, TP>(d, acc, cblk0, wc, wp, lane, OHW, pixmap); break;
        case ACCFLOW_ACT_TANH: conv_epilogue_impl<-1, ACCFLOW_ACT_TANH, WC, WP, TC, TP>(d, acc, cblk0, wc, wp, lane, OHW, pixmap); break;
        default: conv_epilogue_impl<-1, ACCFLOW_ACT_NONE, WC, WP, TC, TP>(d, acc, cblk0, wc, wp, lane, OHW, pixmap);
      }
  }
#undef ACCFLOW_EPI_CASE
}

// flattened (b, oy, ox) pixel tiles: local pixel j of workgroup blockIdx.x is global pixel blockIdx.x*BP + j
template <int WC, int WP, int TC, int TP>
__device__ __forceinline__ void conv_epilogue(const accflow_conv_desc& d, f32x16 (&acc)[TC][TP], int cblk0, int wc,
                                              int wp, int lane, int OHW, int Ptot) {
  constexpr int BP = WP * TP * 32;
  conv_epilogue_px<WC, WP, TC, TP>(d, acc, cblk0, wc, wp, lane, OHW, [&](int j, int& b) {
    const int p = blockIdx.x * BP + j;
    if (p >= Ptot) return -1;
    b = p / OHW;
    return p - b * OHW;
  });
}

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
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) i32x4* ktab_ptr;
__device__ __forceinline__ ktab_ptr as_ktab(const int* p) { return (ktab_ptr)(unsigned long long)p; }

template <int XPT>
__device__ __forceinline__ void gather_x(const XLoaderCtx& c, ktab_ptr ktab, int k0, __amdgpu_buffer_rsrc_t r0,
                                         __amdgpu_buffer_rsrc_t r1, float (&xr)[XPT]) {
  k0 = __builtin_amdgcn_readfirstlane(k0);
  i32x4 e[XPT];
#pragma unroll
  for (int i = 0; i < XPT; ++i) e[i] = ktab[k0 + i];
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int iy = c.iy0 + e[i].y, ix = c.ix0 + e[i].z;
    const bool ok = (unsigned)iy < (unsigned)c.H && (unsigned)ix < (unsigned)c.W;  // pvalid folded into iy0
    const unsigned koff = (unsigned)(e[i].x * c.HW + e[i].y * c.W + e[i].z) * 4u;  // wave-uniform
    const unsigned off = ok ? (e[i].w ? c.pixbyte1 : c.pixbyte0) + koff : 0xFFFFFFFFu;
    xr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(e[i].w ? r1 : r0, (int)off, 0, 0));
  }
}

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

  conv_epilogue<WC, WP, TC, TP>(d, acc, cblk0, wc, wp, lane, OHW, Ptot);
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
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NT, int OFF, int N>
__device__ __forceinline__ void split8_bf16(const float (&x)[N], u32x4 (&out)[NT]) {
  float r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = x[OFF + j];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    unsigned w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x2 v = {r[2 * j], r[2 * j + 1]};
      const bf16x2 b = __builtin_convertvector(v, bf16x2);  // v_cvt_pk_bf16_f32, round to nearest even
      w[j] = __builtin_bit_cast(unsigned, b);
      if (t + 1 < NT) {
        r[2 * j] -= __builtin_bit_cast(float, w[j] << 16);
        r[2 * j + 1] -= __builtin_bit_cast(float, w[j] & 0xFFFF0000u);
      }
    }
    { u32x4 v4 = {w[0], w[1], w[2], w[3]}; out[t] = v4; }
  }
}

// Displaced store of a 128 x 128 all-pairs correlation tile (rows = query pixel p, the "channel" side; columns =
// target pixel q), layout E_0[dy][dx][p] of corr_disp.hip: dy = (y2 - y1) mod H8, dx = (x2 - x1) mod W8.  Elements of
// one output row lie on a DIAGONAL of the tile, so the accumulators go through LDS - T[q][p], 64 target columns at
// a time - and are read back with lane = target column, p = (q - u) mod 128 for the wave-uniform diagonal u: the 64
// lanes of a store then hold consecutive p of (normally) one (dy, dx) row, 256 contiguous bytes.  Both LDS passes
// are bank-conflict free (row pitch 132 words: 16-B writes land on 4q + c, reads on 5*lane + c).
constexpr int DISP_PITCH = 132;
constexpr int DISP_LDS_BYTES = (64 * DISP_PITCH + 128) * 4;

__device__ __forceinline__ void corr_disp_store(const accflow_conv_desc& d, f32x16 (&acc)[2][2], float* T, int* tab,
                                                int cblk0, int wc, int wp, int lane, int wave, int tid) {
  const int H8 = d.OH, W8 = d.OW, P = H8 * W8;
  const int l31 = lane & 31;
  if (tid < 128) {
    const int p = cblk0 + tid;
    const int y1 = p / W8;
    tab[tid] = p < P ? (y1 << 16) | (p - y1 * W8) : -1;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (wp == h) {
#pragma unroll
      for (int tp = 0; tp < 2; ++tp)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4 v = {acc[tc][tp][4 * r4], acc[tc][tp][4 * r4 + 1], acc[tc][tp][4 * r4 + 2], acc[tc][tp][4 * r4 + 3]};
            *reinterpret_cast<f32x4*>(&T[(tp * 32 + l31) * DISP_PITCH + wc * 64 + tc * 32 + 8 * r4 + 4 * (lane >> 5)]) = v;
          }
    }
    __syncthreads();
    const int q = blockIdx.x * 128 + h * 64 + lane;
    const int y2 = q / W8, x2 = q - y2 * W8;
    const bool qok = q < P;
    for (int it = 0; it < 32; ++it) {
      const int u = wave * 32 + it;
      const int pl = (h * 64 + lane - u) & 127;
      const float v = T[lane * DISP_PITCH + pl];
      const int t = tab[pl];
      if (qok && t >= 0) {
        int dy = y2 - (t >> 16), dx = x2 - (t & 0xFFFF);
        if (dy < 0) dy += H8;
        if (dx < 0) dx += W8;
        d.out[(long long)(dy * W8 + dx) * P + cblk0 + pl] = v;
      }
    }
    if (h == 0) __syncthreads();
  }
}

template <int TC, int TP, int NT, int BK, bool DISP = false>
__global__ __launch_bounds__(256) void conv2d_bf16s_kernel(const accflow_conv_desc d) {
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
  const int cblk0 = blockIdx.y * BC;
  const int OHW = d.OH * d.OW;
  const int Ptot = d.B * OHW;
  const int px_local = tid % BP, kg = tid / BP;
  XLoaderCtx cx;
  {
    const int p = blockIdx.x * BP + px_local;
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
      reinterpret_cast<const char*>(d.wsplit) + (d.wsplit_bs ? (long long)((blockIdx.x * BP) / OHW) * d.wsplit_bs : 0));
  const int kthr = __builtin_amdgcn_readfirstlane(kg * XPT);
  const int K8 = d.Kpad / 8;

  float xr[XPT];
  u32x4 wr[WPT];
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
      split8_bf16<NT, 0>(xr, terms);                                                              \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) Xs[BUF][t][kg * OPT][px_local] = terms[t];   \
      if constexpr (OPT == 2) {                                                                   \
        split8_bf16<NT, 8 * (OPT - 1)>(xr, terms);                                                \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) Xs[BUF][t][kg * OPT + 1][px_local] = terms[t]; \
      }                                                                                           \
    }                                                                                             \
    _Pragma("unroll") for (int j = 0; j < WPT; ++j) {                                             \
      const int v = tid + j * 256;                                                                \
      const int ch = v % BC, o = (v / BC) % OCT, t = v / (BC * OCT);                              \
      if ((j + 1) * 256 <= WCH || v < WCH) Ws[BUF][t][o][ch] = wr[j];                             \
    }                                                                                             \
  } while (0)

  const int nslab = d.Kpad / BK;
  const int l31 = lane & 31, kh = lane >> 5;
  BF_LOAD_SLAB(0);
  BF_STORE_SLAB(0);
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const int cur = s & 1;
    const bool more = s + 1 < nslab;
    if (more) BF_LOAD_SLAB((s + 1) * BK);
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
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][tc], b[0][tp], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][tc], b[1][tp], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][tc], b[0][tp], c, 0, 0, 0);
          acc[tc][tp] = c;
        }
    }
    if (more) BF_STORE_SLAB(cur ^ 1);
    __syncthreads();
  }
#undef BF_LOAD_SLAB
#undef BF_STORE_SLAB
  if constexpr (DISP) {
    corr_disp_store(d, acc, reinterpret_cast<float*>(smem), reinterpret_cast<int*>(smem) + 64 * DISP_PITCH, cblk0, wc, wp,
                    lane, wave, tid);
  } else {
    conv_epilogue<WC, WP, TC, TP>(d, acc, cblk0, wc, wp, lane, OHW, Ptot);
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

#ifdef ACCFLOW_KPROF
__device__ unsigned long long g_kprof[4096 * 16];
#define KP_SLOT(i) g_kprof[((blockIdx.y * gridDim.x + blockIdx.x) & 4095) * 16 + (i)]
#define KPROF_T(v)                                              \
  __builtin_amdgcn_sched_barrier(0);                            \
  const unsigned long long v = __builtin_readcyclecounter();    \
  __builtin_amdgcn_sched_barrier(0)
#define KPROF_WAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define KPROF_ACC(i, v) kp[i] += (v)
#else
#define KPROF_T(v)
#define KPROF_WAIT()
#define KPROF_ACC(i, v)
#endif

// ------------------------------------------------------------------------------------------------
// Direct-A patch kernel: stride-1 "same" convolutions on the split-bf16 matrix cores.
//
// K is ordered (16-channel chunk, tap, channel): a workgroup owns a 4 x 32 pixel tile, stages the
// (4+KH-1) x (32+KW-1) input PATCH of one chunk in LDS once - gathered, split into bf16 terms, written as 16-B
// [term][octet][patch pixel] chunks - and all KH*KW taps read their B fragments from it with a tap-dependent LDS
// offset (zero padding is materialised in the patch, so there is no per-tap bounds logic).
//
// Its predecessor (in the git history: 8 x 16 tiles, weights DMA'd by global_load_lds into a 3-stage LDS ring, one
// counted wait + barrier per step) was instrumented with in-kernel timestamps (ACCFLOW_KPROF; 3x3 128->256, B = 11:
// 2150 cycles per step and wave, 768 of them its 24 MFMAs): 35 % went into ISSUING the 3 weight DMAs (100-185
// cycles each beside MFMAs), 11 % into issuing 12 fragment reads, 10 % into the wait + barrier, and the epilogue was
// another 15-20 % of the workgroup's lifetime.  This kernel removes those terms instead of trying to overlap them:
//   * the weight (A) fragments never touch LDS: the [term][step][octet][CoutPad][8] pack IS the MFMA A layout
//     (lane l: row l&31, octet l>>5), so each wave loads its fragments of the NEXT step straight from L2 into a
//     second register set with 16-byte range-checked buffer loads (scalar step offset, no VALU) - no DMA issue,
//     no weight ring, half the fragment reads;
//   * with the weights out of LDS the only LDS hazard left is the input patch, written once per 16-channel chunk:
//     ONE barrier per chunk (KH*KW steps) instead of one per step;
//   * the pixel tile is 4 rows x 32 columns: B-fragment reads of 32 lanes are contiguous (no bank conflicts) and
//     every store instruction writes two full 128-byte lines.
// Measured after the change (same shape): 1070 cycles per step and workgroup with two workgroups per CU, i.e. the
// matrix pipe ~72 % busy inside the loop.
constexpr int DIR_TH = 4, DIR_TW = 32, DIR_NPMAX = 256;  // tile and max patch pixels (3x3: 204, 1x5: 144, 5x1: 256)

template <int TC, int NT>
__global__ __launch_bounds__(256) void conv2d_direct_bf16s_kernel(const accflow_conv_desc d) {
#ifdef ACCFLOW_KPROF
  const unsigned long long tL0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long kp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  constexpr int WC = 2, WP = 2, TP = 2, OCT = 2;
  constexpr int BC = WC * TC * 32;
  static_assert(DIR_TH * DIR_TW == WP * TP * 32, "4 x 32 pixel tile = 128 accumulator columns");
  constexpr int PSTAGE = NT * OCT * DIR_NPMAX;
  __shared__ u32x4 Pst[2 * PSTAGE];             // [2][NT][OCT][DIR_NPMAX]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave / WP, wp = wave % WP;
  const int l31 = lane & 31, kh = lane >> 5;
  const int cblk0 = blockIdx.y * BC;
  const int OHW = d.OH * d.OW;
  const int tilesX = (d.OW + DIR_TW - 1) / DIR_TW, tilesY = (d.OH + DIR_TH - 1) / DIR_TH;
  const int tb = blockIdx.x / (tilesX * tilesY), trem = blockIdx.x - tb * tilesX * tilesY;
  const int oy0 = (trem / tilesX) * DIR_TH, ox0 = (trem % tilesX) * DIR_TW;
  const int T = d.KH * d.KW;
  const int PW = DIR_TW + d.KW - 1, NP = (DIR_TH + d.KH - 1) * PW;
  const int Cin = d.C0 + d.C1;
  const int nchunk = (Cin + 15) / 16, nstep = nchunk * T;
  const int HW = d.H * d.W;

  // ---- patch staging: item it = tid + 256*i -> (octet = it / NP, patch pixel = it % NP) ----
  unsigned voff0[2], voff1[2];   // byte offset of (b, iy, ix) in source 0 / 1, 0xFFFFFFFF in the zero padding
  int p_oct[2], p_pix[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int it = tid + 256 * i;
    const bool live = it < 2 * NP;
    p_oct[i] = live ? it / NP : 0;
    p_pix[i] = live ? it - p_oct[i] * NP : 0;
    const int py = p_pix[i] / PW, px = p_pix[i] - py * PW;
    const int iy = oy0 - d.padH + py, ix = ox0 - d.padW + px;
    const bool ok = live && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
    voff0[i] = ok ? (unsigned)(((long long)tb * d.in0_bs + iy * d.W + ix) * 4) : 0xFFFFFFFFu;
    voff1[i] = ok ? (unsigned)(((long long)tb * d.in1_bs + iy * d.W + ix) * 4) : 0xFFFFFFFFu;
    if (!live) p_pix[i] = -1;
  }
  const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.in0), 0, (int)(unsigned)((((long long)(d.B - 1)) * d.in0_bs + (long long)d.C0 * HW) * 4),
      0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.in1 ? d.in1 : d.in0), 0,
      (int)(unsigned)(d.in1 ? (((long long)(d.B - 1)) * d.in1_bs + (long long)d.C1 * HW) * 4 : 0), 0x00020000);
  float xa[8], xb[8];
  auto gather_patch = [&](int cc) {
    const int c0 = cc * 16;  // first channel of the chunk (cat index); a chunk never straddles the two sources
    const bool second = c0 >= d.C0;
    const __amdgpu_buffer_rsrc_t rs = second ? rsrc1 : rsrc0;
    const int cs = second ? c0 - d.C0 : c0, cmax = second ? d.C1 : d.C0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int ca = cs + p_oct[0] * 8 + q, cb = cs + p_oct[1] * 8 + q;
      const unsigned va = second ? voff1[0] : voff0[0], vb = second ? voff1[1] : voff0[1];
      const unsigned oa = (ca < cmax && va != 0xFFFFFFFFu) ? va + (unsigned)ca * (unsigned)HW * 4u : 0xFFFFFFFFu;
      const unsigned ob = (cb < cmax && vb != 0xFFFFFFFFu) ? vb + (unsigned)cb * (unsigned)HW * 4u : 0xFFFFFFFFu;
      xa[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)oa, 0, 0));
      xb[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)ob, 0, 0));
    }
  };
  auto store_patch = [&](int stage) {
    u32x4 terms[NT];
    split8_bf16<NT, 0>(xa, terms);
    if (p_pix[0] >= 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) Pst[stage * PSTAGE + (t * OCT + p_oct[0]) * DIR_NPMAX + p_pix[0]] = terms[t];
    }
    split8_bf16<NT, 0>(xb, terms);
    if (p_pix[1] >= 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) Pst[stage * PSTAGE + (t * OCT + p_oct[1]) * DIR_NPMAX + p_pix[1]] = terms[t];
    }
  };

  // ---- A fragments: 16 bytes per lane and (term, 32-row tile) straight from the pack ----
  const long long step_bytes = 2LL * d.CoutPad * 16, term_bytes = (long long)nstep * step_bytes;
  const __amdgpu_buffer_rsrc_t rsrcw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.wpatch), 0,
                                                                        (int)(unsigned)(3 * term_bytes), 0x00020000);
  const unsigned avoff = (unsigned)((kh * d.CoutPad + cblk0 + wc * TC * 32 + l31) * 16);
#define DIR_LOAD_A(STEP, A)                                                                                      \
  _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int tc = 0; tc < TC; ++tc)               \
      A[t][tc] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(                              \
          rsrcw, (int)(avoff + tc * 512), (int)(unsigned)(t * term_bytes + (STEP) * step_bytes), 0))

  // this lane's two accumulator-column pixels inside the patch (tap (0,0)): column j -> row j / 32, col j % 32
  int pbase[TP];
#pragma unroll
  for (int tp = 0; tp < TP; ++tp) pbase[tp] = (wp * TP + tp) * PW + l31;

  f32x16 acc[TC][TP];
#pragma unroll
  for (int tc = 0; tc < TC; ++tc)
#pragma unroll
    for (int tp = 0; tp < TP; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tc][tp][r] = 0.0f;

  bf16x8 aA[NT][TC], aB[NT][TC];
  DIR_LOAD_A(0, aA);
  gather_patch(0);
  store_patch(0);
  __syncthreads();

  int cc = 0, tap = 0, ty = 0, tx = 0;
  // one (chunk, tap) step: prefetch the next step's A, the next chunk's patch at tap 0, B fragments from the patch
  // at this tap's offset, MFMAs; at the chunk's last tap split / store the prefetched patch and synchronise.
#define DIR_STEP(STEP, ACUR, ANXT)                                                                               \
  do {                                                                                                           \
    KPROF_T(tA);                                                                                                 \
    const int pstage = cc & 1;                                                                                   \
    const bool next_chunk = cc + 1 < nchunk;                                                                     \
    if ((STEP) + 1 < nstep) { DIR_LOAD_A((STEP) + 1, ANXT); }                                                    \
    if (tap == 0 && next_chunk) gather_patch(cc + 1);                                                            \
    KPROF_T(tA1);                                                                                                \
    const int toff = ty * PW + tx;                                                                               \
    bf16x8 b[NT][TP];                                                                                            \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int tp = 0; tp < TP; ++tp)             \
        b[t][tp] = __builtin_bit_cast(bf16x8, Pst[pstage * PSTAGE + (t * OCT + kh) * DIR_NPMAX + pbase[tp] + toff]); \
    KPROF_T(tB);                                                                                                 \
    KPROF_WAIT();                                                                                                \
    KPROF_T(tB2);                                                                                                \
    {                                                                                                            \
      constexpr int NPAIR = NT == 3 ? 6 : 3;                                                                     \
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};                                      \
      _Pragma("unroll") for (int pr = 6 - NPAIR; pr < 6; ++pr) _Pragma("unroll") for (int tc = 0; tc < TC; ++tc) \
          _Pragma("unroll") for (int tp = 0; tp < TP; ++tp) acc[tc][tp] =                                        \
              __builtin_amdgcn_mfma_f32_32x32x16_bf16(ACUR[PA[pr]][tc], b[PB[pr]][tp], acc[tc][tp], 0, 0, 0);    \
    }                                                                                                            \
    KPROF_T(tC);                                                                                                 \
    if (++tx == d.KW) { tx = 0; ++ty; }                                                                          \
    if (++tap == T) {                                                                                            \
      if (next_chunk) store_patch(pstage ^ 1);                                                                   \
      KPROF_T(tD);                                                                                               \
      __syncthreads();                                                                                           \
      KPROF_T(tE);                                                                                               \
      KPROF_ACC(3, tD - tC); KPROF_ACC(4, tE - tD);                                                              \
      tap = 0; ty = 0; tx = 0; ++cc;                                                                             \
    }                                                                                                            \
    KPROF_ACC(0, tA1 - tA); KPROF_ACC(7, tB - tA1); KPROF_ACC(1, tB2 - tB); KPROF_ACC(2, tC - tB2); KPROF_ACC(5, 1); \
  } while (0)

#ifdef ACCFLOW_KPROF
  const unsigned long long tK0 = __builtin_readcyclecounter();
  const unsigned long long tR0 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) KP_SLOT(14) = tR0 - tL0;
#endif
  for (int step = 0; step < nstep; step += 2) {
    DIR_STEP(step, aA, aB);
    if (step + 1 < nstep) DIR_STEP(step + 1, aB, aA);
  }
#undef DIR_STEP
#undef DIR_LOAD_A
#ifdef ACCFLOW_KPROF
  {
    const unsigned long long tK1 = __builtin_readcyclecounter();
    if (tid == 0) {
      for (int i = 0; i < 6; ++i) KP_SLOT(i) = kp[i];
      KP_SLOT(6) = tK1 - tK0;
      KP_SLOT(7) = kp[7];
      KP_SLOT(8) = __builtin_amdgcn_s_memrealtime() - tR0;
      KP_SLOT(10) = 1;
    }
  }
#endif
  conv_epilogue_px<WC, WP, TC, TP>(d, acc, cblk0, wc, wp, lane, OHW, [&](int j, int& b) {
    const int oy = oy0 + j / DIR_TW, ox = ox0 + j % DIR_TW;
    b = tb;
    return (oy < d.OH && ox < d.OW) ? oy * d.OW + ox : -1;
  });
#ifdef ACCFLOW_KPROF
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long tS = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) {
    KP_SLOT(15) = tS - tL0;
    KP_SLOT(11) = __builtin_amdgcn_s_memrealtime() - tL0;
  }
#endif
}

// weights for the patch kernel: [3 terms][nchunk*T steps][2 octets][CoutPad][8] bf16, element (t, step = cc*T + tap,
// o, ch, q) = term t of w[ch][cc*16 + o*8 + q][tap] (* scale[ch]), zero beyond Cin / Cout
__global__ void conv_pack_patch_kernel(const float* __restrict__ w, const float* __restrict__ scale, int Cout, int Cin,
                                       int T, int CoutPad, unsigned short* __restrict__ wp) {
  const int nstep = (Cin + 15) / 16 * T;
  const long long per_term = (long long)nstep * 2 * CoutPad * 8;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= per_term) return;
  const int q = (int)(idx & 7);
  long long r = idx >> 3;
  const int ch = (int)(r % CoutPad); r /= CoutPad;
  const int o = (int)(r & 1); r >>= 1;
  const int step = (int)r, cc = step / T, tap = step % T;
  const int c = cc * 16 + o * 8 + q;
  float val = 0.0f;
  if (c < Cin && ch < Cout) {
    val = w[((long long)ch * Cin + c) * T + tap];
    if (scale) val *= scale[ch];
  }
  float rr = val;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const __bf16 bq = (__bf16)rr;
    wp[t * per_term + idx] = __builtin_bit_cast(unsigned short, bq);
    rr -= (float)bq;
  }
}

// w (OIHW fp32, optional per-channel scale) -> three bf16 terms [3][Kpad/8][CoutPad][8], k ordered (c, tap)
__global__ void conv_pack_bf16s_kernel(const float* __restrict__ w, const float* __restrict__ scale, int Cout, int Cin,
                                       int KH, int KW, int Kpad, int CoutPad, unsigned short* __restrict__ ws,
                                       int kmajor, float cscale, const float* __restrict__ gptr) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)Kpad * CoutPad) return;
  w += (long long)blockIdx.y * Cout * Cin * KH * KW;          // batched use: one weight matrix per blockIdx.y
  ws += (long long)blockIdx.y * 3 * Kpad * CoutPad;
  const int k = (int)(idx / CoutPad), o = (int)(idx % CoutPad);
  const int T = KH * KW, K = Cin * T;
  float val = 0.0f;
  if (k < K && o < Cout) {
    const int c = k / T, t = k % T;
    val = kmajor ? w[(long long)k * Cout + o] : w[((long long)o * Cin + c) * T + t];
    if (scale) val *= scale[o];
    val *= cscale;
    if (gptr) val *= gptr[0];
  }
  const long long per_term = (long long)Kpad * CoutPad;
  const long long dst = ((long long)(k / 8) * CoutPad + o) * 8 + (k % 8);
  float r = val;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const __bf16 b = (__bf16)r;
    ws[t * per_term + dst] = __builtin_bit_cast(unsigned short, b);
    r -= (float)b;
  }
}

// Convs with <= 4 output channels (flow heads, the blending mask: update.py:10, AccFlow_.py:19,118) do not
// fill even one 32-wide MFMA tile; they are pure gathers.  One workgroup = 64 pixels x 4 quarters of the
// reduction (wave q owns slabs q, q+4, ...): every lane keeps CO accumulators, weights and table entries
// are wave-uniform scalar loads, activations the same range-checked buffer loads as the MFMA kernel; the
// four partial sums meet in LDS.
template <int CO>
__global__ __launch_bounds__(256) void conv2d_small_cout_kernel(const accflow_conv_desc d) {
  __shared__ float red[4][CO][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int OHW = d.OH * d.OW;
  const int Ptot = d.B * OHW;
  const int p = blockIdx.x * 64 + lane;
  const bool pvalid = p < Ptot;
  const int pb = pvalid ? p / OHW : 0;
  const int prem = pvalid ? p - pb * OHW : 0;
  const int oy = prem / d.OW, ox = prem - oy * d.OW;
  XLoaderCtx cx;
  cx.pvalid = pvalid;
  cx.iy0 = pvalid ? oy * d.stride - d.padH : -(1 << 28);
  cx.ix0 = ox * d.stride - d.padW;
  cx.H = d.H; cx.W = d.W; cx.HW = d.H * d.W;
  cx.pixbyte0 = (unsigned)(((long long)pb * d.in0_bs + cx.iy0 * d.W + cx.ix0) * 4);
  cx.pixbyte1 = (unsigned)(((long long)pb * d.in1_bs + cx.iy0 * d.W + cx.ix0) * 4);
  const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.in0), 0, (int)(unsigned)((((long long)(d.B - 1)) * d.in0_bs + (long long)d.C0 * cx.HW) * 4),
      0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.in1 ? d.in1 : d.in0), 0,
      (int)(unsigned)(d.in1 ? (((long long)(d.B - 1)) * d.in1_bs + (long long)d.C1 * cx.HW) * 4 : 0), 0x00020000);
  const ktab_ptr ktab = as_ktab(d.ktab);
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = 0.0f;
  constexpr int U = 8;
  for (int k0 = q * U; k0 < d.Kpad; k0 += 4 * U) {
    float xr[U];
    gather_x<U>(cx, ktab, k0, rsrc0, rsrc1, xr);
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const float* __restrict__ wrow = d.wpack + (long long)(k0 + i) * d.CoutPad;  // wave-uniform row
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[c] = fmaf(wrow[c], xr[i], acc[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < CO; ++c) red[q][c][lane] = acc[c];
  __syncthreads();
  if (q != 0 || !pvalid) return;
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    if (c >= d.Cout) break;
    float v = ((red[0][c][lane] + red[1][c][lane]) + red[2][c][lane]) + red[3][c][lane];
    if (d.bias) v += d.bias[c];
    v = apply_act(v, d.act);
    const long long o = (long long)c * OHW + prem;
    if (d.epi == ACCFLOW_EPI_ACCUM) v += d.e0[pb * d.e0_bs + o];
    else if (d.epi == ACCFLOW_EPI_RES_RELU) v = fmaxf(d.e0[pb * d.e0_bs + o] + v, 0.0f);
    d.out[pb * d.out_bs + o] = v;
  }
}

// Patch variant of the <= 4-output-channel convolution for stride-1 "same" convs (flow heads 3x3 256->2, blending
// mask 3x3 256->1): the gather kernel above re-reads every activation KH*KW times from L2 (~150 us at B = 11); here an
// 8x16-pixel tile stages the fp32 input patch of a 16-channel chunk in LDS once ([ch][patch pixel], double buffered)
// and every tap reads it from there.  Thread = (pixel, half of the chunk's channels); weights are wave-uniform scalar
// loads from the fp32 pack (k = c*T + tap); the two channel halves meet in LDS at the end.
template <int CO>
__global__ __launch_bounds__(256) void conv2d_small_cout_patch_kernel(const accflow_conv_desc d) {
  constexpr int TH = 8, TW = 16, PMAXP = 192;
  constexpr int TMAX = 25;
  __shared__ float pat[2][16][PMAXP];
  __shared__ __attribute__((aligned(16))) float wsm[2][16 * TMAX * CO];  // [stage][(c*T + tap)*CO + co]
  __shared__ float red[CO][128];
  const int tid = threadIdx.x;
  const int half = __builtin_amdgcn_readfirstlane(tid >> 7);  // waves 0,1 -> channels 0..7 of a chunk, waves 2,3 -> 8..15
  const int j = tid & 127;
  const int OHW = d.OH * d.OW, HW = d.H * d.W;
  const int tilesX = (d.OW + TW - 1) / TW, tilesY = (d.OH + TH - 1) / TH;
  const int tb = blockIdx.x / (tilesX * tilesY), trem = blockIdx.x - tb * tilesX * tilesY;
  const int oy0 = (trem / tilesX) * TH, ox0 = (trem % tilesX) * TW;
  const int T = d.KH * d.KW, PW = TW + d.KW - 1, NP = (TH + d.KH - 1) * PW;
  const int Cin = d.C0 + d.C1, nchunk = (Cin + 15) / 16;
  // staging: item it = tid + 256*i over (channel-in-chunk q = it / NP... ) -> use (pixel, 8-channel octet) items like the
  // matrix-core patch kernel: octet = it / NP, patch pixel = it % NP, 8 loads each
  unsigned voff0[2], voff1[2];
  int p_oct[2], p_pix[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int it = tid + 256 * i;
    const bool live = it < 2 * NP;
    p_oct[i] = live ? it / NP : 0;
    p_pix[i] = live ? it - p_oct[i] * NP : -1;
    const int pp = live ? p_pix[i] : 0;
    const int py = pp / PW, px = pp - py * PW;
    const int iy = oy0 - d.padH + py, ix = ox0 - d.padW + px;
    const bool ok = live && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
    voff0[i] = ok ? (unsigned)(((long long)tb * d.in0_bs + iy * d.W + ix) * 4) : 0xFFFFFFFFu;
    voff1[i] = ok ? (unsigned)(((long long)tb * d.in1_bs + iy * d.W + ix) * 4) : 0xFFFFFFFFu;
  }
  const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.in0), 0, (int)(unsigned)((((long long)(d.B - 1)) * d.in0_bs + (long long)d.C0 * HW) * 4),
      0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.in1 ? d.in1 : d.in0), 0,
      (int)(unsigned)(d.in1 ? (((long long)(d.B - 1)) * d.in1_bs + (long long)d.C1 * HW) * 4 : 0), 0x00020000);
  constexpr int WN = (16 * TMAX * CO + 255) / 256;
  float xa[8], xb[8], wreg[WN];
  const float* __restrict__ wpk = d.wpack;  // [k = c*T + tap][CoutPad]
  auto gather = [&](int cc) {
    const int c0 = cc * 16;
#pragma unroll
    for (int i = 0; i < WN; ++i) {  // this chunk's 16*T*CO weights
      const int idx = tid + 256 * i;
      const int c = idx / (T * CO), r = idx - c * (T * CO);
      const bool ok = idx < 16 * T * CO && c0 + c < Cin && (r % CO) < d.Cout;
      wreg[i] = ok ? wpk[(long long)((c0 + c) * T + r / CO) * d.CoutPad + (r % CO)] : 0.0f;
    }
    const bool second = c0 >= d.C0;
    const __amdgpu_buffer_rsrc_t rs = second ? rsrc1 : rsrc0;
    const int cs = second ? c0 - d.C0 : c0, cmax = second ? d.C1 : d.C0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int ca = cs + p_oct[0] * 8 + q, cb = cs + p_oct[1] * 8 + q;
      const unsigned va = second ? voff1[0] : voff0[0], vb = second ? voff1[1] : voff0[1];
      const unsigned oa = (ca < cmax && va != 0xFFFFFFFFu) ? va + (unsigned)ca * (unsigned)HW * 4u : 0xFFFFFFFFu;
      const unsigned ob = (cb < cmax && vb != 0xFFFFFFFFu) ? vb + (unsigned)cb * (unsigned)HW * 4u : 0xFFFFFFFFu;
      xa[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)oa, 0, 0));
      xb[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)ob, 0, 0));
    }
  };
  auto store = [&](int stage) {
#pragma unroll
    for (int i = 0; i < WN; ++i)
      if (tid + 256 * i < 16 * T * CO) wsm[stage][tid + 256 * i] = wreg[i];
    if (p_pix[0] >= 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) pat[stage][p_oct[0] * 8 + q][p_pix[0]] = xa[q];
    }
    if (p_pix[1] >= 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) pat[stage][p_oct[1] * 8 + q][p_pix[1]] = xb[q];
    }
  };
  const int pbase = (j / TW) * PW + (j % TW);
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = 0.0f;
  gather(0);
  store(0);
  if (nchunk > 1) gather(1);
  __syncthreads();
  for (int cc = 0; cc < nchunk; ++cc) {
    const int st = cc & 1;
    for (int q = 0; q < 8; ++q) {
      const int c = cc * 16 + half * 8 + q;  // wave-uniform
      if (c >= Cin) break;
      const float* prow = &pat[st][half * 8 + q][pbase];
      const float* wrow = &wsm[st][(half * 8 + q) * T * CO];  // same address in every lane: LDS broadcast
      int toff = 0, kx = 0;
      for (int tap = 0; tap < T; ++tap) {
        const float v = prow[toff];
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = fmaf(wrow[tap * CO + co], v, acc[co]);
        if (++kx == d.KW) { kx = 0; toff += PW - d.KW + 1; } else { ++toff; }
      }
    }
    if (cc + 1 < nchunk) {
      store(st ^ 1);                 // gathered during the previous chunk
      if (cc + 2 < nchunk) gather(cc + 2);
    }
    __syncthreads();
  }
  if (half == 1) {
#pragma unroll
    for (int co = 0; co < CO; ++co) red[co][j] = acc[co];
  }
  __syncthreads();
  if (half != 0) return;
  const int oy = oy0 + j / TW, ox = ox0 + j % TW;
  if (oy >= d.OH || ox >= d.OW) return;
  const int rem = oy * d.OW + ox;
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    if (co >= d.Cout) break;
    float v = acc[co] + red[co][j];
    if (d.bias) v += d.bias[co];
    v = apply_act(v, d.act);
    const long long o = (long long)co * OHW + rem;
    if (d.epi == ACCFLOW_EPI_ACCUM) v += d.e0[tb * d.e0_bs + o];
    else if (d.epi == ACCFLOW_EPI_RES_RELU) v = fmaxf(d.e0[tb * d.e0_bs + o] + v, 0.0f);
    d.out[tb * d.out_bs + o] = v;
  }
}

__global__ void conv_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, int Cout,
                                 int Cin, int KH, int KW, int C0, int tap_major, int Kpad, int CoutPad,
                                 float* __restrict__ wpack, int4* __restrict__ ktab) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)Kpad * CoutPad) return;
  const int k = (int)(idx / CoutPad), o = (int)(idx % CoutPad);
  const int T = KH * KW, K = Cin * T;
  int c = 0, t = 0;
  if (k < K) {
    if (tap_major) { t = k / Cin; c = k % Cin; } else { c = k / T; t = k % T; }
  }
  const int ky = t / KW, kx = t % KW;
  float val = 0.0f;
  if (k < K && o < Cout) {
    val = w[(((long long)o * Cin + c) * KH + ky) * KW + kx];
    if (scale) val *= scale[o];
  }
  wpack[idx] = val;
  if (o == 0) {
    int4 e;
    if (k < K) { e.x = c < C0 ? c : c - C0; e.y = ky; e.z = kx; e.w = c < C0 ? 0 : 1; }
    else { e.x = 0; e.y = 1 << 20; e.z = 1 << 20; e.w = 0; }
    ktab[k] = e;
  }
}

// k-table of a 1x1, single-source conv: entry k = {channel k, 0, 0, source 0}; padding entries never in range
__global__ void conv_ktab_kernel(int Cin, int Kpad, int4* __restrict__ ktab) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= Kpad) return;
  int4 e;
  if (k < Cin) { e.x = k; e.y = 0; e.z = 0; e.w = 0; } else { e.x = 0; e.y = 1 << 20; e.z = 1 << 20; e.w = 0; }
  ktab[k] = e;
}

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

template <int TC, int TP>
int launch_conv_bf16s(const accflow_conv_desc& d, hipStream_t st) {
  constexpr int BC = 2 * TC * 32, BP = 2 * TP * 32;
  const long long Ptot = (long long)d.B * d.OH * d.OW;
  dim3 grid(cdiv(Ptot, BP), cdiv(d.Cout, BC));
  if (d.mode == ACCFLOW_CONV_BF16X6) {
    if constexpr (TP == 2) hipLaunchKernelGGL((conv2d_bf16s_kernel<TC, TP, 3, 16>), grid, dim3(256), 0, st, d);
    else hipLaunchKernelGGL((conv2d_bf16s_kernel<TC, TP, 3, 32>), grid, dim3(256), 0, st, d);
  } else {
    hipLaunchKernelGGL((conv2d_bf16s_kernel<TC, TP, 2, 32>), grid, dim3(256), 0, st, d);
  }
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

template <int TC>
int launch_conv_direct(const accflow_conv_desc& d, hipStream_t st) {
  const int tiles = cdiv(d.OW, DIR_TW) * cdiv(d.OH, DIR_TH);
  dim3 grid((unsigned)((long long)d.B * tiles), cdiv(d.Cout, 2 * TC * 32));
  if (d.mode == ACCFLOW_CONV_BF16X6) hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<TC, 3>), grid, dim3(256), 0, st, d);
  else hipLaunchKernelGGL((conv2d_direct_bf16s_kernel<TC, 2>), grid, dim3(256), 0, st, d);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

bool direct_eligible(const accflow_conv_desc& d) {
  if (!d.wpatch || d.wsplit_bs || d.mode == ACCFLOW_CONV_F32 || d.offset || d.stride != 1 || d.Cout <= 32) return false;
  if (d.OH != d.H || d.OW != d.W) return false;                                // "same" convolutions only
  if ((DIR_TH + d.KH - 1) * (DIR_TW + d.KW - 1) > DIR_NPMAX) return false;
  if (d.C0 + d.C1 < 16) return false;                                          // 2 / 3-channel stems: im2col kernel
  if (d.in1 && (d.C0 % 16)) return false;                                      // a chunk must not straddle the sources
  return true;
}

long long patch_min_blocks() {  // ACCFLOW_PATCH_MIN_BLOCKS=0 forces the patch kernel on small grids (tests)
  static const long long v = [] { const char* e = getenv("ACCFLOW_PATCH_MIN_BLOCKS"); return e ? atoll(e) : 300LL; }();
  return v;
}

}  // namespace

extern "C" long long accflow_conv_patch_elems(int Cout, int Cin, int KH, int KW) {
  return 3LL * ((Cin + 15) / 16) * KH * KW * 2 * accflow_conv_coutpad(Cout) * 8;
}

extern "C" int accflow_conv_pack_patch(const float* w, const float* scale, int Cout, int Cin, int KH, int KW,
                                       void* wpatch, void* stream) {
  if (!w || !wpatch || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return 1;
  const long long n = accflow_conv_patch_elems(Cout, Cin, KH, KW) / 3;
  hipLaunchKernelGGL(conv_pack_patch_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), w, scale, Cout, Cin,
                     KH * KW, accflow_conv_coutpad(Cout), reinterpret_cast<unsigned short*>(wpatch));
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_conv_pack_bf16s(const float* w, const float* scale, int Cout, int Cin, int KH, int KW,
                                       void* wsplit, void* stream) {
  if (!w || !wsplit || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return 1;
  const int Kpad = accflow_conv_kpad(Cin, KH, KW), CoutPad = accflow_conv_coutpad(Cout);
  const long long n = (long long)Kpad * CoutPad;
  hipLaunchKernelGGL(conv_pack_bf16s_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), w, scale, Cout, Cin, KH,
                     KW, Kpad, CoutPad, reinterpret_cast<unsigned short*>(wsplit), 0, 1.0f, nullptr);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

// All-pairs correlation as B independent 1x1 "convolutions" on the split-bf16 matrix cores: for pair b the
// K-major feature map fmap1[b] (C x P) plays the weights (C -> P output "channels"), fmap2[b] the input, so
// out[i][j] = <f1[:, i], f2[:, j]> / sqrt(C) lands directly in the (P x P) level-0 layout.  ws: Kpad*CoutPad*3
// uint16 + Kpad*4 int32 of workspace, reused pair after pair on the same stream.
// disp != 0: level 0 in the displacement-indexed layout of corr_disp.hip instead.
int accflow_corr_level0_bf16s(const float* fmap1, const float* fmap2, float* lvl0, void* ws, int B, int C, int H8,
                              int W8, int mode, int disp, hipStream_t st) {
  const int P = H8 * W8;
  const int Kpad = accflow_conv_kpad(C, 1, 1), CoutPad = accflow_conv_coutpad(P);
  unsigned short* wsplit = reinterpret_cast<unsigned short*>(ws);
  int* ktab = reinterpret_cast<int*>(wsplit + 3LL * Kpad * CoutPad);
  const float cscale = 1.0f / sqrtf((float)C);
  for (int b = 0; b < B; ++b) {
    const long long n = (long long)Kpad * CoutPad;
    hipLaunchKernelGGL(conv_pack_bf16s_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, fmap1 + (long long)b * C * P, nullptr,
                       P, C, 1, 1, Kpad, CoutPad, wsplit, 1, cscale, nullptr);
    if (b == 0) hipLaunchKernelGGL(conv_ktab_kernel, dim3(cdiv(Kpad, 256)), dim3(256), 0, st, C, Kpad, reinterpret_cast<int4*>(ktab));
    accflow_conv_desc d = {};
    d.in0 = fmap2 + (long long)b * C * P; d.in0_bs = (long long)C * P; d.C0 = C; d.C1 = 0;
    d.B = 1; d.H = H8; d.W = W8; d.OH = H8; d.OW = W8; d.KH = 1; d.KW = 1; d.stride = 1; d.padH = 0; d.padW = 0;
    d.Cout = P; d.wpack = reinterpret_cast<const float*>(wsplit) /* unused in split modes */; d.ktab = ktab;
    d.Kpad = Kpad; d.CoutPad = CoutPad; d.out = lvl0 + (long long)b * P * P; d.out_bs = (long long)P * P;
    d.act = ACCFLOW_ACT_NONE; d.epi = ACCFLOW_EPI_STORE; d.wsplit = wsplit; d.mode = mode;
    if (disp) {
      dim3 grid(cdiv(P, 128), cdiv(P, 128));
      if (mode == ACCFLOW_CONV_BF16X6) hipLaunchKernelGGL((conv2d_bf16s_kernel<2, 2, 3, 16, true>), grid, dim3(256), 0, st, d);
      else hipLaunchKernelGGL((conv2d_bf16s_kernel<2, 2, 2, 32, true>), grid, dim3(256), 0, st, d);
      continue;
    }
    const int rc = accflow_conv2d_f32(&d, st);
    if (rc) return rc;
  }
  return (int)hipGetLastError();
}

// GMA aggregation (gma/modules.py:102-115) as B independent 1x1 convolutions on the split-bf16 matrix cores:
// out[b][d][i] = fmap[b][d][i] + gamma * sum_j v[b][d][j] * attnT[b][j][i].  The TRANSPOSED attention (j-major) is
// exactly a (1, P channels, h, w) activation tensor, v[b] the (D x P) weight matrix (re-split every call, gamma folded
// in), and the residual add is the conv's accumulate epilogue.  ws as accflow_corr_volume_ws_bytes-style scratch:
// 3*Kpad*CoutPad uint16 + Kpad int4.
int accflow_gma_aggregate_conv(const float* attnT, const float* v, const float* fmap, const float* gamma, float* out,
                               long long out_bs, void* ws, int mode, int B, int D, int H, int W, hipStream_t st) {
  const int P = H * W;
  const int Kpad = accflow_conv_kpad(P, 1, 1), CoutPad = accflow_conv_coutpad(D);
  unsigned short* wsplit = reinterpret_cast<unsigned short*>(ws);
  int* ktab = reinterpret_cast<int*>(wsplit + 3LL * Kpad * CoutPad * B);
  hipLaunchKernelGGL(conv_ktab_kernel, dim3(cdiv(Kpad, 256)), dim3(256), 0, st, P, Kpad, reinterpret_cast<int4*>(ktab));
  const long long n = (long long)Kpad * CoutPad;
  hipLaunchKernelGGL(conv_pack_bf16s_kernel, dim3(cdiv(n, 256), B), dim3(256), 0, st, v, nullptr, D, P, 1, 1, Kpad, CoutPad,
                     wsplit, 0, 1.0f, gamma);
  accflow_conv_desc d = {};
  d.in0 = attnT; d.in0_bs = (long long)P * P; d.C0 = P; d.C1 = 0;
  d.B = B; d.H = H; d.W = W; d.OH = H; d.OW = W; d.KH = 1; d.KW = 1; d.stride = 1;
  d.Cout = D; d.wpack = reinterpret_cast<const float*>(wsplit); d.ktab = ktab; d.Kpad = Kpad; d.CoutPad = CoutPad;
  d.out = out; d.out_bs = out_bs;
  d.act = ACCFLOW_ACT_NONE; d.epi = ACCFLOW_EPI_ACCUM; d.e0 = fmap; d.e0_bs = (long long)D * P;
  d.wsplit = wsplit; d.wsplit_bs = 3LL * Kpad * CoutPad * 2; d.mode = mode;
  if (P % 64 == 0) {  // 64-pixel tiles never straddle two pairs: as few launches as 32-bit buffer offsets allow
    const long long per_pair = (long long)P * P * 4;
    const int chunk = (int)(((1LL << 32) - 1) / per_pair);
    if (chunk < 1) return 1;
    for (int b0 = 0; b0 < B; b0 += chunk) {
      accflow_conv_desc e = d;
      e.B = B - b0 < chunk ? B - b0 : chunk;
      e.in0 = attnT + (long long)b0 * P * P; e.out = out + (long long)b0 * out_bs; e.e0 = fmap + (long long)b0 * D * P;
      e.wsplit = wsplit + 3LL * Kpad * CoutPad * b0;
      const int rc = accflow_conv2d_f32(&e, st);
      if (rc) return rc;
    }
    return 0;
  }
  for (int b = 0; b < B; ++b) {                           // ragged sizes: one launch per pair
    accflow_conv_desc e = d;
    e.B = 1; e.in0 = attnT + (long long)b * P * P; e.out = out + (long long)b * out_bs; e.e0 = fmap + (long long)b * D * P;
    e.wsplit = wsplit + 3LL * Kpad * CoutPad * b; e.wsplit_bs = 0;
    const int rc = accflow_conv2d_f32(&e, st);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int accflow_conv_kpad(int Cin, int KH, int KW) {
  const int K = Cin * KH * KW;
  return (K + 31) / 32 * 32;  // multiple of every slab depth in use (16 and 32)
}

extern "C" int accflow_conv_coutpad(int Cout) { return (Cout + 127) / 128 * 128; }

extern "C" int accflow_conv_pack_f32(const float* w, const float* scale, int Cout, int Cin, int KH, int KW,
                                     int C0, int tap_major, float* wpack, int* ktab, void* stream) {
  if (!w || !wpack || !ktab || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return 1;
  const int Kpad = accflow_conv_kpad(Cin, KH, KW), CoutPad = accflow_conv_coutpad(Cout);
  const long long n = (long long)Kpad * CoutPad;
  hipLaunchKernelGGL(conv_pack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), w, scale, Cout,
                     Cin, KH, KW, C0, tap_major, Kpad, CoutPad, wpack, reinterpret_cast<int4*>(ktab));
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_conv2d_f32(const accflow_conv_desc* desc, void* stream) {
  if (!desc) return 1;
  const accflow_conv_desc& d = *desc;
  if (!d.in0 || !d.wpack || !d.ktab || !d.out || d.B <= 0 || d.Cout <= 0 || d.OH <= 0 || d.OW <= 0) return 1;
  if (d.Kpad != accflow_conv_kpad(d.C0 + d.C1, d.KH, d.KW) || d.CoutPad != accflow_conv_coutpad(d.Cout)) return 1;
  if ((d.epi == ACCFLOW_EPI_RES_RELU || d.epi == ACCFLOW_EPI_ACCUM) && !d.e0) return 1;
  if (d.epi == ACCFLOW_EPI_GRU_ZR && (!d.e0 || !d.out2 || (d.Cout & 1))) return 1;
  if (d.epi == ACCFLOW_EPI_GRU_Q && (!d.e0 || !d.e1)) return 1;
  if (d.offset && !d.dmask) return 1;
  if ((long long)d.B * d.OH * d.OW >= (1LL << 31)) return 1;
  // sources are addressed through 32-bit buffer offsets: each must span < 4 GiB (callers chunk the batch)
  if ((((long long)(d.B - 1)) * d.in0_bs + (long long)d.C0 * d.H * d.W) * 4 >= (1LL << 32)) return 1;
  if (d.in1 && (((long long)(d.B - 1)) * d.in1_bs + (long long)d.C1 * d.H * d.W) * 4 >= (1LL << 32)) return 1;
  hipStream_t st = as_stream(stream);
  const long long Ptot = (long long)d.B * d.OH * d.OW;
  if (d.Cout <= 4 && !d.offset && (d.epi == ACCFLOW_EPI_STORE || d.epi == ACCFLOW_EPI_ACCUM || d.epi == ACCFLOW_EPI_RES_RELU)) {
    const bool same = d.stride == 1 && d.OH == d.H && d.OW == d.W && d.KH * d.KW >= 2 && d.C0 + d.C1 >= 16 &&
                      (8 + d.KH - 1) * (16 + d.KW - 1) <= 192 && d.KH * d.KW <= 25 && (!d.in1 || d.C0 % 16 == 0);
    if (same) {
      dim3 pgrid((unsigned)((long long)d.B * cdiv(d.OW, 16) * cdiv(d.OH, 8)));
      if (d.Cout <= 2) hipLaunchKernelGGL((conv2d_small_cout_patch_kernel<2>), pgrid, dim3(256), 0, st, d);
      else hipLaunchKernelGGL((conv2d_small_cout_patch_kernel<4>), pgrid, dim3(256), 0, st, d);
      ACCFLOW_RETURN_LAUNCH_STATUS();
    }
    dim3 grid(cdiv(Ptot, 64));
    if (d.Cout <= 2) hipLaunchKernelGGL((conv2d_small_cout_kernel<2>), grid, dim3(256), 0, st, d);
    else hipLaunchKernelGGL((conv2d_small_cout_kernel<4>), grid, dim3(256), 0, st, d);
    ACCFLOW_RETURN_LAUNCH_STATUS();
  }
  if (direct_eligible(d)) {
    const long long nb = (long long)d.B * cdiv(d.OW, DIR_TW) * cdiv(d.OH, DIR_TH);
    if (d.Cout > 64 && nb * cdiv(d.Cout, 128) >= patch_min_blocks()) return launch_conv_direct<2>(d, st);  // 128 ch
    if (nb * cdiv(d.Cout, 64) >= patch_min_blocks()) return launch_conv_direct<1>(d, st);                  //  64 ch
  }
  if (d.wsplit_bs) {  // per-batch-item weights: 64-pixel tiles that never straddle items
    if (d.mode == ACCFLOW_CONV_F32 || !d.wsplit || d.offset || ((d.OH * d.OW) % 64) || d.Cout <= 32) return 1;
    return d.Cout > 64 ? launch_conv_bf16s<2, 1>(d, st) : launch_conv_bf16s<1, 1>(d, st);
  }
  if (d.mode != ACCFLOW_CONV_F32 && d.wsplit && !d.offset && d.Cout > 32) {
    // split-bf16 matrix-core path (k order must be (c, tap): the tap-major pack is deformable-only)
    auto nb = [&](int bc, int bp) { return (long long)cdiv(Ptot, bp) * cdiv(d.Cout, bc); };
    if (d.Cout <= 64) return nb(64, 128) >= 384 ? launch_conv_bf16s<1, 2>(d, st) : launch_conv_bf16s<1, 1>(d, st);
    if (d.Cout % 192 == 0 && d.Cout % 128 != 0 && nb(192, 128) >= 384) return launch_conv_bf16s<3, 2>(d, st);  // 192 x 128
    if (nb(128, 128) >= 384) return launch_conv_bf16s<2, 2>(d, st);
    if (nb(128, 64) >= 384) return launch_conv_bf16s<2, 1>(d, st);
    return launch_conv_bf16s<1, 1>(d, st);
  }
  // Tile choice: the largest tile that still yields >= MIN_BLOCKS workgroups (256 CUs x ~1.5), since the
  // fusion chain runs at batch 1 (7 680 pixels) where 128x128 tiles would leave most CUs idle.
  constexpr long long MIN_BLOCKS = 384;
  auto blocks = [&](int bc, int bp) { return (long long)cdiv(Ptot, bp) * cdiv(d.Cout, bc); };
  if (d.Cout <= 32) {
    if (blocks(32, 256) >= MIN_BLOCKS) return launch_conv<1, 4, 1, 2>(d, st);   // 32 ch x 256 px
    return launch_conv<1, 4, 1, 1>(d, st);                                       // 32 ch x 128 px
  }
  if (d.Cout <= 64) {
    if (blocks(64, 128) >= MIN_BLOCKS) return launch_conv<2, 2, 1, 2>(d, st);   // 64 ch x 128 px
    return launch_conv<2, 2, 1, 1>(d, st);                                       // 64 ch x 64 px
  }
  if (d.Cout % 96 == 0 && d.Cout % 128 != 0 && blocks(96, 128) >= MIN_BLOCKS)
    return launch_conv<1, 4, 3, 1>(d, st);                                       // 96 ch x 128 px
  if (blocks(128, 128) >= MIN_BLOCKS) return launch_conv<2, 2, 2, 2>(d, st);    // 128 ch x 128 px
  if (blocks(128, 64) >= MIN_BLOCKS) return launch_conv<2, 2, 2, 1>(d, st);     // 128 ch x 64 px
  return launch_conv<2, 2, 1, 1>(d, st);                                         // 64 ch x 64 px
}

#ifdef ACCFLOW_KPROF
extern "C" int accflow_debug_occupancy(int* out) {
  int n = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[n++], conv2d_direct_bf16s_kernel<2, 3>, 256, 0);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[n++], conv2d_direct_bf16s_kernel<2, 2>, 256, 0);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[n++], conv2d_direct_bf16s_kernel<1, 3>, 256, 0);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[n++], conv2d_bf16s_kernel<2, 2, 3, 16>, 256, 0);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[n++], conv2d_bf16s_kernel<2, 1, 3, 32>, 256, 0);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[n++], conv2d_bf16s_kernel<1, 2, 3, 16>, 256, 0);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[n++], conv2d_bf16s_kernel<1, 1, 3, 32>, 256, 0);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[n++], conv2d_bf16s_kernel<3, 2, 3, 16>, 256, 0);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  out[n++] = (int)(pr.maxSharedMemoryPerMultiProcessor / 1024); out[n++] = (int)(pr.sharedMemPerBlock / 1024);
  out[n++] = pr.regsPerMultiprocessor; out[n++] = pr.regsPerBlock;
  return n;
}
extern "C" int accflow_debug_kprof(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kprof), 4096 * 16 * 8);
  if (reset) { void* p; hipGetSymbolAddress(&p, HIP_SYMBOL(g_kprof)); hipMemset(p, 0, 4096 * 16 * 8); }
  return (int)hipGetLastError();
}
#endif
