// Device code shared by the convolution translation units (conv2d.hip: C-ABI, dispatch, packing, small-Cout kernels;
// conv2d_f32.hip: fp32-MFMA im2col kernel; conv2d_bf16s.hip: split-bf16 im2col kernel + displaced correlation store;
// conv2d_direct.hip: split-bf16 direct-A patch kernel).  Everything here is internal linkage.
#pragma once
#include "common.h"
#include <stdlib.h>

// launch entry points of the kernel families (one translation unit each, so that they compile in parallel)
int accflow_launch_conv_f32(const accflow_conv_desc& d, int wc, int wp, int tc, int tp, hipStream_t st);
int accflow_launch_conv_bf16s(const accflow_conv_desc& d, int tc, int tp, hipStream_t st);
int conv_ksplit_reduce_launch(const accflow_conv_desc& d, int Z, hipStream_t st);  // conv2d_direct.hip
int accflow_launch_corr_disp_bf16s(const accflow_conv_desc& d, hipStream_t st);
int accflow_launch_corr_disp_direct(const accflow_conv_desc& d, hipStream_t st);
int accflow_launch_conv_direct(const accflow_conv_desc& d, int tc, hipStream_t st);
bool accflow_conv_direct_eligible(const accflow_conv_desc& d);
// multi-source S16 kernel (conv2d_s16m.hip): lay < 0 = automatic wave layout
int accflow_launch_conv_s16m(const accflow_conv_desc& d, int lay, hipStream_t st);
void accflow_s16m_from_legacy(accflow_conv_desc& e);   // in0 / in1 S16 form -> src[]
// accflow_conv_stat_slots() runs the dispatcher in a dry mode (call-scoped, per host thread): a launcher that sees the
// pointer set reports how many statistic slots per plane its kernel would write (0: none) instead of launching
extern thread_local int* accflow_tls_dry_slots;
extern thread_local int* accflow_tls_dry_route;  // same protocol: 1 = direct kernel that can normalise on load
#define ACCFLOW_DRY_RUN(SLOTS)                                 \
  do {                                                         \
    if (accflow_tls_dry_slots) {                               \
      *accflow_tls_dry_slots = (SLOTS);                        \
      return 0;                                                \
    }                                                          \
  } while (0)
#ifndef ACCFLOW_EPI_GRU_AHEAD
#define ACCFLOW_EPI_GRU_AHEAD 3     // (1 = round 1-5 behaviour; 4 spills 5 registers and costs 6 % conv rate: profiles/r06_ab_gru_epilogue_prefetch.txt)
#endif
constexpr int ACCFLOW_TAPGEMM_MAXROWS = 18;   // ACCFLOW_EPI_TAPGEMM: rows of the second product (3x3 taps x 2 channels)
#ifndef ACCFLOW_CONV_XCD_ORDER
#define ACCFLOW_CONV_XCD_ORDER 1   // (0: measurement builds - the plain launch order of rounds 1-5)
#endif
#ifndef ACCFLOW_CONV_XCD_MIN_K
#define ACCFLOW_CONV_XCD_MIN_K 8192  // reduction depth from which the direct kernel reorders its launch index (conv2d_direct_kernel.h)
#endif
constexpr int DIR_TH = 4, DIR_TW = 32, DIR_NPMAX = 256;  // direct kernel: tile and max patch pixels (3x3: 204, 1x5: 144, 5x1: 256)

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

// ---- InstanceNorm statistics in the epilogue (accflow_conv_desc.stats) ----
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row, delivered to all of them (quad swaps, half-row mirror, row mirror)
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror
  return v;
}
// sum over the 32 lanes of a half-wave, delivered to all of them
__device__ __forceinline__ float half32_sum(float v) {
  v = row16_sum(v);
  return v + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));  // xor 16
}
// one partial record {sum, M2 about the partial's own mean, count}
__device__ __forceinline__ void stat_store(const accflow_conv_desc& d, int b, int ch, int slot, float s, float m2, float n) {
  float* p = d.stats + (((long long)b * d.Cout + ch) * d.stat_slots + slot) * 3;
  p[0] = s; p[1] = m2; p[2] = n;
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
// stat_b / stat_slot: batch item and statistics slot of this wave's pixels (accflow_conv_desc.stats; STORE + NONE only)
// OUT16: compile the S16 copy (accflow_conv_desc.out16) - only the direct kernel's fp16 instantiations do (the other
// kernels never see out16, and every copy of this epilogue costs compile time in each of their instantiations)
// h' = (1 - z) h + z q (update.py:51) with ONE fixed rounding sequence - an explicit fma of z q onto the rounded product (1 - z) h.
// Written as a plain expression, hipcc's default contraction (-ffp-contract=fast) chose the fused form per call site: the
// packed-operand GRU epilogue (round 6) and the general one differed in the last bit.
__device__ __forceinline__ float gru_blend(float z, float h, float q) { return __builtin_fmaf(z, q, (1.0f - z) * h); }

template <int EPI, int ACT, int WC, int WP, int TC, int TP, class PixMap, bool OUT16 = false>
__device__ __forceinline__ void conv_epilogue_impl(const accflow_conv_desc& d, f32x16 (&acc)[TC][TP], int cblk0, int wc,
                                                   int wp, int lane, int OHW, PixMap pixmap, int stat_b = 0,
                                                   int stat_slot = 0) {
  constexpr unsigned MASKED = 0xFFFFFFFFu;
  const int l31 = lane & 31, lh4 = (lane >> 5) * 4;
  const int epi = EPI >= 0 ? EPI : d.epi;  // EPI < 0: read from the descriptor (combinations the estimators do not use)
  const int half = d.Cout >> 1;
  const bool has_h = epi != ACCFLOW_EPI_STORE, has_z = epi == ACCFLOW_EPI_GRU_Q, zr = epi == ACCFLOW_EPI_GRU_ZR;
  const int nout = zr ? half : d.Cout;  // channels of d.out
  auto span = [&](long long bs, int nch) { return (int)(unsigned)((((long long)(d.B - 1)) * bs + (long long)nch * OHW) * 4); };
  // channel-block scatter (accflow_conv_desc.cb; STORE / ACCUM): byte offset of channel ch, and the range of such a tensor
  const int cb = d.cb;
  auto chbyte = [&](int ch, long long cbs) -> int {
    if (!cb) return ch * OHW * 4;
    const int blk = ch / cb;
    return (int)(unsigned)(((long long)blk * cbs + (long long)(ch - blk * cb) * OHW) * 4);
  };
  auto span_cb = [&](long long bs, int nch, long long cbs) {
    if (!cb) return span(bs, nch);
    return (int)(unsigned)((((long long)(d.B - 1)) * bs + (long long)((nch - 1) / cb) * cbs + (long long)cb * OHW) * 4);
  };
  // (a NULL fp32 destination - allowed when the S16 copy is requested - gets an empty range: its stores are dropped)
  const __amdgpu_buffer_rsrc_t r_out =
      __builtin_amdgcn_make_buffer_rsrc(d.out, 0, d.out ? span_cb(d.out_bs, nout, d.out_cbs) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_o2 =
      __builtin_amdgcn_make_buffer_rsrc(zr ? d.out2 : d.out, 0, (zr && d.out2) ? span(d.out2_bs, half) : 0, 0x00020000);
  // S16 copy (accflow_conv_desc.out16): n16 channels in O16 octets, 2 term planes of OHW 16-byte chunks per octet
  const bool has16 = OUT16 && d.out16 != nullptr;
  const int n16 = zr ? half : d.Cout, O16 = (n16 + 7) >> 3;
  const __amdgpu_buffer_rsrc_t r_o16 = __builtin_amdgcn_make_buffer_rsrc(
      d.out16 ? d.out16 : (void*)d.out, 0,
      has16 ? (cb ? (int)(unsigned)((((long long)(d.B - 1)) * d.out16_bs + (long long)((n16 - 1) / cb) * d.out16_cbs +
                                     (long long)(cb / 8) * 2 * OHW * 4) * 4)
                  : (int)(unsigned)((((long long)(d.B - 1)) * d.out16_bs + (long long)O16 * 2 * OHW * 4) * 4)) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_e0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(has_h ? d.e0 : d.out), 0, has_h ? span_cb(d.e0_bs, zr ? half : d.Cout, d.e0_cbs) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_e1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(has_z ? d.e1 : d.out), 0, has_z ? span(d.e1_bs, d.Cout) : 0, 0x00020000);
  // pre-activation addend (GRU epilogues only): indexed like out2 / e1 by the conv's own output channel
  const bool has_pre = (EPI == ACCFLOW_EPI_GRU_ZR || EPI == ACCFLOW_EPI_GRU_Q) && d.pre != nullptr;
  const __amdgpu_buffer_rsrc_t r_pre = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(has_pre ? d.pre : d.out), 0, has_pre ? span(d.pre_bs, d.Cout) : 0, 0x00020000);

  // per-lane byte offsets of (batch item, pixel, + the 4-row step of the upper half-wave) in each tensor
  unsigned vo_out[TP], vo_o2[TP], vo_e0[TP], vo_e1[TP], vo_pre[TP], vo_16[TP];
#pragma unroll
  for (int tp = 0; tp < TP; ++tp) {
    int b;
    const int rem = pixmap(wp * TP * 32 + tp * 32 + l31, b);
    const bool ok = rem >= 0;
    const long long lp = (long long)rem + (long long)lh4 * OHW;
    // this lane's 8 bytes (channels 4*(lane>>5) .. +3 of an octet) inside the pixel's 16-byte chunk
    vo_16[tp] = ok && has16 ? (unsigned)((b * d.out16_bs + (long long)rem * 4) * 4 + lh4 * 2) : MASKED;
    vo_out[tp] = ok ? (unsigned)((b * d.out_bs + lp) * 4) : MASKED;
    vo_o2[tp] = ok && zr ? (unsigned)((b * d.out2_bs + lp) * 4) : MASKED;
    vo_e0[tp] = ok && has_h ? (unsigned)((b * d.e0_bs + lp) * 4) : MASKED;
    vo_e1[tp] = ok && has_z ? (unsigned)((b * d.e1_bs + lp) * 4) : MASKED;
    vo_pre[tp] = ok && has_pre ? (unsigned)((b * d.pre_bs + lp) * 4) : MASKED;
  }
  const int rowbase = cblk0 + wc * TC * 32;  // first channel of this wave's rows (wave-uniform)
  const int OHW4 = OHW * 4;
  // bias through SCALAR loads (lgkmcnt: independent of the stores' vmcnt), requested one group ahead - waiting for
  // them inside their own group cost a full SMEM latency per group, 13 us of a 100 us workgroup lifetime
  typedef const __attribute__((address_space(4))) float* cfloat_ptr;
  const cfloat_ptr sbias = (cfloat_ptr)(unsigned long long)d.bias;
  float sb0[2], sb1[2];
  // fp16 pack: per-row accumulator multiplier (exact powers of two), fetched like the bias; 1 when absent, and
  // fmaf(acc, 1, bias) is bitwise acc + bias
  const cfloat_ptr sscale = (cfloat_ptr)(unsigned long long)d.wscale16;
  float ss0[2], ss1[2];
  // One GROUP = accumulator row r of tile tc for both pixel tiles: channel chu = rowbase + tc*32 + (r&3) + 8*(r>>2)
  // in the lower half-wave, chu + 4 in the upper one.  Operands of group g+1 are requested before the stores of g.
  // Operand prefetch distance.  Round 6: an ablation build without the GRU epilogues showed them to be a THIRD of the GRU
  // kernels' time (7.32 -> 4.88 ms per step, profiles/r06_gru_epilogue_ablation.txt): 16 groups per wave, each waiting for
  // operands (h, z, the context addend: 12 dword loads) requested only ONE group earlier - a memory round trip per group.
  // The GRU forms request them ACCFLOW_EPI_GRU_AHEAD groups ahead (the K loop's fragment registers are free by now) - which
  // turned out to be worth < 1 %: the epilogues are bound by their TRAFFIC (430 MB per GRU half-step at B = 11: the context
  // addend 129, h 86 + 43, z 43 + 43, r*h 43, the pre-split h 43), not by latency (profiles/r06_ab_gru_epilogue_prefetch.txt).
  constexpr bool GRU_EPI = EPI == ACCFLOW_EPI_GRU_ZR || EPI == ACCFLOW_EPI_GRU_Q;
  constexpr int AH = GRU_EPI ? ACCFLOW_EPI_GRU_AHEAD : 1, NS = AH + 1;
  float h[NS][TP], z[NS][TP], pa[NS][TP];
  float s16v[TP][4];
  bool bad16 = false;
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) pa[i][tp] = 0.0f;
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
            r_e0, (int)((live_ && in_) ? vo_e0[tp] : MASKED), live_ ? chbyte(che_, d.e0_cbs) : 0, 0));           \
        if (has_z) ZZ[tp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(                      \
            r_e1, (int)(in_ ? vo_e1[tp] : MASKED), chu_ * OHW4, 0));                                             \
        if (has_pre) pa[(G) % NS][tp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(          \
            r_pre, (int)(in_ ? vo_pre[tp] : MASKED), chu_ * OHW4, 0));                                           \
      }                                                                                                          \
    }                                                                                                            \
  } while (0)
#define EPI_BIAS(G, S)                                                                                           \
  do {                                                                                                           \
    const int chu_ = EPI_CHU(G);                                                                                 \
    sb0[S] = d.bias ? sbias[min(chu_, d.Cout - 1)] : 0.0f;                                                       \
    sb1[S] = d.bias ? sbias[min(chu_ + 4, d.Cout - 1)] : 0.0f;                                                   \
    ss0[S] = d.wscale16 ? sscale[min(chu_, d.Cout - 1)] : 1.0f;                                                  \
    ss1[S] = d.wscale16 ? sscale[min(chu_ + 4, d.Cout - 1)] : 1.0f;                                              \
  } while (0)
#pragma unroll
  for (int g0 = 0; g0 < AH && g0 < TC * 16; ++g0) EPI_FETCH(g0, h[g0 % NS], z[g0 % NS]);
  EPI_BIAS(0, 0);
  constexpr bool CAN_STATS = EPI == ACCFLOW_EPI_STORE && ACT == ACCFLOW_ACT_NONE;
  float stat_n = 0.0f;
  if constexpr (CAN_STATS) {
    if (d.stats) {
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) stat_n += vo_out[tp] != MASKED ? 1.0f : 0.0f;
      stat_n = half32_sum(stat_n);
    }
  }
#pragma unroll
  for (int g = 0; g < TC * 16; ++g) {
    __builtin_amdgcn_sched_barrier(0);
    const float bv = lh4 ? sb1[g & 1] : sb0[g & 1];
    const float sv = lh4 ? ss1[g & 1] : ss0[g & 1];
    __builtin_amdgcn_sched_barrier(0);
    if (g + AH < TC * 16) EPI_FETCH(g + AH, h[(g + AH) % NS], z[(g + AH) % NS]);
    if (g + 1 < TC * 16) EPI_BIAS(g + 1, (g + 1) & 1);
    const int tc = g / 16, r = g & 15;
    const int chu = EPI_CHU(g);
    const bool in = chu + lh4 < d.Cout;
    if constexpr (CAN_STATS) {
      if (d.stats) {  // {sum, M2, n} of this half-wave's pixels of channel chu + lh4 (wave-uniform branch)
        float s = 0.0f;
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) s += vo_out[tp] != MASKED ? fmaf(acc[tc][tp][r], sv, bv) : 0.0f;
        s = half32_sum(s);
        const float mean = s / fmaxf(stat_n, 1.0f);
        float q = 0.0f;
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) {
          const float dv = fmaf(acc[tc][tp][r], sv, bv) - mean;
          q += vo_out[tp] != MASKED ? dv * dv : 0.0f;
        }
        q = half32_sum(q);
        if (l31 == 0 && in) stat_store(d, stat_b, chu + lh4, stat_slot, s, q, stat_n);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) {
      const float v = apply_act(fmaf(acc[tc][tp][r], sv, bv) + pa[g % NS][tp], ACT);
#ifdef ACCFLOW_KPROF_NOSTORE
      if (v != 12345.678f) continue;
#endif
      const float hh = h[g % NS][tp], zz = z[g % NS][tp];
      float o = v;
      if (epi == ACCFLOW_EPI_RES_RELU) o = fmaxf(hh + v, 0.0f);
      else if (epi == ACCFLOW_EPI_GRU_Q) o = gru_blend(zz, hh, v);
      else if (epi == ACCFLOW_EPI_ACCUM) o = hh + v;
      if (zr && chu >= half) {  // r gate rows (Cout % 16 == 0: both half-waves on the same side): r * h into out2
        o = v * hh;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), r_o2, (int)(in ? vo_o2[tp] : MASKED),
                                              (chu - half) * OHW4, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), r_out, (int)(in ? vo_out[tp] : MASKED),
                                              chbyte(chu, d.out_cbs), 0);
      }
      s16v[tp][g & 3] = o;
    }
    // S16 copy: rows r = 4m .. 4m+3 of a tile are 4 consecutive channels of one octet (the upper half-wave holds the
    // octet's other 4): after the 4th row every lane packs its 4 values * 2^ASHIFT into fp16 hi / lo and writes 8 bytes
    // into each term's chunk - the 64 lanes of a store cover 32 whole 16-byte chunks, 512 contiguous bytes
    if constexpr (OUT16) if (has16 && (g & 3) == 3) {
      const int cs0 = (zr ? chu - half : chu) - 3;          // first channel of the 4-row group (lower half-wave)
      const int oct = cs0 >> 3;
      if (cs0 >= 0 && oct < O16) {                           // (wave-uniform)
        const int nvalid = n16 - (cs0 + lh4);                // channels of this lane's group that exist
        const bool full = (oct + 1) * 8 <= n16;              // (wave-uniform) every channel of the octet exists
        constexpr float ASC16 = (float)(1 << ACCFLOW_F16_ASHIFT);
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) {
          unsigned hi2[2], lo2[2];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const float a = (2 * k < nvalid) ? s16v[tp][2 * k] * ASC16 : 0.0f;
            const float b = (2 * k + 1 < nvalid) ? s16v[tp][2 * k + 1] * ASC16 : 0.0f;
            bad16 |= !(fabsf(a) < 65520.0f) | !(fabsf(b) < 65520.0f);
            const f32x2_ v2 = {a, b};
            const f16x2_ hq = __builtin_convertvector(v2, f16x2_);
            const f32x2_ back = __builtin_convertvector(hq, f32x2_);
            const f32x2_ rest = {a - back[0], b - back[1]};
            const f16x2_ lq = __builtin_convertvector(rest, f16x2_);
            hi2[k] = __builtin_bit_cast(unsigned, hq);
            lo2[k] = __builtin_bit_cast(unsigned, lq);
          }
          // byte offset of the octet's hi plane; lo plane + OHW * 16 (channel blocks: block, then the octet inside it)
          const int so = cb ? (int)(unsigned)(((long long)(cs0 / cb) * d.out16_cbs) * 4 + ((cs0 % cb) >> 3) * 2 * OHW * 16)
                            : oct * 2 * OHW * 16;
          if (full) {
            const u32x2_ hv = {hi2[0], hi2[1]}, lv = {lo2[0], lo2[1]};
            __builtin_amdgcn_raw_buffer_store_b64(hv, r_o16, (int)vo_16[tp], so, 0);
            __builtin_amdgcn_raw_buffer_store_b64(lv, r_o16, (int)vo_16[tp], so + OHW * 16, 0);
          } else {  // partial last octet: whole pairs only (see accflow_conv_desc.out16)
            const unsigned v4 = nvalid >= 3 ? vo_16[tp] : MASKED, v2o = (nvalid >= 1 && nvalid <= 2) ? vo_16[tp] : MASKED;
            const u32x2_ hv = {hi2[0], hi2[1]}, lv = {lo2[0], lo2[1]};
            __builtin_amdgcn_raw_buffer_store_b64(hv, r_o16, (int)v4, so, 0);
            __builtin_amdgcn_raw_buffer_store_b64(lv, r_o16, (int)v4, so + OHW * 16, 0);
            __builtin_amdgcn_raw_buffer_store_b32(hi2[0], r_o16, (int)v2o, so, 0);
            __builtin_amdgcn_raw_buffer_store_b32(lo2[0], r_o16, (int)v2o, so + OHW * 16, 0);
          }
        }
      }
    }
  }
  if constexpr (OUT16) if (has16 && bad16 && d.guard) atomicOr(d.guard, 1);
#undef EPI_BIAS
#undef EPI_FETCH
#undef EPI_CHU
}

// LEAN form of the plain-store epilogue (ACCFLOW_EPI_STORE, activation NONE / RELU; fp32 and / or S16 destination) for a
// wave whose 32 * TC channel rows all exist, without channel-block scatter and statistics - what the encoders' and the
// fusion chain's convolutions need.  In-kernel stamps on the general form above showed ~700 cycles per 4-element group
// with the stores REMOVED (profiles/r04_s16m_kprof.txt: 11 000 - 34 000 cycles per workgroup, 10 - 26 % of its lifetime):
// ~1 KB of straight-line code per group - per-element masks, the channel-block divisions (branched over, but fetched
// around), scalar bias / scale loads waited for group by group - executed once per workgroup, i.e. fetched cold.  Here
// the per-row scale and bias arrive as 8 vector loads per 32 rows issued together, and an element is fma, (max), store.
// Same arithmetic, bit for bit: act(fmaf(acc, scale, bias) + 0).
template <int ACT, int WC, int WP, int TC, int TP, class PixMap>
__device__ __forceinline__ void conv_epilogue_lean(const accflow_conv_desc& d, f32x16 (&acc)[TC][TP], int cblk0, int wc,
                                                   int wp, int lane, int OHW, PixMap pixmap) {
  static_assert(ACT == ACCFLOW_ACT_NONE || ACT == ACCFLOW_ACT_RELU, "lean epilogue: no transcendental activations");
  constexpr unsigned MASKED = 0xFFFFFFFFu;
  const int l31 = lane & 31, lh4 = (lane >> 5) * 4;
  const int rowbase = cblk0 + wc * TC * 32;
  const int OHW4 = OHW * 4;
  const bool has32 = d.out != nullptr, has16 = d.out16 != nullptr;
  const int O16 = (d.Cout + 7) >> 3;
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(
      d.out, 0, has32 ? (int)(unsigned)((((long long)(d.B - 1)) * d.out_bs + (long long)d.Cout * OHW) * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_o16 = __builtin_amdgcn_make_buffer_rsrc(
      d.out16 ? d.out16 : (void*)d.out, 0,
      has16 ? (int)(unsigned)((((long long)(d.B - 1)) * d.out16_bs + (long long)O16 * 2 * OHW * 4) * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.bias ? d.bias : d.wscale16), 0,
                                                                      d.bias ? d.Cout * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.wscale16 ? d.wscale16 : d.bias), 0,
                                                                      d.wscale16 ? d.CoutPad * 4 : 0, 0x00020000);
  unsigned vo_out[TP], vo_16[TP];
#pragma unroll
  for (int tp = 0; tp < TP; ++tp) {
    int b;
    const int rem = pixmap(wp * TP * 32 + tp * 32 + l31, b);
    const bool ok = rem >= 0;
    vo_16[tp] = ok && has16 ? (unsigned)((b * d.out16_bs + (long long)rem * 4) * 4 + lh4 * 2) : MASKED;
    // (accflow_conv_desc.p32 bit 0: the fp32 destination in the pixel-major layout - this lane's 4 channels are 16 contiguous bytes)
    vo_out[tp] = !(ok && has32) ? MASKED
                 : (d.p32 & 1) ? (unsigned)((b * d.out_bs + (long long)rem * 8 + lh4) * 4)
                               : (unsigned)((b * d.out_bs + (long long)rem + (long long)lh4 * OHW) * 4);
  }
  const bool p32o = (d.p32 & 1) != 0;
  typedef float f32x4_ __attribute__((ext_vector_type(4)));
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  f32x4_ bv[TC][4], sv[TC][4];
#pragma unroll
  for (int tc = 0; tc < TC; ++tc)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int off = (rowbase + tc * 32 + 8 * m + lh4) * 4;
      bv[tc][m] = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(r_b, off, 0, 0));
      sv[tc][m] = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(r_s, off, 0, 0));
    }
  const bool no_scale = d.wscale16 == nullptr;
  bool bad16 = false;
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  constexpr float ASC16 = (float)(1 << ACCFLOW_F16_ASHIFT);
#pragma unroll
  for (int tc = 0; tc < TC; ++tc)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float o[TP][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 4 * m + q;
        const float s = no_scale ? 1.0f : sv[tc][m][q];
        const int so = (rowbase + tc * 32 + 8 * m + q) * OHW4;
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) {
          float v = fmaf(acc[tc][tp][r], s, bv[tc][m][q]) + 0.0f;
          if (ACT == ACCFLOW_ACT_RELU) v = fmaxf(v, 0.0f);
          o[tp][q] = v;
#ifndef ACCFLOW_KPROF_NOSTORE
          if (!p32o) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r_out, (int)vo_out[tp], so, 0);
#endif
        }
      }
      if (p32o) {    // (wave-uniform) one 16-byte store per pixel: channels 4 * (lane >> 5) .. + 3 of octet (rowbase + tc*32 + 8m) / 8
        const int so32 = ((rowbase + tc * 32 + 8 * m) >> 3) * OHW * 32;
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) {
          const u32x4_ pv = {__builtin_bit_cast(unsigned, o[tp][0]), __builtin_bit_cast(unsigned, o[tp][1]),
                             __builtin_bit_cast(unsigned, o[tp][2]), __builtin_bit_cast(unsigned, o[tp][3])};
          __builtin_amdgcn_raw_buffer_store_b128(pv, r_out, (int)vo_out[tp], so32, 0);
        }
      }
      if (has16) {   // (wave-uniform) the 4 rows = channels 4 * (lane >> 5) .. + 3 of octet (rowbase + tc*32 + 8m) / 8
        const int so16 = ((rowbase + tc * 32 + 8 * m) >> 3) * 2 * OHW * 16;
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) {
          unsigned hi2[2], lo2[2];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const float a = o[tp][2 * k] * ASC16, b = o[tp][2 * k + 1] * ASC16;
            bad16 |= !(fabsf(a) < 65520.0f) | !(fabsf(b) < 65520.0f);
            const f32x2_ v2 = {a, b};
            const f16x2_ hq = __builtin_convertvector(v2, f16x2_);
            const f32x2_ back = __builtin_convertvector(hq, f32x2_);
            const f32x2_ rest = {a - back[0], b - back[1]};
            const f16x2_ lq = __builtin_convertvector(rest, f16x2_);
            hi2[k] = __builtin_bit_cast(unsigned, hq);
            lo2[k] = __builtin_bit_cast(unsigned, lq);
          }
          const u32x2_ hv = {hi2[0], hi2[1]}, lv = {lo2[0], lo2[1]};
#ifndef ACCFLOW_KPROF_NOSTORE
          __builtin_amdgcn_raw_buffer_store_b64(hv, r_o16, (int)vo_16[tp], so16, 0);
          __builtin_amdgcn_raw_buffer_store_b64(lv, r_o16, (int)vo_16[tp], so16 + OHW * 16, 0);
#endif
        }
      }
    }
  if (has16 && bad16 && d.guard) atomicOr(d.guard, 1);
}


// The GRU epilogues of the refinement loop on PACKED operands (round 6; the direct kernel's 5-tap S16 instantiations, 4 x 1
// wave layout).  An ablation build showed the general epilogue above to be a third of the GRU kernels' time
// (profiles/r06_gru_epilogue_ablation.txt): per 4-row group and pixel tile it issues 4 dword loads for each of h, z and the
// context addend and 4 dword stores, every one a 128-byte segment of another channel plane.  Here every operand a lane needs
// for its 4 rows of a group is ONE load: the state h from the pre-split tensor (accflow_conv_desc.e0_fmt: 8 bytes per term
// plane, h = (hi + lo) / 2^ASHIFT), z and the context addend from the PIXEL-MAJOR fp32 layout (accflow_conv_desc.p32:
// (B, C/8, H*W, 8) - this lane's 4 channels are 16 contiguous bytes), z is written the same way, and the q launch writes the
// new state pre-split ONLY (no fp32 copy: 43 MB per launch at B = 11).  Same arithmetic as the general form, element for
// element: act(fmaf(acc, scale, bias) + pre); z-rows store it, r-rows store (r * h) pre-split, q: (1 - z) h + z q.
// The pixel tiles are processed in two halves (registers: the operand blocks of 2 tiles, double-buffered over the groups).
template <bool ZR, int TP, class PixMap>
__device__ __forceinline__ void conv_epilogue_gru16(const accflow_conv_desc& d, f32x16 (&acc)[1][TP], int cblk0, int wc, int lane,
                                                    int OHW, PixMap pixmap) {
  static_assert(TP == 4, "the 4 x 1 wave layout: 32 channels x 128 pixels per wave");
  constexpr unsigned MASKED = 0xFFFFFFFFu;
  constexpr int TPH = 2;
  typedef float f32x4_ __attribute__((ext_vector_type(4)));
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  const int l31 = lane & 31, lh4 = (lane >> 5) * 4;
  const int rowbase = cblk0 + wc * 32;            // first output channel of this wave (a multiple of 32)
  const int half = d.Cout >> 1;
  const bool zrow = ZR && rowbase < half;         // (wave-uniform) GRU_ZR: this workgroup holds z rows (else r rows)
  const bool need_h = !zrow;                      // r rows and q read the state
  const int nst = ZR ? half : d.Cout;             // channels of the state / of out16
  const int cst = ZR ? rowbase - half : rowbase;  // this wave's first state channel (r rows / q)
  auto rng = [&](long long bs, int nch) { return (int)(unsigned)((((long long)(d.B - 1)) * bs + (long long)nch * OHW) * 4); };
  const __amdgpu_buffer_rsrc_t r_pre = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.pre), 0, rng(d.pre_bs, d.Cout), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_z = __builtin_amdgcn_make_buffer_rsrc(ZR ? (float*)d.out : const_cast<float*>(d.e1), 0,
                                                                      ZR ? rng(d.out_bs, half) : rng(d.e1_bs, d.Cout), 0x00020000);
  const int s16rng = (int)(unsigned)((((long long)(d.B - 1)) * d.e0_bs + (long long)((nst + 7) >> 3) * 2 * OHW * 4) * 4);
  const __amdgpu_buffer_rsrc_t r_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.e0), 0, s16rng, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_o16 = __builtin_amdgcn_make_buffer_rsrc(
      d.out16, 0, (int)(unsigned)((((long long)(d.B - 1)) * d.out16_bs + (long long)((nst + 7) >> 3) * 2 * OHW * 4) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.bias ? d.bias : d.wscale16), 0,
                                                                      d.bias ? d.Cout * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.wscale16), 0, d.CoutPad * 4, 0x00020000);
  constexpr float ASC16 = (float)(1 << ACCFLOW_F16_ASHIFT);
  bool bad16 = false;
#pragma unroll
  for (int hh = 0; hh < TP / TPH; ++hh) {
    unsigned vo_p[TPH], vo_z[TPH], vo_h[TPH], vo_o[TPH];
#pragma unroll
    for (int t = 0; t < TPH; ++t) {
      int b;
      const int rem = pixmap((hh * TPH + t) * 32 + l31, b);
      const bool ok = rem >= 0;
      vo_p[t] = ok ? (unsigned)((b * d.pre_bs + (long long)rem * 8 + lh4) * 4) : MASKED;
      vo_z[t] = ok ? (unsigned)((b * (ZR ? d.out_bs : d.e1_bs) + (long long)rem * 8 + lh4) * 4) : MASKED;
      vo_h[t] = ok ? (unsigned)((b * d.e0_bs + (long long)rem * 4) * 4 + lh4 * 2) : MASKED;
      vo_o[t] = ok ? (unsigned)((b * d.out16_bs + (long long)rem * 4) * 4 + lh4 * 2) : MASKED;
    }
    f32x4_ pre[2][TPH], zz[2][TPH], bv[2], sv[2];
    u32x2_ hhi[2][TPH], hlo[2][TPH];
#define GRU16_FETCH(M, S)                                                                                          \
    do {                                                                                                           \
      const int ch_ = rowbase + 8 * (M);                      /* output channel of the group's first row */       \
      bv[S] = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(r_b, (ch_ + lh4) * 4, 0, 0));       \
      sv[S] = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(r_s, (ch_ + lh4) * 4, 0, 0));       \
      const int sop_ = (ch_ >> 3) * OHW * 32;                 /* octet of the pixel-major fp32 tensors */          \
      const int so16_ = ((cst + 8 * (M)) >> 3) * 2 * OHW * 16; /* octet of the pre-split state */                  \
      _Pragma("unroll") for (int t = 0; t < TPH; ++t) {                                                            \
        pre[S][t] = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(r_pre, (int)vo_p[t], sop_, 0)); \
        if (!ZR) zz[S][t] = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(r_z, (int)vo_z[t], sop_, 0)); \
        if (need_h) {                                                                                              \
          hhi[S][t] = __builtin_bit_cast(u32x2_, __builtin_amdgcn_raw_buffer_load_b64(r_h, (int)vo_h[t], so16_, 0)); \
          hlo[S][t] = __builtin_bit_cast(u32x2_, __builtin_amdgcn_raw_buffer_load_b64(r_h, (int)vo_h[t], so16_ + OHW * 16, 0)); \
        }                                                                                                          \
      }                                                                                                            \
    } while (0)
    GRU16_FETCH(0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int S = m & 1;
      if (m + 1 < 4) GRU16_FETCH(m + 1, S ^ 1);          // (requested before this group's stores: one in-order vmcnt)
#pragma unroll
      for (int t = 0; t < TPH; ++t) {
        const int tp = hh * TPH + t;
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = apply_act(fmaf(acc[0][tp][4 * m + q], sv[S][q], bv[S][q]) + pre[S][t][q],   // (no bias: an empty range reads 0)
                                    ZR ? ACCFLOW_ACT_SIGMOID : ACCFLOW_ACT_TANH);
          if (need_h) {
            const unsigned wh = hhi[S][t][q >> 1], wl = hlo[S][t][q >> 1];
            const unsigned short hq = (unsigned short)((q & 1) ? wh >> 16 : wh & 0xFFFFu), lq = (unsigned short)((q & 1) ? wl >> 16 : wl & 0xFFFFu);
            const float hv = ((float)__builtin_bit_cast(_Float16, hq) + (float)__builtin_bit_cast(_Float16, lq)) * (1.0f / ASC16);
            o[q] = ZR ? v * hv : gru_blend(zz[S][t][q], hv, v);
          } else {
            o[q] = v;
          }
        }
        if (zrow) {   // z rows: the gate itself, pixel-major fp32 (read back by the q launch of this half-step)
          const u32x4_ pv = {__builtin_bit_cast(unsigned, o[0]), __builtin_bit_cast(unsigned, o[1]), __builtin_bit_cast(unsigned, o[2]),
                             __builtin_bit_cast(unsigned, o[3])};
          __builtin_amdgcn_raw_buffer_store_b128(pv, r_z, (int)vo_z[t], ((rowbase + 8 * m) >> 3) * OHW * 32, 0);
        } else {      // r * h (GRU_ZR) / the new state (GRU_Q): pre-split only
          unsigned hi2[2], lo2[2];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const float a = o[2 * k] * ASC16, b2 = o[2 * k + 1] * ASC16;
            bad16 |= !(fabsf(a) < 65520.0f) | !(fabsf(b2) < 65520.0f);
            const f32x2_ v2 = {a, b2};
            const f16x2_ hq2 = __builtin_convertvector(v2, f16x2_);
            const f32x2_ back = __builtin_convertvector(hq2, f32x2_);
            const f32x2_ rest = {a - back[0], b2 - back[1]};
            const f16x2_ lq2 = __builtin_convertvector(rest, f16x2_);
            hi2[k] = __builtin_bit_cast(unsigned, hq2);
            lo2[k] = __builtin_bit_cast(unsigned, lq2);
          }
          const u32x2_ hv2 = {hi2[0], hi2[1]}, lv2 = {lo2[0], lo2[1]};
          const int so16 = ((cst + 8 * m) >> 3) * 2 * OHW * 16;
          __builtin_amdgcn_raw_buffer_store_b64(hv2, r_o16, (int)vo_o[t], so16, 0);
          __builtin_amdgcn_raw_buffer_store_b64(lv2, r_o16, (int)vo_o[t], so16 + OHW * 16, 0);
        }
      }
    }
#undef GRU16_FETCH
  }
  if (bad16 && d.guard) atomicOr(d.guard, 1);
}

template <int WC, int WP, int TC, int TP, class PixMap, bool OUT16 = false>
__device__ __forceinline__ void conv_epilogue_px(const accflow_conv_desc& d, f32x16 (&acc)[TC][TP], int cblk0, int wc,
                                                 int wp, int lane, int OHW, PixMap pixmap, int stat_b = 0,
                                                 int stat_slot = 0) {
  // the (epilogue, activation) pairs the estimators use are compiled as straight-line code (update.py, extractor.py,
  // AccFlow_.py mirrors); any other pair takes the descriptor-driven copy
#define ACCFLOW_EPI_CASE(E, A)                                                                          \
  case (E) * 8 + (A): conv_epilogue_impl<E, A, WC, WP, TC, TP, PixMap, OUT16>(d, acc, cblk0, wc, wp, lane, OHW, pixmap, stat_b, stat_slot); break;
  switch (d.epi * 8 + d.act) {
    ACCFLOW_EPI_CASE(ACCFLOW_EPI_STORE, ACCFLOW_ACT_NONE)
    ACCFLOW_EPI_CASE(ACCFLOW_EPI_STORE, ACCFLOW_ACT_RELU)
    ACCFLOW_EPI_CASE(ACCFLOW_EPI_STORE, ACCFLOW_ACT_SIGMOID)
    ACCFLOW_EPI_CASE(ACCFLOW_EPI_RES_RELU, ACCFLOW_ACT_RELU)
    ACCFLOW_EPI_CASE(ACCFLOW_EPI_RES_RELU, ACCFLOW_ACT_NONE)   // relu(e0 + conv): e0 = a partial sum of the same convolution
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

// flattened (b, oy, ox) pixel tiles: local pixel j of pixel tile `ptile` is global pixel ptile*BP + j
template <int WC, int WP, int TC, int TP>
__device__ __forceinline__ void conv_epilogue(const accflow_conv_desc& d, f32x16 (&acc)[TC][TP], int cblk0, int wc,
                                              int wp, int lane, int OHW, int Ptot, int ptile) {
  constexpr int BP = WP * TP * 32;
  // statistics slots (only offered by the launcher when OHW % BP == 0: a tile never straddles two batch items)
  const int tile0 = ptile * BP;
  const int stat_b = tile0 / OHW, stat_slot = ((tile0 - stat_b * OHW) / BP) * WP + wp;
  // per-batch-item weights (GMA aggregation) carry per-item row scales: [item][CoutPad]
  accflow_conv_desc e = d;
  if (d.wsplit_bs && d.wscale16) e.wscale16 = d.wscale16 + (long long)stat_b * d.CoutPad;
  conv_epilogue_px<WC, WP, TC, TP>(e, acc, cblk0, wc, wp, lane, OHW, [&](int j, int& b) {
    const int p = tile0 + j;
    if (p >= Ptot) return -1;
    b = p / OHW;
    return p - b * OHW;
  }, stat_b, stat_slot);
}

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

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// hi + lo fp16 split of 8 floats times the power of two `s` (round to nearest each); bad |= a scaled value outside
// fp16's range (or NaN)
template <int OFF, int N>
__device__ __forceinline__ void split8_f16(const float (&x)[N], u32x4 (&out)[2], bool& bad, float s) {
  unsigned hi[4], lo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = x[OFF + 2 * j] * s, b = x[OFF + 2 * j + 1] * s;
    bad |= !(fabsf(a) < 65520.0f) | !(fabsf(b) < 65520.0f);
    const f32x2 v = {a, b};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f32x2 back = __builtin_convertvector(h, f32x2);
    const f32x2 r = {a - back[0], b - back[1]};
    const f16x2 l = __builtin_convertvector(r, f16x2);
    hi[j] = __builtin_bit_cast(unsigned, h);
    lo[j] = __builtin_bit_cast(unsigned, l);
  }
  { u32x4 v4 = {hi[0], hi[1], hi[2], hi[3]}; out[0] = v4; }
  { u32x4 v4 = {lo[0], lo[1], lo[2], lo[3]}; out[1] = v4; }
}

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
// target pixel q), layout E_0[p/128][dy][dx][p%128] of corr_disp.hip: dy = (y2 - y1) mod H8, dx = (x2 - x1) mod W8.  Elements of
// one output row lie on a DIAGONAL of the tile, so the accumulators go through LDS - T[q][p], 64 target columns at
// a time - and are read back with lane = target column, p = (q - u) mod 128 for the wave-uniform diagonal u: the 64
// lanes of a store then hold consecutive p of (normally) one (dy, dx) row, 256 contiguous bytes.  Both LDS passes
// are bank-conflict free (row pitch 132 words: 16-B writes land on 4q + c, reads on 5*lane + c).
constexpr int DISP_PITCH = 132;
constexpr int DISP_LDS_BYTES = (64 * DISP_PITCH + 128) * 4;

template <class QMap>
__device__ __forceinline__ void corr_disp_store(const accflow_conv_desc& d, f32x16 (&acc)[2][2], float* T, int* tab,
                                                int cblk0, int wc, int wp, int lane, int wave, int tid, QMap qmap) {
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
    const float osc = d.acc_scale != 0.0f ? d.acc_scale : 1.0f;  // fp16 split: undoes the operands' power-of-two scales
    const int q = qmap(h * 64 + lane);  // global target pixel of accumulator column h*64 + lane, or -1
    const bool qok = q >= 0;
    const int y2 = (qok ? q : 0) / W8, x2 = (qok ? q : 0) - y2 * W8;
    // 32-bit byte offsets into this pair's level 0 through a range-checked descriptor (0xFFFFFFFF = masked), the loop
    // unrolled so that the LDS reads of several diagonals are in flight: the first form (64-bit address arithmetic,
    // one LDS round trip per diagonal) took 17 us of a 27 us workgroup lifetime in the correlation GEMM
    // (the descriptor covers this workgroup's E_0[p/128] slab: P * 512 bytes, so there is no size limit per pair)
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
        d.out + (long long)(cblk0 >> 7) * P * 128, 0, (int)((unsigned)P * 512u), 0x00020000);
#pragma unroll 8
    for (int it = 0; it < 32; ++it) {
      const int u = wave * 32 + it;
      const int pl = (h * 64 + lane - u) & 127;
      const float v = T[lane * DISP_PITCH + pl] * osc;
      const int t = tab[pl];
      int dy = y2 - (t >> 16), dx = x2 - (t & 0xFFFF);
      dy += (dy >> 31) & H8;
      dx += (dx >> 31) & W8;
      const unsigned off = ((unsigned)(dy * W8 + dx) * 128u + (unsigned)pl) * 4u;  // E_0[p/128][dy][dx][p%128]
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, (int)((qok && t >= 0) ? off : 0xFFFFFFFFu), 0, 0);
    }
    if (h == 0) __syncthreads();
  }
}


// Level 0 AND level 1 of the displaced pyramid from one 128 x 128 accumulator tile whose 128 target columns are TWO
// image rows: column j = h*64 + c is target pixel (y2 = 2*yo + h, x2 = xc*64 + c).  Level 0 goes out exactly as in
// corr_disp_store (half h = row 2*yo + h at a time through T[c][p]); the 2x2 mean of level 1 - F.avg_pool2d(2, 2) on
// the volume's last two dims, raft/corr.py:20-22 - needs (2yo, 2k), (2yo, 2k+1), (2yo+1, 2k), (2yo+1, 2k+1) of ONE
// query pixel p: the first two are summed while half 0 sits in LDS (S0[k][p] = a + b), the other two added while half 1
// does: ((a + b) + c) + d, then * 0.25 - the order corr_disp_pool_kernel uses, so the result is bit-identical to
// pooling the stored level 0 (the accumulator scale `osc` is a power of two and commutes with the sums).  This removes
// the pooling pass that re-read the whole level 0 (2.6 GB per C3 step).
// Level-1 stores: lane -> p = ph*64 + lane, k = (c + lane/2) mod 32 for the wave-uniform (ph, c): x1 and 2k advance
// together (W8 even), so dx1 = (xc*32 + k) - (x1 >> 1) is constant along the lanes and the 64 lanes write 256
// contiguous bytes of E_1[p/128][dy1][dx1][p%128]; the LDS reads (pitch 132) are bank-conflict free.
constexpr int DISP2_LDS_BYTES = DISP_LDS_BYTES + 32 * DISP_PITCH * 4;
// cache policy of the displaced volume's stores (measurement builds: -DACCFLOW_CORR_STORE_AUX=2 = non-temporal)
#ifndef ACCFLOW_CORR_STORE_AUX
#define ACCFLOW_CORR_STORE_AUX 0
#endif

__device__ __forceinline__ void corr_disp_store2(const accflow_conv_desc& d, f32x16 (&acc)[2][2], float* T, int* tab, float* S0,
                                                 float* __restrict__ lvl1, int cblk0, int yo, int xc, int wc, int wp, int lane,
                                                 int wave, int tid) {
  const int H8 = d.OH, W8 = d.OW, P = H8 * W8, H1 = H8 >> 1, W1 = W8 >> 1;
  const int l31 = lane & 31;
  if (tid < 128) {
    const int p = cblk0 + tid;
    const int y1 = p / W8;
    tab[tid] = p < P ? (y1 << 16) | (p - y1 * W8) : -1;
  }
  const float osc = d.acc_scale != 0.0f ? d.acc_scale : 1.0f;
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
      d.out + (long long)(cblk0 >> 7) * P * 128, 0, (int)((unsigned)P * 512u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rl1 = __builtin_amdgcn_make_buffer_rsrc(
      lvl1 + (long long)(cblk0 >> 7) * H1 * W1 * 128, 0, (int)((unsigned)(H1 * W1) * 512u), 0x00020000);
  const bool pool_row = yo < H1;  // (odd H8: the last single row has no level-1 cell)
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
    const int y2 = 2 * yo + h, x2 = xc * 64 + lane;
    const bool qok = y2 < H8 && x2 < W8;
#pragma unroll 8
    for (int it = 0; it < 32; ++it) {
      const int u = wave * 32 + it;
      const int pl = (h * 64 + lane - u) & 127;
      const float v = T[lane * DISP_PITCH + pl] * osc;
      const int t = tab[pl];
      int dy = y2 - (t >> 16), dx = x2 - (t & 0xFFFF);
      dy += (dy >> 31) & H8;
      dx += (dx >> 31) & W8;
      const unsigned off = ((unsigned)(dy * W8 + dx) * 128u + (unsigned)pl) * 4u;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, (int)((qok && t >= 0) ? off : 0xFFFFFFFFu), 0, ACCFLOW_CORR_STORE_AUX);
    }
    if (pool_row) {
#pragma unroll 4
      for (int it = 0; it < 16; ++it) {
        const int n = wave * 16 + it;
        const int pl = (n & 1) * 64 + lane, k = ((n >> 1) + (lane >> 1)) & 31;
        const float a = T[(2 * k) * DISP_PITCH + pl], b = T[(2 * k + 1) * DISP_PITCH + pl];
        if (h == 0) {
          S0[k * DISP_PITCH + pl] = a + b;
        } else {
          const float v = (((S0[k * DISP_PITCH + pl] + a) + b) * osc) * 0.25f;
          const int t = tab[pl];
          const int xo = xc * 32 + k;
          int dy = yo - ((t >> 16) >> 1), dx = xo - ((t & 0xFFFF) >> 1);
          dy += (dy >> 31) & H1;
          dx += (dx >> 31) & W1;
          const unsigned off = ((unsigned)(dy * W1 + dx) * 128u + (unsigned)pl) * 4u;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rl1, (int)((xo < W1 && t >= 0) ? off : 0xFFFFFFFFu), 0, ACCFLOW_CORR_STORE_AUX);
        }
      }
    }
    if (h == 0) __syncthreads();
  }
}


#ifdef ACCFLOW_KPROF
__device__ unsigned long long g_kprof[4096 * 16];  // (one copy per translation unit; only conv2d_direct.hip reads it back)
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

}  // namespace
