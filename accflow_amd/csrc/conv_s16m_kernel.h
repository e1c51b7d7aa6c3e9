// The multi-source S16 convolution kernel (round 4) - included by the translation units that instantiate it
// (conv_s16m_v*.hip, one wave layout each, compiled in parallel).
//
// Same arithmetic, weight-pack layout, LDS patch scheme and epilogue as the direct-A patch kernel's S16 form
// (conv2d_direct_kernel.h: fp16 hi + lo split, 3 MFMAs per product, A fragments straight from L2, the input patch of a
// channel chunk DMA'd into LDS and shared by all taps) - results are bit-identical to it for the shapes both accept.
// What is new is the FRONT END:
//
//   * up to 4 SOURCES (accflow_conv_desc.src[]), each an S16 tensor with its own geometry: channel count, spatial
//     size, a pixel step + origin (patch position (Y, X) reads input pixel (step*Y + oy, step*X + ox)) and its own
//     sub-kernel (KH x KW, padH, padW).  The reduction runs source after source, 16-channel chunk after chunk, tap
//     after tap; the weight pack is ordered the same way (accflow_conv_pack_multi16), with one power-of-two row scale
//     per output channel over all sources.  This expresses, without copying a byte:
//       - torch.cat([...], dim=1) of 3 / 4 tensors feeding a conv (AccPlus, AccFlow_.py:98-107) - the two-source
//         limit of in0 / in1 made the caller materialise those cats (nine copy launches per fusion step);
//       - STRIDE-2 convolutions (extractor.py:9,52: 3x3 / 1x1, stride 2) as stride-1 work over the four pixel-parity
//         classes of the input: class (py, px) is a source with step 2, origin (py, px) and the taps of that parity
//         (1, 2, 2 and 4 of the nine) - exactly the products of the strided conv, no zero taps, the input read from
//         its ordinary S16 tensor (the DMA's per-lane source offsets do the de-interleave).  These convs ran on the
//         im2col kernel at 25-150 TFLOP/s.
//   * WAVE LAYOUTS beside the 128-channel x (4 x 32)-pixel one: 64 channels x (8 x 32) pixels (12 MFMAs per wave and
//     16-deep step instead of 6 for the encoders' 64-channel layers, whose (4 x 32)-pixel kernel ran at 240 TFLOP/s
//     where the 128-channel shapes reach 350) and 96 channels x (8 x 32) pixels (no padded quarter for 96 -> 96).
#pragma once
#include "conv2d_direct_kernel.h"

namespace {

template <int LAY> struct s16m_lay;
template <> struct s16m_lay<0> { static constexpr int WC = 4, WP = 1, TCW = 1, TP = 4, TH = 4; };  // 128 ch x 128 px
template <> struct s16m_lay<1> { static constexpr int WC = 2, WP = 2, TCW = 1, TP = 4, TH = 8; };  //  64 ch x 256 px
template <> struct s16m_lay<2> { static constexpr int WC = 2, WP = 2, TCW = 1, TP = 2, TH = 4; };  //  64 ch x 128 px
template <> struct s16m_lay<3> { static constexpr int WC = 1, WP = 4, TCW = 3, TP = 2, TH = 8; };  //  96 ch x 256 px
// (A 128 ch x 256 px layout - WC 4, TP 8: every A fragment feeds 8 pixel tiles instead of 4 - was built after ablation
// builds, tools/s16m_ablation.sh / profiles/r04_s16m_ablation.txt, showed the A stream from L2 to be the second largest
// cost of the loop after the MFMAs: -19 % kernel time without it, -3 % with half the LDS reads.  Its 128 accumulator
// registers leave 2 waves per SIMD, and it measured 3 - 6 % SLOWER than layout 0 on every update-block shape
// (profiles/r04_s16m_bench.txt keeps the column); removed again.)

constexpr int S16M_TW = 32;
// ablation switches of experiment builds (tools/s16m_ablation.sh): what bounds the loop?  The product build has them all 0.
#ifndef S16M_ABL_NOA
#define S16M_ABL_NOA 0
#endif
#ifndef S16M_ABL_NODMA
#define S16M_ABL_NODMA 0
#endif
#ifndef S16M_ABL_NOB
#define S16M_ABL_NOB 0
#endif
#ifndef S16M_ABL_NOMFMA
#define S16M_ABL_NOMFMA 0
#endif
// experiment builds only (tools/precision_probe_s16m.sh): which of the fp16 split's three products this kernel runs - bit 0 =
// w_lo * x_hi, bit 1 = w_hi * x_lo, bit 2 = w_hi * x_hi.  The product build runs all three.
#ifndef S16M_PAIRMASK
#define S16M_PAIRMASK 7
#endif
#ifndef S16M_LEAN_EPILOGUE
#define S16M_LEAN_EPILOGUE 1   // (0: always the general epilogue - A/B builds)
#endif
// 16-byte chunks of one LDS stage: [2 terms][oc octets][pitch] with oc * pitch = S16M_CAP / 2
__host__ __device__ constexpr int s16m_cap(int TH) { return TH == 4 ? 1024 : 1536; }

// per-source quantities both sides of the pipeline derive from the descriptor
struct s16m_geom {
  int PW, NP, T, KW, oc, NPS, nch, nq, n16;
};
template <int TH>
__host__ __device__ __forceinline__ s16m_geom s16m_geometry(const accflow_conv_src& S) {
  s16m_geom g;
  g.PW = S16M_TW + S.KW - 1;
  g.NP = (TH + S.KH - 1) * g.PW;
  g.T = S.KH * S.KW;
  g.KW = S.KW;
  g.nq = (g.NP + 63) >> 6;
  // 1x1 sub-kernels stage 4 octets per chunk when the patch is small enough (two 16-deep steps per barrier)
  g.oc = (g.T == 1 && g.nq * 64 * 8 <= s16m_cap(TH)) ? 4 : 2;
  g.NPS = s16m_cap(TH) / (2 * g.oc);
  g.n16 = (S.C + 15) >> 4;                        // 16-channel groups = steps per tap (the pack's step count: n16 * T)
  g.nch = (g.n16 + (g.oc >> 1) - 1) / (g.oc >> 1);  // (the last chunk of a 4-octet source may hold one group only)
  return g;
}

// Source s of the descriptor, read from the KERNARG segment with scalar loads.  Indexing d.src[] with a run-time s would make
// hipcc keep a private copy of the whole by-value descriptor in scratch memory (672 bytes per lane, every access a scratch
// load); the descriptor is the kernel's only argument, so src[] sits at offsetof(accflow_conv_desc, src) of the segment.
static_assert(sizeof(accflow_conv_src) == 64, "accflow_conv_src is read as 16 dwords");
__device__ __forceinline__ accflow_conv_src s16m_src(int s) {
  typedef const __attribute__((address_space(4))) int* kint_ptr;
  const kint_ptr k = (kint_ptr)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() +
                                __builtin_offsetof(accflow_conv_desc, src)) + s * 16;
  accflow_conv_src S;
  S.ptr = (const void*)(((unsigned long long)(unsigned)k[1] << 32) | (unsigned)k[0]);
  S.bs = (long long)(((unsigned long long)(unsigned)k[3] << 32) | (unsigned)k[2]);
  S.C = k[4]; S.Hs = k[5]; S.Ws = k[6];
  S.step = k[7]; S.oy = k[8]; S.ox = k[9];
  S.KH = k[10]; S.KW = k[11]; S.padH = k[12]; S.padW = k[13];
  S.reserved = 0;
  return S;
}

// ---- A fragments by hand-placed loads and waits ----
// The weights of step s + 1 are requested at the top of step s and needed at the top of step s + 1.  With compiler-managed
// buffer loads hipcc has to cover the (data-dependent) number of LDS-DMA pieces issued in between - gfx950 counts every
// vector-memory operation in ONE in-order vmcnt - and falls back to `s_waitcnt vmcnt(0)` right after the request in every
// second step: an exposed L2 round trip per pair of steps, in this kernel AND in conv2d_direct_kernel.h (the same loop).
// An ablation build without the A loads ran 19 % faster although their bytes are a quarter of the L2's bandwidth
// (profiles/r04_s16m_ablation.txt).  Here the loads are inline assembly the compiler does not track, and the wait sits at
// the END of the step that issued them - after its MFMAs, behind a counted vmcnt that leaves the DMA pieces of that step
// in flight: 1 - 4 % per launch (profiles/r04_s16m_bench.txt).  Requesting the fragments TWO steps ahead into a third
// register set was measured too and is 8 - 10 % SLOWER (175 vs 162 us on the 1x5 GRU conv, 155 vs 141 on 128 -> 256 3x3):
// the A stream costs issue slots, L2 bandwidth and clock (the chip holds 1.95 GHz in this loop, 2.15 without the A
// loads: profiles/r04_s16m_kprof_clock.txt), not exposed latency.
// (s16m_load_a / s16m_wait_vm / s16m_wait_vm_but live in conv2d_direct_kernel.h since round 6: the direct kernel's
// tap-specialised loop uses them too)
// Lean form of the residual epilogue out = relu(e0 + relu(fmaf(acc, scale, bias))) (ACCFLOW_EPI_RES_RELU with ACT_RELU,
// extractor.py:62-63; fp32 and / or S16 destination): conv_epilogue_lean (conv_common.h) plus the residual operand, whose
// 16 * TP dwords of a 32-row tile are requested together BEFORE that tile's stores (gfx950's single in-order vmcnt: a load
// behind a store cannot be waited for without the store's acknowledgement).  Same arithmetic as the general form, bit for bit.
// E16: the residual operand is an S16 tensor (accflow_conv_desc.e0_fmt): a lane's 4 rows of a group are 8 bytes of the
// pixel's chunk in each term plane - two 8-byte loads instead of four dwords - and e0 = (hi + lo) / 2^ACCFLOW_F16_ASHIFT.
template <int WC, int WP, int TC, int TP, bool E16, class PixMap>
__device__ __forceinline__ void conv_epilogue_lean_res(const accflow_conv_desc& d, f32x16 (&acc)[TC][TP], int cblk0, int wc,
                                                       int wp, int lane, int OHW, PixMap pixmap) {
  constexpr unsigned MASKED = 0xFFFFFFFFu;
  const int l31 = lane & 31, lh4 = (lane >> 5) * 4;
  const int rowbase = cblk0 + wc * TC * 32;
  const int OHW4 = OHW * 4;
  const bool has32 = d.out != nullptr, has16 = d.out16 != nullptr;
  const int O16 = (d.Cout + 7) >> 3;
  auto span = [&](long long bs) { return (int)(unsigned)((((long long)(d.B - 1)) * bs + (long long)d.Cout * OHW) * 4); };
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(d.out, 0, has32 ? span(d.out_bs) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_e0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.e0), 0,
      E16 ? (int)(unsigned)((((long long)(d.B - 1)) * d.e0_bs + (long long)O16 * 2 * OHW * 4) * 4) : span(d.e0_bs), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_o16 = __builtin_amdgcn_make_buffer_rsrc(
      d.out16 ? d.out16 : (void*)d.out, 0,
      has16 ? (int)(unsigned)((((long long)(d.B - 1)) * d.out16_bs + (long long)O16 * 2 * OHW * 4) * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.bias ? d.bias : d.wscale16), 0,
                                                                      d.bias ? d.Cout * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.wscale16 ? d.wscale16 : d.bias), 0,
                                                                      d.wscale16 ? d.CoutPad * 4 : 0, 0x00020000);
  unsigned vo_out[TP], vo_16[TP], vo_e0[TP];
#pragma unroll
  for (int tp = 0; tp < TP; ++tp) {
    int b;
    const int rem = pixmap(wp * TP * 32 + tp * 32 + l31, b);
    const bool ok = rem >= 0;
    vo_16[tp] = ok && has16 ? (unsigned)((b * d.out16_bs + (long long)rem * 4) * 4 + lh4 * 2) : MASKED;
    vo_out[tp] = ok && has32 ? (unsigned)((b * d.out_bs + (long long)rem + (long long)lh4 * OHW) * 4) : MASKED;
    vo_e0[tp] = !ok ? MASKED
                : E16 ? (unsigned)((b * d.e0_bs + (long long)rem * 4) * 4 + lh4 * 2)
                      : (unsigned)((b * d.e0_bs + (long long)rem + (long long)lh4 * OHW) * 4);
  }
  typedef float f32x4_ __attribute__((ext_vector_type(4)));
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  const bool no_scale = d.wscale16 == nullptr;
  bool bad16 = false;
  constexpr float ASC16 = (float)(1 << ACCFLOW_F16_ASHIFT);
  // scale / bias of all rows first (8 vector loads per 32 rows), then the residual operand one 4-row group ahead of the
  // stores (2 x 4 * TP registers: the kernel's register count must stay the main loop's)
  f32x4_ bv[TC][4], sv[TC][4];
#pragma unroll
  for (int tc = 0; tc < TC; ++tc)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int off = (rowbase + tc * 32 + 8 * m + lh4) * 4;
      bv[tc][m] = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(r_b, off, 0, 0));
      sv[tc][m] = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(r_s, off, 0, 0));
    }
  float e[2][TP][4];
  typedef _Float16 f16x4_ __attribute__((ext_vector_type(4)));
#define LEAN_FETCH(G, E)                                                                                          \
  do {                                                                                                            \
    if constexpr (E16) {                                                                                          \
      const int so16_ = ((rowbase + ((G) >> 2) * 32 + 8 * ((G) & 3)) >> 3) * 2 * OHW * 16;                        \
      _Pragma("unroll") for (int tp = 0; tp < TP; ++tp) {                                                         \
        const f16x4_ h_ = __builtin_bit_cast(f16x4_, __builtin_amdgcn_raw_buffer_load_b64(r_e0, (int)vo_e0[tp], so16_, 0)); \
        const f16x4_ l_ = __builtin_bit_cast(f16x4_, __builtin_amdgcn_raw_buffer_load_b64(r_e0, (int)vo_e0[tp], so16_ + OHW * 16, 0)); \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) E[tp][q] = ((float)h_[q] + (float)l_[q]) * (1.0f / ASC16);  \
      }                                                                                                           \
    } else {                                                                                                      \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) _Pragma("unroll") for (int tp = 0; tp < TP; ++tp)             \
          E[tp][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(                              \
              r_e0, (int)vo_e0[tp], (rowbase + ((G) >> 2) * 32 + 8 * ((G) & 3) + q) * OHW4, 0));                  \
    }                                                                                                             \
  } while (0)
  LEAN_FETCH(0, e[0]);
#pragma unroll
  for (int g = 0; g < TC * 4; ++g) {
    const int tc = g >> 2, m = g & 3;
    if (g + 1 < TC * 4) { LEAN_FETCH(g + 1, e[(g + 1) & 1]); }
    float o[TP][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 4 * m + q;
      const float sc = no_scale ? 1.0f : sv[tc][m][q];
      const int so = (rowbase + tc * 32 + 8 * m + q) * OHW4;
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) {
        const float v = fmaxf(fmaf(acc[tc][tp][r], sc, bv[tc][m][q]) + 0.0f, 0.0f);
        const float w = fmaxf(e[g & 1][tp][q] + v, 0.0f);
        o[tp][q] = w;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, w), r_out, (int)vo_out[tp], so, 0);
      }
    }
    if (has16) {
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
        __builtin_amdgcn_raw_buffer_store_b64(hv, r_o16, (int)vo_16[tp], so16, 0);
        __builtin_amdgcn_raw_buffer_store_b64(lv, r_o16, (int)vo_16[tp], so16 + OHW * 16, 0);
      }
    }
  }
#undef LEAN_FETCH
  if (has16 && bad16 && d.guard) atomicOr(d.guard, 1);
}

// KT9 (round 6): every source is a 3x3, step-1 source (the encoders' residual blocks, AccPlus's concatenation convolutions, the
// decoder heads): the K loop is the direct kernel's tap-specialised form - a chunk = 9 straight-line taps whose LDS fragment
// addresses are one base register + immediates, two chunks per trip - instead of the generic step with its run-time tap / row /
// pair / source bookkeeping.  The host launches it only when accflow_s16m_all_3x3(desc) holds.
template <int LAY, bool KT9 = false>
__global__ __launch_bounds__(256, 2) void conv_s16m_kernel(const accflow_conv_desc d) {
  using L = s16m_lay<LAY>;
  constexpr int WC = L::WC, WP = L::WP, TCW = L::TCW, TP = L::TP, TH = L::TH, TW = S16M_TW;
  static_assert(WC * WP == 4 && WP * TP == TH, "4 waves; one 32-pixel accumulator tile per tile row");
  constexpr int BC = WC * TCW * 32;
  constexpr int CAP = s16m_cap(TH);
  constexpr int NQMAX = CAP / 4 / 64;
  __shared__ u32x4 Pst[2 * CAP];

#ifdef ACCFLOW_KPROF
  const unsigned long long tL0 = __builtin_readcyclecounter();
  const unsigned long long tR0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long kp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave / WP, wp = wave % WP;
  const int l31 = lane & 31, kh = lane >> 5;
  const int cblk0 = blockIdx.y * BC;
  const int OHW = d.OH * d.OW;
  const int tilesX = (d.OW + TW - 1) / TW, tilesY = (d.OH + TH - 1) / TH;
  const int tb = blockIdx.x / (tilesX * tilesY), trem = blockIdx.x - tb * tilesX * tilesY;
  const int oy0 = (trem / tilesX) * TH, ox0 = (trem % tilesX) * TW;

  // accflow_conv_desc.split_c0: the channel blocks from split_c0 on are a second convolution over source 0 alone (a residual
  // block's 1x1 stride-2 projection riding in its strided 3x3 launch): their reduction ends behind source 0's steps
  const bool second = d.split_c0 > 0 && cblk0 >= d.split_c0;
  const int nsrc = second ? 1 : d.nsrc;
  // ---- the chunk list: sources in order, each ceil(octets / oc) chunks; split-K part z takes chunks [c_begin, c_end) ----
  int nchunk = 0, nstep = 0;       // (nstep: ALL sources' steps - the pack's term stride)
  for (int s = 0; s < d.nsrc; ++s) {
    const s16m_geom g = s16m_geometry<TH>(s16m_src(s));
    if (s < nsrc) nchunk += g.nch;
    nstep += g.n16 * g.T;
  }
  const int c_begin = (int)((long long)nchunk * blockIdx.z / gridDim.z);
  const int c_end = (int)((long long)nchunk * (blockIdx.z + 1) / gridDim.z);

  // ---- staging side ----
  unsigned pixo[NQMAX];
  __amdgpu_buffer_rsrc_t st_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.bias), 0, 0, 0x00020000);
  unsigned st_item = 0;
  int st_O = 0, st_HW = 0, st_oc = 2, st_NPS = CAP / 4, st_nq = 0, st_nch = 1;
  auto stage_geom = [&](const accflow_conv_src& S) __attribute__((always_inline)) {
    const s16m_geom g = s16m_geometry<TH>(S);
    st_O = (S.C + 7) >> 3;
    st_HW = S.Hs * S.Ws;
    st_oc = g.oc; st_NPS = g.NPS; st_nq = g.nq; st_nch = g.nch;
    st_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(S.ptr), 0, (int)(unsigned)((((long long)(d.B - 1)) * S.bs + (long long)st_O * 2 * st_HW * 4) * 4), 0x00020000);
    st_item = (unsigned)((long long)tb * S.bs * 4);
#pragma unroll
    for (int q = 0; q < NQMAX; ++q) {
      const int pp = q * 64 + lane;
      const int py = pp / g.PW, px = pp - py * g.PW;
      const int iy = S.step * (oy0 - S.padH + py) + S.oy, ix = S.step * (ox0 - S.padW + px) + S.ox;
      const bool ok = pp < g.NP && (unsigned)iy < (unsigned)S.Hs && (unsigned)ix < (unsigned)S.Ws;
      pixo[q] = ok ? (unsigned)(iy * S.Ws + ix) * 16u : 0xFFFFFFFFu;
    }
  };
  auto stage_geom_of = [&](int s) __attribute__((always_inline)) { stage_geom(s16m_src(s)); };
  // DMA of chunk `cc` of the staged source into LDS stage `stage`: rows (term, octet) dealt to the waves
  auto issue_dma = [&](int stage, int cc) __attribute__((always_inline)) {
    const int oct0 = cc * st_oc;
    int issued = 0;
    for (int r = wave; r < 2 * st_oc; r += 4) {
      const int t = r >= st_oc ? 1 : 0, o = r - t * st_oc;
      const int oct = oct0 + o;
      const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)(st_item + (unsigned)((oct * 2 + t) * st_HW) * 16u));
      const bool live = oct < st_O;
#pragma unroll
      for (int q = 0; q < NQMAX; ++q) {
        if (q >= st_nq) break;
        const unsigned voff = live ? pixo[q] : 0xFFFFFFFFu;
        const int slot = __builtin_amdgcn_readfirstlane(stage * CAP + r * st_NPS + q * 64);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(st_rsrc, (__attribute__((address_space(3))) void*)&Pst[slot], 16, (int)voff,
                                                 (int)soff, 0, 0);
        ++issued;
      }
    }
    return issued;
  };

  // ---- cursors: (cs, cc) = source / chunk being multiplied, (ss, sc) = next chunk to stage ----
  int cs = 0, cc = c_begin, step0 = 0;
  {
    // locate chunk c_begin and the global step index of its first step
    for (int s = 0; s + 1 < nsrc; ++s) {
      const s16m_geom g = s16m_geometry<TH>(s16m_src(s));
      if (s != cs || cc < g.nch) break;
      cc -= g.nch;
      step0 += g.n16 * g.T;
      ++cs;
    }
  }
  // consume-side geometry of source cs
  int PW = TW, KW = 1, T = 1, npair = 1, NPS = CAP / 4, nch = 1, oc = 2, n16 = 1;
  int pbase[TP];
  auto consume_geom = [&](const accflow_conv_src& S) __attribute__((always_inline)) {
    const s16m_geom g = s16m_geometry<TH>(S);
    PW = g.PW; KW = g.KW; T = g.T; NPS = g.NPS; nch = g.nch; oc = g.oc; n16 = g.n16;
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) pbase[tp] = kh * NPS + (wp * TP + tp) * PW + l31;
  };
  auto consume_geom_of = [&](int s) __attribute__((always_inline)) { consume_geom(s16m_src(s)); };
  consume_geom_of(cs);
  step0 += cc * (oc >> 1) * T;
  npair = min(oc >> 1, n16 - cc * (oc >> 1));   // 16-channel groups of the chunk being multiplied
  int ss = cs, sc = cc;
  stage_geom_of(ss);

  // ---- A fragments: 16 bytes per lane and (term, 32-row tile) straight from the pack ----
  const long long step_bytes = 2LL * d.CoutPad * 16, term_bytes = (long long)nstep * step_bytes;
  s16m_i32x4 wdesc;
  {
    const unsigned long long wp = (unsigned long long)d.wpatch16;
    wdesc[0] = (int)(unsigned)wp;
    wdesc[1] = (int)(unsigned)((wp >> 32) & 0xFFFFu);
    wdesc[2] = (int)(unsigned)(3 * term_bytes);
    wdesc[3] = 0x00020000;
  }
  unsigned avoff[TCW];
#pragma unroll
  for (int tc = 0; tc < TCW; ++tc) avoff[tc] = (unsigned)((kh * d.CoutPad + cblk0 + (wc * TCW + tc) * 32 + l31) * 16);
#define S16M_LOAD_A(STEP, A)                                                                                     \
  _Pragma("unroll") for (int t = 0; t < 2; ++t) _Pragma("unroll") for (int tc = 0; tc < TCW; ++tc)               \
      s16m_load_a(A[t][tc], avoff[tc], wdesc,                                                                    \
                  __builtin_amdgcn_readfirstlane((int)(unsigned)(t * term_bytes + (long long)(STEP) * step_bytes)))

  f32x16 acc[TCW][TP];
#pragma unroll
  for (int tc = 0; tc < TCW; ++tc)
#pragma unroll
    for (int tp = 0; tp < TP; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tc][tp][r] = 0.0f;

  u32x4 aA[2][TCW], aB[2][TCW];
  int gc = c_begin;            // global index of the chunk being multiplied (LDS stage = gc & 1)
  int gstep = step0;           // global step index (weight pack order)
  S16M_LOAD_A(gstep, aA);
  // stage the first chunk
  auto stage_next = [&](int stage) __attribute__((always_inline)) {
    const int n = issue_dma(stage, sc);
    if (++sc == st_nch) {
      sc = 0;
      if (++ss < nsrc) stage_geom_of(ss);
    }
    return n;
  };
  if (c_begin < c_end) stage_next(c_begin & 1);
  s16m_wait_vm<0>();
  __syncthreads();

  // global step index of chunk boundary g (the pack's step order)
  auto step_of_chunk = [&](int g) __attribute__((always_inline)) {
    int acc_steps = 0;
    for (int s = 0; s < nsrc; ++s) {
      const s16m_geom gm = s16m_geometry<TH>(s16m_src(s));
      if (g >= gm.nch) { acc_steps += gm.n16 * gm.T; g -= gm.nch; }
      else { acc_steps += min(g * (gm.oc >> 1), gm.n16) * gm.T; break; }
    }
    return acc_steps;
  };
  const int step_end = step_of_chunk(c_end);
  int pair = 0, tap = 0, ty = 0, tx = 0;
#define S16M_STEP(ACUR, ANXT)                                                                                    \
  do {                                                                                                           \
    KPROF_T(tA);                                                                                                 \
    const int pstage = gc & 1;                                                                                   \
    const bool next_chunk = gc + 1 < c_end;                                                                      \
    const bool loadn = gstep + 1 < step_end && !S16M_ABL_NOA;                                                    \
    if (loadn) { S16M_LOAD_A(gstep + 1, ANXT); }                                                                 \
    int ndma = 0;                                                                                                \
    if (tap == 0 && pair == 0 && next_chunk && !S16M_ABL_NODMA) ndma = stage_next(pstage ^ 1);                   \
    KPROF_T(tA1);                                                                                                \
    const int toff = pstage * CAP + 2 * pair * NPS + ty * PW + tx;                                               \
    bf16x8 b[2][TP];                                                                                             \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) _Pragma("unroll") for (int tp = 0; tp < TP; ++tp)              \
        b[t][tp] = S16M_ABL_NOB ? __builtin_bit_cast(bf16x8, Pst[pbase[tp] & 1023])                              \
                                : __builtin_bit_cast(bf16x8, Pst[t * oc * NPS + pbase[tp] + toff]);              \
    KPROF_T(tB);                                                                                                 \
    KPROF_WAIT();                                                                                                \
    KPROF_T(tB2);                                                                                                \
    {                                                                                                            \
      constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};                                                        \
      _Pragma("unroll") for (int pr = 0; pr < 3; ++pr) _Pragma("unroll") for (int tc = 0; tc < TCW; ++tc)        \
          _Pragma("unroll") for (int tp = 0; tp < TP; ++tp)                                                      \
            if (!((S16M_PAIRMASK >> pr) & 1)) {                                                                   \
              /* (experiment builds) the fragment was requested by inline assembly: its registers must stay reserved  \
                 until the data has landed, MFMA or not */                                                        \
              asm volatile("" :: "v"(ACUR[PA[pr]][tc]));                                                            \
            } else {                                                                                              \
              if (S16M_ABL_NOMFMA) acc[tc][tp][pr] += __builtin_bit_cast(float, ACUR[PA[pr]][tc][0] ^ __builtin_bit_cast(u32x4, b[PB[pr]][tp])[1]); \
              else acc[tc][tp] = dir_mfma<true>(__builtin_bit_cast(bf16x8, ACUR[PA[pr]][tc]), b[PB[pr]][tp], acc[tc][tp]); } \
    }                                                                                                            \
    KPROF_T(tC);                                                                                                 \
    KPROF_ACC(0, tA1 - tA); KPROF_ACC(1, tB - tA1); KPROF_ACC(2, tB2 - tB); KPROF_ACC(3, tC - tB2); KPROF_ACC(7, 1); \
    ++gstep;                                                                                                     \
    if (++tx == KW) { tx = 0; ++ty; }                                                                            \
    if (++tap == T) { tap = 0; ty = 0; tx = 0; ++pair; }                                                         \
    if (pair == npair) {                                                                                         \
      s16m_wait_vm<0>();   /* this chunk's DMA (the next chunk's patch) and the next step's weights */           \
      KPROF_T(tD);                                                                                               \
      __syncthreads();                                                                                           \
      KPROF_T(tE);                                                                                               \
      KPROF_ACC(4, tD - tC); KPROF_ACC(5, tE - tD);                                                              \
      pair = 0; ++gc;                                                                                            \
      if (++cc == nch && gc < c_end) { cc = 0; ++cs; consume_geom_of(cs); }                                      \
      npair = min(oc >> 1, n16 - cc * (oc >> 1));                                                                \
    } else if (loadn) {                                                                                          \
      s16m_wait_vm_but(ndma);   /* the next step's weights; the DMA pieces issued after them stay in flight */   \
    }                                                                                                            \
  } while (0)

#ifdef ACCFLOW_KPROF
  const unsigned long long tK0 = __builtin_readcyclecounter();
#endif
  if constexpr (KT9) {
    constexpr int PW9 = TW + 2, NPS9 = CAP / 4, NP9 = (TH + 2) * PW9, NDMA9 = (NP9 + 63) / 64;
    static_assert(NP9 <= NPS9, "a 3x3 patch row fits the 2-octet stage pitch");
    const u32x4* const pb0 = &Pst[kh * NPS9 + wp * TP * PW9 + l31];
    auto chunk9 = [&](u32x4 (&A0)[2][TCW], u32x4 (&A1)[2][TCW]) __attribute__((always_inline)) {
      const int pstage = gc & 1;
      const bool next_chunk = gc + 1 < c_end;
      const u32x4* const pb = pb0 + pstage * CAP;
      dir_static_for<9>([&](auto tap_) {
        constexpr int TAP = decltype(tap_)::value;
        u32x4 (&ACUR)[2][TCW] = (TAP & 1) ? A1 : A0;
        u32x4 (&ANXT)[2][TCW] = (TAP & 1) ? A0 : A1;
        if constexpr (TAP < 8) { S16M_LOAD_A(gstep + 1, ANXT); }
        else { if (next_chunk) { S16M_LOAD_A(gstep + 1, ANXT); } }
        if constexpr (TAP == 0) { if (next_chunk) stage_next(pstage ^ 1); }
        bf16x8 b[2][TP];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int tp = 0; tp < TP; ++tp)
            b[t][tp] = __builtin_bit_cast(bf16x8, pb[t * 2 * NPS9 + tp * PW9 + (TAP / 3) * PW9 + TAP % 3]);
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int tc = 0; tc < TCW; ++tc)
#pragma unroll
            for (int tp = 0; tp < TP; ++tp)
              acc[tc][tp] = dir_mfma<true>(__builtin_bit_cast(bf16x8, ACUR[PA[pr]][tc]), b[PB[pr]][tp], acc[tc][tp]);
        ++gstep;
#if ACCFLOW_DIRECT_KT_BUPFRONT
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * TP, 0);          // (all fragment reads of the tap first, see the direct kernel)
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * TCW * TP, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
        // the next tap's weights must have landed; behind tap 0 the DMA pieces issued after them may stay in flight
        if constexpr (TAP == 0) { if (next_chunk) s16m_wait_vm<NDMA9>(); else s16m_wait_vm<0>(); }
        else if constexpr (TAP < 8) { s16m_wait_vm<0>(); }
      });
      s16m_wait_vm<0>();   // the next chunk's patch (and the next step's weights)
      __syncthreads();
      ++gc;
    };
    while (gc < c_end) {
      chunk9(aA, aB);
      if (gc < c_end) chunk9(aB, aA);
    }
  } else {
  for (int it = gstep; it < step_end; it += 2) {
    S16M_STEP(aA, aB);
    if (it + 1 < step_end) S16M_STEP(aB, aA);
  }
  }
#ifdef ACCFLOW_KPROF
  const unsigned long long tK1 = __builtin_readcyclecounter();
#endif
#undef S16M_STEP
#undef S16M_LOAD_A

  auto pixmap = [&](int j, int& b) {
    const int oy = oy0 + j / TW, ox = ox0 + j % TW;
    b = tb;
    return (oy < d.OH && ox < d.OW) ? oy * d.OW + ox : -1;
  };
  if (gridDim.z > 1) {  // raw partial sums of this K-part; conv_ksplit_reduce_kernel applies bias / act / epilogue
    accflow_conv_desc e = d;
    e.out = d.kws + (long long)blockIdx.z * d.B * d.Cout * OHW;
    e.out_bs = (long long)d.Cout * OHW;
    e.bias = nullptr;
    e.wscale16 = nullptr;
    e.out16 = nullptr;
    e.cb = 0;
    conv_epilogue_impl<ACCFLOW_EPI_STORE, ACCFLOW_ACT_NONE, WC, WP, TCW, TP>(e, acc, cblk0, wc, wp, lane, OHW, pixmap);
    return;
  }
  // the lean plain-store form where it applies (wave-uniform): conv_common.h
  // (split_c0: the second convolution takes no activation; the host admits the form only where this lean branch or - act
  // NONE for both, the InstanceNorm encoder - the general one below applies)
  if (d.epi == ACCFLOW_EPI_STORE && !d.cb && !d.stats && cblk0 + (wc + 1) * TCW * 32 <= d.Cout &&
      (d.act == ACCFLOW_ACT_NONE || d.act == ACCFLOW_ACT_RELU) && S16M_LEAN_EPILOGUE) {
    if (d.act == ACCFLOW_ACT_RELU && !second) conv_epilogue_lean<ACCFLOW_ACT_RELU, WC, WP, TCW, TP>(d, acc, cblk0, wc, wp, lane, OHW, pixmap);
    else conv_epilogue_lean<ACCFLOW_ACT_NONE, WC, WP, TCW, TP>(d, acc, cblk0, wc, wp, lane, OHW, pixmap);
  } else if (d.epi == ACCFLOW_EPI_RES_RELU && d.act == ACCFLOW_ACT_RELU && !d.cb && cblk0 + (wc + 1) * TCW * 32 <= d.Cout &&
             S16M_LEAN_EPILOGUE) {
    if (d.e0_fmt) conv_epilogue_lean_res<WC, WP, TCW, TP, true>(d, acc, cblk0, wc, wp, lane, OHW, pixmap);
    else conv_epilogue_lean_res<WC, WP, TCW, TP, false>(d, acc, cblk0, wc, wp, lane, OHW, pixmap);
  } else {
    conv_epilogue_px<WC, WP, TCW, TP, decltype(pixmap), true>(d, acc, cblk0, wc, wp, lane, OHW, pixmap, tb, trem * WP + wp);
  }
#ifdef ACCFLOW_KPROF
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long tS = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) {
    for (int i = 0; i < 8; ++i) KP_SLOT(i) = kp[i];
    KP_SLOT(8) = tK0 - tL0;      // prologue
    KP_SLOT(9) = tK1 - tK0;      // loop
    KP_SLOT(10) = 1;
    KP_SLOT(11) = tS - tK1;      // epilogue until its stores are issued
    KP_SLOT(12) = __builtin_readcyclecounter() - tS;   // store drain
    KP_SLOT(13) = __builtin_amdgcn_s_memrealtime() - tR0;   // lifetime in 10 ns ticks: cycles / this = the clock held
    KP_SLOT(14) = tR0;
  }
#endif
}

}  // namespace

// (kt9: the tap-specialised 3x3 instantiation; the caller has checked accflow_s16m_all_3x3)
int accflow_s16m_launch_0(const accflow_conv_desc& d, dim3 grid, hipStream_t st, bool kt9);
int accflow_s16m_launch_1(const accflow_conv_desc& d, dim3 grid, hipStream_t st, bool kt9);
int accflow_s16m_launch_2(const accflow_conv_desc& d, dim3 grid, hipStream_t st, bool kt9);
int accflow_s16m_launch_3(const accflow_conv_desc& d, dim3 grid, hipStream_t st, bool kt9);
