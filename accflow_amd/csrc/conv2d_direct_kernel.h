// The direct-A patch kernel (template) - included by the translation units that instantiate it (conv2d_direct_v*.hip,
// one group of instantiations each, so that they compile in parallel: the kernel's fully unrolled loop and its epilogue
// copies make a single translation unit with all ~20 instantiations take > 8 minutes) and by conv2d_direct.hip for
// dir_mfma / the shared constants.
#pragma once
#include "conv_common.h"
#include <utility>

namespace {

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
// __launch_bounds__(256, 2) makes hipcc keep the accumulators in VGPR-form MFMAs (143-165 registers in total instead of
// ~160 + 64 accumulation registers): three workgroups per CU, +7 % on the bench.  A single-buffered A set (127
// registers, four workgroups per CU) measured the same to 2 % slower and is not kept.
// An 8 x 32-pixel tile for <= 64 output channels (12 MFMAs per step and wave instead of 6, but 3 staging items per
// thread, 8 fragment reads per step and a 340-pixel patch) measured 281 vs 244 us on 64->64 3x3 at 7 x 240 x 512: not kept.

// experiment builds only (tools/precision_probe.sh): which of the fp16 split's three products run - bit 0 = w_lo * x_hi,
// bit 1 = w_hi * x_lo, bit 2 = w_hi * x_hi.  The product build runs all three.
#ifndef ACCFLOW_DIRECT_LEAN
#define ACCFLOW_DIRECT_LEAN 1   // 1: lean epilogue for the update block's store + ReLU convs.  Round 4 (profiles/r04_ab_direct_lean.txt):
                               // +1 % single-stream conv rate, -1.7 % on the pipelined step - off; re-measured on the round-6 loops
                               // (profiles/r06_ab_direct_lean.txt): +0.8 % conv rate, 23.05 / 23.01 / 23.15 vs 23.16 / 23.27 / 23.16 ms - on
#endif
// (lean only for the SHORT reductions - convc1: 22 steps, convf1: 7, the flow head's tap GEMM: 16 - was measured too:
// convc1 0.92 -> 0.84 ms per step single-stream, the pipelined step 25.22 -> 25.46 ms; same file.  0 = off.)
#ifndef ACCFLOW_DIRECT_LEAN_MAXSTEPS
#define ACCFLOW_DIRECT_LEAN_MAXSTEPS 0
#endif
#ifndef ACCFLOW_F16_PAIRMASK
#define ACCFLOW_F16_PAIRMASK 7
#endif
// waves per SIMD the tap-specialised (KT = 5) instantiations are compiled for (measurement builds: 3 / 4)
#ifndef ACCFLOW_DIRECT_KT_BUPFRONT
#define ACCFLOW_DIRECT_KT_BUPFRONT 0
#endif
#ifndef ACCFLOW_DIRECT_KT_WAVES
#define ACCFLOW_DIRECT_KT_WAVES 3
#endif
// experiment builds (tools/ab.sh with a second library): raise the wave's issue priority around a step's MFMA burst - measured
// with 1 and 3 on the update block's kernel: no effect (profiles/r05_ab_setprio.txt)
#ifndef ACCFLOW_DIRECT_SETPRIO
#define ACCFLOW_DIRECT_SETPRIO 0
#endif
template <class F, int... I>
__device__ __forceinline__ void dir_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void dir_static_for(F&& f) { dir_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// ---- A fragments by hand-placed loads and waits (conv_s16m_kernel.h explains why; shared with the tap-specialised loop below) ----
typedef int s16m_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void s16m_load_a(u32x4& dst, unsigned voff, s16m_i32x4 desc, int soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(desc), "s"(soff) : "memory");
}
// (The wait carries no register operands on purpose: "+v" operands make every wait a new definition of the fragments, and
// hipcc then copies the registers - not yet written by the load in flight - in front of it.  Without them the fragments
// flow from the load straight to the MFMAs of the NEXT step, which sit behind this step's closing branch; sched_barrier
// keeps the machine scheduler from moving anything across the wait.)
template <int N>
__device__ __forceinline__ void s16m_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// all but the n youngest vector-memory operations of this wave (n uniform: the DMA pieces issued after the A loads)
__device__ __forceinline__ void s16m_wait_vm_but(int n) {
#define S16M_W(N) case N: s16m_wait_vm<N>(); break;
  switch (n) {
    S16M_W(1) S16M_W(2) S16M_W(3) S16M_W(4) S16M_W(5) S16M_W(6) S16M_W(7) S16M_W(8) S16M_W(9) S16M_W(10)
    S16M_W(11) S16M_W(12) S16M_W(13) S16M_W(14) S16M_W(15) S16M_W(16) S16M_W(17) S16M_W(18) S16M_W(19) S16M_W(20)
    S16M_W(21) S16M_W(22) S16M_W(23) S16M_W(24) S16M_W(25) S16M_W(26) S16M_W(27) S16M_W(28) S16M_W(29) S16M_W(30)
    default: s16m_wait_vm<0>();
  }
#undef S16M_W
}

template <bool F16>
__device__ __forceinline__ f32x16 dir_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ACCFLOW_EPI_TAPGEMM - FlowHead (update.py:12-13): delta = conv2(relu(conv1(net))) with conv2 3x3, 256 -> 2.  conv2 already ran
// as "all nine taps at once": an 18-row 1x1 product over conv1's output followed by a shifted sum (accflow_tap_sum_f32) - but
// conv1 wrote its 256 channels to HBM as an S16 tensor (86 MB per B = 11 launch) for a 28-us launch at 28 TFLOP/s to read
// them back.  Here the 18-row product is conv1's epilogue.  W4 layout: a wave holds 32 channels x 128 pixels in four
// accumulator tiles; lane (column n, half h) of a tile holds rows 8 i + 4 h + j - and a 32x32x16 MFMA wants from the same lane
// 8 consecutive k of column n: the reduction is order-free, so the tap matrix is packed with its input channels in
// accumulator order (accflow_conv_desc.tg_w16) and each tile feeds TWO 16-deep products (i = 0,1 and i = 2,3) straight
// from registers after bias + ReLU + the fp16 split an out16 store would have made.  The four waves' 18 x 128 partial
// sums meet in LDS (fixed order: the result is deterministic), and the workgroup writes its 128-channel part; the tap sum
// adds the parts.  24 MFMAs per wave on top of the main loop's 864 (3x3, 128 channels in).
template <bool F16, class PixMap>
__device__ __forceinline__ void conv_epilogue_tapgemm(const accflow_conv_desc& d, f32x16 (&acc)[1][4], int cblk0, int wave,
                                                      int lane, int tid, int OHW, PixMap pixmap, float* red) {
  typedef float f32x4_ __attribute__((ext_vector_type(4)));
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
  const int l31 = lane & 31, kh = lane >> 5, lh4 = kh * 4;
  const int rowbase = cblk0 + wave * 32;
  const int R = d.tg_rows;
  const __amdgpu_buffer_rsrc_t r_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.bias ? d.bias : d.wscale16), 0,
                                                                      d.bias ? d.Cout * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.wscale16), 0, d.CoutPad * 4, 0x00020000);
  f32x4_ bv[4], sv[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int off = (rowbase + 8 * m + lh4) * 4;
    bv[m] = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(r_b, off, 0, 0));
    sv[m] = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(r_s, off, 0, 0));
  }
  // the second product's A fragments: [term][Cout / 16 steps][octet][tg_coutpad][8], this wave's two steps
  const long long step_bytes2 = 2LL * d.tg_coutpad * 16, term_bytes2 = (long long)(d.Cout >> 4) * step_bytes2;
  const __amdgpu_buffer_rsrc_t r_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.tg_w16), 0,
                                                                       (int)(unsigned)(2 * term_bytes2), 0x00020000);
  const unsigned av2 = (unsigned)((kh * d.tg_coutpad + l31) * 16);
  bf16x8 a2[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t)
      a2[s][t] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                               r_w2, (int)av2, (int)(unsigned)(t * term_bytes2 + ((rowbase >> 4) + s) * step_bytes2), 0));
  constexpr float ASC16 = (float)(1 << ACCFLOW_F16_ASHIFT);
  bool bad16 = false;
#pragma unroll
  for (int tp = 0; tp < 4; ++tp) {
    f32x16 z2;
#pragma unroll
    for (int r = 0; r < 16; ++r) z2[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      unsigned hi[4], lo[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {     // k slots 2 e2, 2 e2 + 1 of this lane = rows 8 m + 4 h + q, q + 1
        const int m = 2 * s + (e2 >> 1), q = (2 * e2) & 3;
        float va = fmaf(acc[0][tp][4 * m + q], sv[m][q], bv[m][q]) + 0.0f;
        float vb = fmaf(acc[0][tp][4 * m + q + 1], sv[m][q + 1], bv[m][q + 1]) + 0.0f;
        va = fmaxf(va, 0.0f) * ASC16;
        vb = fmaxf(vb, 0.0f) * ASC16;
        bad16 |= !(va < 65520.0f) | !(vb < 65520.0f);
        const f32x2_ v2 = {va, vb};
        const f16x2_ hq = __builtin_convertvector(v2, f16x2_);
        const f32x2_ back = __builtin_convertvector(hq, f32x2_);
        const f32x2_ rest = {va - back[0], vb - back[1]};
        const f16x2_ lq = __builtin_convertvector(rest, f16x2_);
        hi[e2] = __builtin_bit_cast(unsigned, hq);
        lo[e2] = __builtin_bit_cast(unsigned, lq);
      }
      const u32x4 hv = {hi[0], hi[1], hi[2], hi[3]}, lv = {lo[0], lo[1], lo[2], lo[3]};
      const bf16x8 bh = __builtin_bit_cast(bf16x8, hv), bl = __builtin_bit_cast(bf16x8, lv);
      z2 = dir_mfma<F16>(a2[s][1], bh, z2);      // w_lo * x_hi, w_hi * x_lo, w_hi * x_hi: the main loop's order
      z2 = dir_mfma<F16>(a2[s][0], bl, z2);
      z2 = dir_mfma<F16>(a2[s][0], bh, z2);
    }
    // (the main loop's last step ended with a barrier: no wave reads the patch any more)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 8 * (r >> 2) + lh4 + (r & 3);
      if (row < R) red[(wave * R + row) * (DIR_TH * DIR_TW) + tp * 32 + l31] = z2[r];
    }
  }
  if (bad16 && d.guard) atomicOr(d.guard, 1);
  __syncthreads();
  const long long obase = (long long)(cblk0 >> 7) * d.tg_out_ps;
  constexpr int NPX = DIR_TH * DIR_TW;
  for (int o = tid; o < R * NPX; o += 256) {
    const int row = o / NPX, px = o - row * NPX;
    const float v = ((red[o] + red[R * NPX + o]) + red[2 * R * NPX + o]) + red[3 * R * NPX + o];
    int b;
    const int rem = pixmap(px, b);
    if (rem >= 0) d.tg_out[obase + (long long)b * d.tg_out_bs + (long long)row * OHW + rem] = v * d.tg_scale[row];
  }
}

// Wave layout.  W4 = false: the 4 waves tile the 128 (64) channels x 128 pixels as 2 x 2, each wave TC x 2 accumulator
// tiles: the two waves of a channel half load the SAME A fragments from L2.  Round-2 PMC on the 128-channel kernel (whole
// C3 step: matrix pipe 38 % busy, waves 33 % parked at s_waitcnt, 2.34 GHz) and arithmetic on its operand traffic - per
// 16-deep step every wave pulls 4 KB of A fragments, 48 KB per CU and step round with 3 workgroups per CU, ~25 TB/s
// chip-wide at full matrix-pipe rate against the ~17-19 TB/s the L2s deliver - say the A stream caps the pipe near 70 %.
// W4 = true (128-channel kernel): the waves split the CHANNELS four ways (32 each) and every wave covers all 128 pixels
// (1 x 4 accumulator tiles, same 64 accumulator registers): no A fragment is loaded twice inside a workgroup, which
// halves the L2 traffic per MFMA; the B fragments (LDS, 4 instead of 2 reads per step and term) take up the slack of the
// LDS array, which ran at ~17 % of its bandwidth.
// NORM: in0 is the RAW output of a convolution whose InstanceNorm statistics are known (accflow_conv_desc.in_norm =
// {mean, 1/sqrt(var + eps)} per (batch item, channel)): the patch loader applies relu((x - mean) * rstd) on the way into
// LDS (zero padding stays zero), i.e. extractor.py:56-57 `relu(norm1(conv1(x)))` is never materialised.  The C0 <= 256
// pairs of this workgroup's batch item sit in LDS (2 KB).
// S16: the sources are pre-split "S16" tensors (accflow_conv_desc.in_fmt): [octet][term][H][W] planes of 16-byte chunks
// that ARE the LDS patch image's rows.  A wave stages its share of a chunk with four `buffer_load_dwordx4 ... lds` DMA
// pieces (64 consecutive patch pixels of one (term, octet) row each; per-lane source offset = the pixel inside the
// plane, 0xFFFFFFFF in the zero padding, which the DMA writes as zeros; the plane rides in the scalar offset): no
// gather into registers, no conversion, no LDS store instruction, 16 registers fewer.  Round 3's precision probe
// (profiles/r03_precision_probe.txt) showed the kernel bound by exactly that staging work, not by the matrix pipe.
constexpr int DIR_NORM_MAXC = 256;
// PM: which of the fp16 split's three products run (bits as ACCFLOW_F16_PAIRMASK) - 7 in every product instantiation.  Round 5
// instantiated PM = 6 / 5 / 4 for the GMA aggregation GEMM alone (its activation operand is the attention matrix: softmax
// probabilities averaged over 14 400 targets): C5 EPE 6.1e-5 -> 4.8e-4 / 5.4e-4 / 7.0e-4 px for -2.7 ms of 80 - inside the 1e-3
// gate on these weights with a factor 2, not taken (profiles/r05_agg_precision_probe.txt).
// KT > 0 (S16 instantiations, conv2d_direct_v_s16k.hip): the taps of a chunk are a COMPILE-TIME count (5: the GRU's 1x5 / 5x1) and
// the K loop is straight-line code per chunk - see the loop below.
// KWC (with KT): the kernel WIDTH as a compile-time constant too (KH = KT / KWC), round 6: every LDS fragment address of the
// straight-line loop is then ONE base register per stage + an immediate offset.  With run-time tap offsets hipcc hoisted one
// address register per (pixel tile, tap, stage) out of the loop - 40 VGPRs in the 5-tap kernel (161 in all).
template <int TC, int NT, bool F16 = false, bool W4 = false, bool NORM = false, bool S16 = false, int PM = 7, bool TG = false, int KT = 0,
          int KWC = 0>
__global__ __launch_bounds__(256, (KT == 5 ? ACCFLOW_DIRECT_KT_WAVES : 2)) void conv2d_direct_bf16s_kernel(const accflow_conv_desc d) {
  static_assert(!F16 || NT == 2, "the fp16 split has two terms");
  static_assert(KT == 0 || ((S16 || (F16 && NORM)) && (KT & 1)), "the tap-specialised loop: S16 sources or the fp16 normalise-on-load gather, an odd tap count");
  static_assert(!TG || (W4 && S16 && PM == 7), "ACCFLOW_EPI_TAPGEMM: the 4 x 1 wave layout over S16 sources");
  static_assert(!W4 || TC == 2, "the 4 x 1 wave layout is the 128-channel kernel's");
  static_assert(!S16 || (F16 && !NORM), "S16 sources hold the fp16 split");
#ifdef ACCFLOW_KPROF
  const unsigned long long tL0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long kp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  constexpr int WC = W4 ? 4 : 2, WP = W4 ? 1 : 2, TP = W4 ? 4 : 2, OCT = 2;
  constexpr int TCW = W4 ? 1 : TC;              // accumulator tiles per wave along the channels
  constexpr int BC = WC * TCW * 32;
  static_assert(BC == 2 * TC * 32 && DIR_TH * DIR_TW == WP * TP * 32, "4 x 32 pixel tile = 128 accumulator columns");
  constexpr int PSTAGE = NT * OCT * DIR_NPMAX;
  // (TG: after the loop the same memory holds the four waves' partial tap products, [4][rows <= 18][128 pixels] floats)
  constexpr int TGWORDS = TG ? 4 * ACCFLOW_TAPGEMM_MAXROWS * DIR_TH * DIR_TW / 4 : 0;
  __shared__ u32x4 Pst[2 * PSTAGE > TGWORDS ? 2 * PSTAGE : TGWORDS];             // [2][NT][OCT][DIR_NPMAX]
  __shared__ float Nrm[NORM ? 2 * DIR_NORM_MAXC : 2];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave / WP, wp = wave % WP;
  const int l31 = lane & 31, kh = lane >> 5;
  // XCD-aware workgroup order for DEEP reductions (round 6; the GMA aggregation GEMM: K = 14 400, an 829-MB activation operand
  // that streams from HBM).  Workgroups are handed to the 8 XCDs round-robin in launch order - x fastest - so the channel blocks
  // (y) of ONE pixel tile are gridDim.x launches apart: on different XCDs unless gridDim.x % 8 == 0, i.e. the same activation
  // chunk is pulled into two L2s.  Re-read the launch index so that the channel blocks of a pixel tile are 8 launches apart:
  // same XCD, dispatched together - the second one finds the chunk in that XCD's L2: configs[4] 72.9 -> 71.75 ms pipelined,
  // 74.7 -> 73.3 one at a time.  NOT for the ordinary convolutions (K <= 2 304): there the weight fragments are the scarce
  // stream, and interleaving two channel blocks on a CU halves their L1 reuse - the C3 step lost 0.1 ms, the conv family 1.4 %
  // (profiles/r06_ab_xcd_order.txt).
  int bx = blockIdx.x, by = blockIdx.y;
#if ACCFLOW_CONV_XCD_ORDER
  if (gridDim.y > 1 && (d.C0 + d.C1) * d.KH * d.KW >= ACCFLOW_CONV_XCD_MIN_K) {
    const int X = gridDim.x, Y = gridDim.y, L = bx + X * by, full = X & ~7;
    if (L < full * Y) {
      const int G = L / (8 * Y), r = L - G * 8 * Y;
      by = r >> 3;
      bx = G * 8 + (r & 7);
    } else {                                  // (the last, partial group of pixel tiles: plain order)
      const int r = L - full * Y, rem = X - full;
      by = r / rem;
      bx = full + r - by * rem;
    }
  }
#endif
  const int cblk0 = by * BC;
  const int OHW = d.OH * d.OW;
  const int tilesX = (d.OW + DIR_TW - 1) / DIR_TW, tilesY = (d.OH + DIR_TH - 1) / DIR_TH;
  const int tb = bx / (tilesX * tilesY), trem = bx - tb * tilesX * tilesY;
  const int oy0 = (trem / tilesX) * DIR_TH, ox0 = (trem % tilesX) * DIR_TW;
#ifdef ACCFLOW_DIRECT_STAGGER
  // (measurement builds) de-synchronise the workgroups of a GRU launch: every second one starts ACCFLOW_DIRECT_STAGGER x 4 us
  // late, so that the epilogue (traffic-bound) of one falls under the K loop (matrix-bound) of its neighbours on the CU
  if (KT == 5 && (blockIdx.x & 1)) {
    for (int i = 0; i < ACCFLOW_DIRECT_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  const int T = d.KH * d.KW;
  const int PW = DIR_TW + d.KW - 1, NP = (DIR_TH + d.KH - 1) * PW;
  const int Cin = d.C0 + d.C1;
  // 1x1 convolutions stage 32 channels (4 octets x 128 pixels: the same LDS footprint) per chunk and run the two
  // 16-deep steps of a chunk as "taps" 0 / 1, so that they too synchronise once per two steps
  const bool wide = T == 1;
  const int CCH = wide ? 32 : 16;                  // channels per chunk
  const int NOCT = wide ? 4 : 2;                   // octets per chunk
  const int NPS = wide ? DIR_NPMAX / 2 : DIR_NPMAX;  // patch-pixel pitch of one octet row
  const int TPC = wide ? 2 : T;                    // steps per chunk
  const int nstep = (Cin + 15) / 16 * T, nchunk = (Cin + CCH - 1) / CCH;
  const int HW = d.H * d.W;
  // split-K: workgroup z of gridDim.z accumulates chunks [c_begin, c_end) and stores raw partial sums into d.kws
  const int c_begin = (int)((long long)nchunk * blockIdx.z / gridDim.z);
  const int c_end = (int)((long long)nchunk * (blockIdx.z + 1) / gridDim.z);
  const int step_end = min(c_end * TPC, nstep);

  // ---- patch staging: item it = tid + 256*i -> (octet = it / NP, patch pixel = it % NP) ----
  unsigned voff0[2], voff1[2];   // byte offset of (b, iy, ix) in source 0 / 1, 0xFFFFFFFF in the zero padding
  int p_oct[2], p_pix[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int it = tid + 256 * i;
    const bool live = it < NOCT * NP;
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
  // ---- S16 sources: DMA pieces ----
  // LDS row (term t, octet o of the chunk) = NPS chunks; a wave issues 4 pieces per chunk: 3x3-type chunks (2 octets,
  // 256-pixel rows): wave w owns row w = (t, o) = (w >> 1, w & 1), pieces q = 0..3; 1x1 chunks (4 octets, 128-pixel
  // rows): rows 2w and 2w + 1, pieces q = 0, 1 each.
  unsigned pixo[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  const int O0 = (d.C0 + 7) >> 3, O1 = (d.C1 + 7) >> 3;
  const __amdgpu_buffer_rsrc_t rs16_0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.in0), 0, (int)(unsigned)((((long long)(d.B - 1)) * d.in0_bs + (long long)O0 * 2 * HW * 4) * 4),
      0x00020000);
  const __amdgpu_buffer_rsrc_t rs16_1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.in1 ? d.in1 : d.in0), 0,
      (int)(unsigned)(d.in1 ? (((long long)(d.B - 1)) * d.in1_bs + (long long)O1 * 2 * HW * 4) * 4 : 0), 0x00020000);
  if constexpr (S16) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pp = q * 64 + lane;
      const int py = pp / PW, px = pp - py * PW;
      const int iy = oy0 - d.padH + py, ix = ox0 - d.padW + px;
      const bool ok = pp < NP && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
      pixo[q] = ok ? (unsigned)(iy * d.W + ix) * 16u : 0xFFFFFFFFu;
    }
  }
  auto issue_dma = [&](int stage, int cc) {
    const int c0 = cc * CCH;
    const bool second = c0 >= d.C0;
    const int cs = second ? c0 - d.C0 : c0, osrc = second ? O1 : O0;
    const long long bs = second ? d.in1_bs : d.in0_bs;
    const unsigned item = (unsigned)((long long)tb * bs * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wide ? 2 * wave + (i >> 1) : wave;
      const int q = wide ? (i & 1) : i;
      if (!wide && q * 64 >= NP) continue;                   // (uniform) this piece lies beyond the patch
      const int t = wide ? row >> 2 : row >> 1, o = wide ? row & 3 : row & 1;
      const int oct = (cs >> 3) + o;
      const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)(item + (unsigned)((oct * 2 + t) * HW) * 16u));
      const unsigned voff = oct < osrc ? (q == 0 ? pixo[0] : q == 1 ? pixo[1] : q == 2 ? pixo[2] : pixo[3]) : 0xFFFFFFFFu;
      const int slot = __builtin_amdgcn_readfirstlane(stage * PSTAGE + t * (OCT * DIR_NPMAX) + o * NPS + q * 64);
      if (second)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs16_1, (__attribute__((address_space(3))) void*)&Pst[slot], 16, (int)voff,
                                                 (int)soff, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs16_0, (__attribute__((address_space(3))) void*)&Pst[slot], 16, (int)voff,
                                                 (int)soff, 0, 0);
    }
  };
  float xa[8], xb[8];
  // Loop-invariant part of a staged element's address: pixel + the octet's 8 * p_oct channels (0xFFFFFFFF in the zero
  // padding: out of the descriptor's range whatever is added).  Per chunk only a SCALAR channel offset remains, which
  // rides in the load's soffset operand: no vector address arithmetic per element (it was half of the staging VALU
  // work: 75 of ~150 instructions per thread and chunk).
  unsigned vq0[2], vq1[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned po = (unsigned)p_oct[i] * 8u * (unsigned)HW * 4u;
    vq0[i] = voff0[i] != 0xFFFFFFFFu ? voff0[i] + po : 0xFFFFFFFFu;
    vq1[i] = voff1[i] != 0xFFFFFFFFu ? voff1[i] + po : 0xFFFFFFFFu;
  }
  auto gather_patch = [&](int cc) {
    const int c0 = cc * CCH;  // first channel of the chunk (cat index); a chunk never straddles the two sources
    const bool second = c0 >= d.C0;
    const __amdgpu_buffer_rsrc_t rs = second ? rsrc1 : rsrc0;
    const int cs = second ? c0 - d.C0 : c0, cmax = second ? d.C1 : d.C0;
    if (cs + CCH <= cmax) {  // (workgroup-uniform) every channel of the chunk exists
      const unsigned va = second ? vq1[0] : vq0[0], vb = second ? vq1[1] : vq0[1];
      const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(cs) * (unsigned)HW * 4u;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int sq = (int)(so + (unsigned)q * (unsigned)HW * 4u);
        xa[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)va, sq, 0));
        xb[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)vb, sq, 0));
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {  // the last chunk of a source whose channel count is not a multiple of the chunk
      const int ca = cs + p_oct[0] * 8 + q, cb = cs + p_oct[1] * 8 + q;
      const unsigned va = second ? voff1[0] : voff0[0], vb = second ? voff1[1] : voff0[1];
      const unsigned oa = (ca < cmax && va != 0xFFFFFFFFu) ? va + (unsigned)ca * (unsigned)HW * 4u : 0xFFFFFFFFu;
      const unsigned ob = (cb < cmax && vb != 0xFFFFFFFFu) ? vb + (unsigned)cb * (unsigned)HW * 4u : 0xFFFFFFFFu;
      xa[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)oa, 0, 0));
      xb[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)ob, 0, 0));
    }
  };
  bool bad = false;  // F16: an activation outside the scaled fp16 range was seen
  constexpr float ASC = (float)(1 << ACCFLOW_F16_ASHIFT);
  if constexpr (NORM) {
    for (int i = tid; i < 2 * d.C0; i += 256) Nrm[i] = d.in_norm[(long long)tb * 2 * d.C0 + i];
    __syncthreads();
  }
  auto store_patch = [&](int stage, int cc) {
    if constexpr (NORM) {  // relu(norm(x)) for the pixels inside the image; padding and missing channels stay 0
      const int ca = cc * CCH + p_oct[0] * 8, cb = cc * CCH + p_oct[1] * 8;
      const bool ina = voff0[0] != 0xFFFFFFFFu, inb = voff0[1] != 0xFFFFFFFFu;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const bool oka = ina && ca + q < d.C0, okb = inb && cb + q < d.C0;
        const float ma = Nrm[2 * min(ca + q, d.C0 - 1)], ra = Nrm[2 * min(ca + q, d.C0 - 1) + 1];
        const float mb = Nrm[2 * min(cb + q, d.C0 - 1)], rb = Nrm[2 * min(cb + q, d.C0 - 1) + 1];
        xa[q] = oka ? fmaxf((xa[q] - ma) * ra, 0.0f) : 0.0f;
        xb[q] = okb ? fmaxf((xb[q] - mb) * rb, 0.0f) : 0.0f;
      }
    }
    u32x4 terms[NT];
    if constexpr (F16) split8_f16<0>(xa, terms, bad, ASC);
    else split8_bf16<NT, 0>(xa, terms);
    if (p_pix[0] >= 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) Pst[stage * PSTAGE + t * (OCT * DIR_NPMAX) + p_oct[0] * NPS + p_pix[0]] = terms[t];
    }
    if constexpr (F16) split8_f16<0>(xb, terms, bad, ASC);
    else split8_bf16<NT, 0>(xb, terms);
    if (p_pix[1] >= 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) Pst[stage * PSTAGE + t * (OCT * DIR_NPMAX) + p_oct[1] * NPS + p_pix[1]] = terms[t];
    }
  };

  // ---- A fragments: 16 bytes per lane and (term, 32-row tile) straight from the pack ----
  const long long step_bytes = 2LL * d.CoutPad * 16, term_bytes = (long long)nstep * step_bytes;
  const __amdgpu_buffer_rsrc_t rsrcw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(F16 ? d.wpatch16 : d.wpatch), 0,
                                                                        (int)(unsigned)(3 * term_bytes), 0x00020000);
  const unsigned avoff = (unsigned)((kh * d.CoutPad + cblk0 + wc * TCW * 32 + l31) * 16);
#define DIR_LOAD_A(STEP, A)                                                                                      \
  _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int tc = 0; tc < TCW; ++tc)              \
      A[t][tc] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(                              \
          rsrcw, (int)(avoff + tc * 512), (int)(unsigned)(t * term_bytes + (STEP) * step_bytes), 0))

  // this lane's two accumulator-column pixels inside the patch (tap (0,0)): column j -> row j / 32, col j % 32
  int pbase[TP];
#pragma unroll
  for (int tp = 0; tp < TP; ++tp) pbase[tp] = (wp * TP + tp) * PW + l31;

  f32x16 acc[TCW][TP];
#pragma unroll
  for (int tc = 0; tc < TCW; ++tc)
#pragma unroll
    for (int tp = 0; tp < TP; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tc][tp][r] = 0.0f;

  bf16x8 aA[NT][TCW], aB[NT][TCW];
  DIR_LOAD_A(c_begin * TPC, aA);
  if constexpr (S16) {
    issue_dma(c_begin & 1, c_begin);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    gather_patch(c_begin);
    store_patch(c_begin & 1, c_begin);
  }
  __syncthreads();

  int cc = c_begin, tap = 0, ty = 0, tx = 0;
  // one (chunk, tap) step: prefetch the next step's A, the next chunk's patch at tap 0, B fragments from the patch
  // at this tap's offset, MFMAs; at the chunk's last tap split / store the prefetched patch and synchronise.
#define DIR_STEP(STEP, ACUR, ANXT)                                                                               \
  do {                                                                                                           \
    KPROF_T(tA);                                                                                                 \
    const int pstage = cc & 1;                                                                                   \
    const bool next_chunk = cc + 1 < c_end;                                                                      \
    if ((STEP) + 1 < step_end) { DIR_LOAD_A((STEP) + 1, ANXT); }                                                 \
    if (tap == 0 && next_chunk) {                                                                                \
      if constexpr (S16) issue_dma(pstage ^ 1, cc + 1); else gather_patch(cc + 1);                               \
    }                                                                                                            \
    KPROF_T(tA1);                                                                                                \
    const int toff = wide ? tap * 2 * NPS : ty * PW + tx;                                                        \
    bf16x8 b[NT][TP];                                                                                            \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int tp = 0; tp < TP; ++tp)             \
        b[t][tp] = __builtin_bit_cast(bf16x8, Pst[pstage * PSTAGE + t * (OCT * DIR_NPMAX) + kh * NPS + pbase[tp] + toff]); \
    KPROF_T(tB);                                                                                                 \
    KPROF_WAIT();                                                                                                \
    KPROF_T(tB2);                                                                                                \
    {                                                                                                            \
      if (ACCFLOW_DIRECT_SETPRIO) __builtin_amdgcn_s_setprio(ACCFLOW_DIRECT_SETPRIO);                            \
      constexpr int NPAIR = NT == 3 ? 6 : 3;                                                                     \
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};                                      \
      _Pragma("unroll") for (int pr = 6 - NPAIR; pr < 6; ++pr) _Pragma("unroll") for (int tc = 0; tc < TCW; ++tc) \
          _Pragma("unroll") for (int tp = 0; tp < TP; ++tp) if (!F16 || (((ACCFLOW_F16_PAIRMASK & PM) >> (pr - 3)) & 1))  \
              acc[tc][tp] = dir_mfma<F16>(ACUR[PA[pr]][tc], b[PB[pr]][tp], acc[tc][tp]);                         \
      if (ACCFLOW_DIRECT_SETPRIO) __builtin_amdgcn_s_setprio(0);                                                 \
    }                                                                                                            \
    KPROF_T(tC);                                                                                                 \
    if (++tx == d.KW) { tx = 0; ++ty; }                                                                          \
    if (++tap == TPC || (STEP) + 1 == step_end) {                                                                \
      if constexpr (S16) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }                                    \
      else { if (next_chunk) store_patch(pstage ^ 1, cc + 1); }                                                  \
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
  if constexpr (KT > 0) {
    // Tap-specialised K loop (launch_conv_direct picks it when KH * KW == KT and the chunk is the 16-channel one).  The generic
    // loop above keeps tap / row / column / chunk counters and tests them in every step: ~165 scalar instructions and 14 branches
    // per 16-deep step next to 12 MFMAs (ISA count of the 3x3 S16 kernel).  Here a chunk is KT steps of straight-line code: the
    // tap offsets inside the patch are loop-invariant scalars, the prefetch of the next step's weights is unconditional inside a
    // chunk, the patch DMA of the next chunk is issued at tap 0, and the only branches left are per chunk: 48 instructions per
    // step instead of ~260.  KT is odd, so the two register sets of the weight fragments swap roles from chunk to chunk: two
    // chunks per trip.  What it buys is small (0.15-0.2 ms of a 24-ms step with the GRU's 5-tap convolutions on it): the loop is
    // not bound by instruction issue but by the weight-fragment stream from L2 - a build whose fragment loads all hit ONE line (a
    // bug on the way here) ran the step 1.2 ms faster - and the compiler still waits vmcnt(0) in front of every tap's LDS reads
    // while a patch DMA is outstanding.
    static_assert(KWC > 0 && KT % KWC == 0, "the tap-specialised loop knows the kernel's width");
    constexpr int PWC_ = DIR_TW + KWC - 1;
    // Round 6: the weight fragments of the NEXT tap are requested by inline assembly the compiler does not track, and waited
    // for at the END of the tap that issued them with a counted vmcnt that leaves the patch DMA pieces issued behind them
    // (tap 0) in flight - compiler-managed, the first MFMA of tap 1 sat behind `s_waitcnt vmcnt(0)`: the DMA of the next
    // chunk's patch had to land within tap 0 (conv_s16m_kernel.h's loop has worked this way since round 4).
    s16m_i32x4 wdesc;
    {
      const unsigned long long wp_ = (unsigned long long)(F16 ? d.wpatch16 : d.wpatch);
      wdesc[0] = (int)(unsigned)wp_;
      wdesc[1] = (int)(unsigned)((wp_ >> 32) & 0xFFFFu);
      wdesc[2] = (int)(unsigned)(3 * term_bytes);
      wdesc[3] = 0x00020000;
    }
    // (DMA pieces a wave issues per chunk, issue_dma: those with q * 64 < NP - a compile-time count here)
    constexpr int NPC = (DIR_TH + KT / KWC - 1) * PWC_;
    constexpr int NDMA = !S16 ? 16 : (NPC + 63) / 64 < 4 ? (NPC + 63) / 64 : 4;    // (gather form: gather_patch's 16 loads)
    auto load_a_asm = [&](int step, bf16x8 (&A)[NT][TCW]) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int tc = 0; tc < TCW; ++tc)
        {
          u32x4 frag;
          s16m_load_a(frag, avoff + tc * 512, wdesc,
                      __builtin_amdgcn_readfirstlane((int)(unsigned)(t * term_bytes + (long long)step * step_bytes)));
          A[t][tc] = __builtin_bit_cast(bf16x8, frag);
        }
    };
    constexpr int PWC = DIR_TW + KWC - 1;       // (= PW: the host launches this instantiation for KW == KWC only)
    const u32x4* const pb0 = &Pst[kh * DIR_NPMAX + wp * TP * PWC + l31];
    auto chunk = [&](int cq, bf16x8 (&A0)[NT][TCW], bf16x8 (&A1)[NT][TCW]) __attribute__((always_inline)) {
      const int pstage = cq & 1;
      const bool next_chunk = cq + 1 < c_end;
      const int sbase = cq * KT;
      const u32x4* const pb = pb0 + pstage * PSTAGE;
      dir_static_for<KT>([&](auto tap_) {
        constexpr int TAP = decltype(tap_)::value;     // (not `t`: DIR_LOAD_A's term loop uses that name)
        bf16x8 (&ACUR)[NT][TCW] = (TAP & 1) ? A1 : A0;
        bf16x8 (&ANXT)[NT][TCW] = (TAP & 1) ? A0 : A1;
        const int snext = sbase + TAP + 1;
        if constexpr (TAP + 1 < KT) { load_a_asm(snext, ANXT); }
        else { if (next_chunk) { load_a_asm(snext, ANXT); } }
        if constexpr (TAP == 0) {
          if (next_chunk) {
            if constexpr (S16) issue_dma(pstage ^ 1, cq + 1);
            else gather_patch(cq + 1);       // (16 compiler-tracked dword loads; split + stored behind the chunk's last tap)
          }
        }
        bf16x8 b[NT][TP];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
          for (int tp = 0; tp < TP; ++tp)
            b[tt][tp] = __builtin_bit_cast(bf16x8, pb[tt * (OCT * DIR_NPMAX) + tp * PWC + (TAP / KWC) * PWC + TAP % KWC]);
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int tc = 0; tc < TCW; ++tc)
#pragma unroll
            for (int tp = 0; tp < TP; ++tp) acc[tc][tp] = dir_mfma<F16>(ACUR[PA[pr]][tc], b[PB[pr]][tp], acc[tc][tp]);
#if ACCFLOW_DIRECT_KT_BUPFRONT
        // (measurement builds) all of a tap's fragment reads first, then its MFMAs: with the immediate-offset addresses the
        // machine scheduler re-uses the spent weight-fragment registers for the lo fragments and issues their reads in pairs
        // between the MFMAs, each behind an immediate lgkmcnt wait
        __builtin_amdgcn_sched_group_barrier(0x100, NT * TP, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * TCW * TP, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);     // (taps stay in order: hoisting later taps' fragment reads costs registers)
        // the next tap's weights must have landed; behind tap 0 the DMA pieces issued after them may stay in flight
        if constexpr (TAP == 0 && KT > 1) { if (next_chunk) s16m_wait_vm<NDMA>(); else s16m_wait_vm<0>(); }
        else if constexpr (TAP + 1 < KT) { s16m_wait_vm<0>(); }
      });
      s16m_wait_vm<0>();   // the next chunk's patch (and the next step's weights)
      if constexpr (!S16) { if (next_chunk) store_patch(pstage ^ 1, cq + 1); }
      __syncthreads();
    };
    for (int cq = c_begin; cq < c_end; cq += 2) {
      chunk(cq, aA, aB);
      if (cq + 1 < c_end) chunk(cq + 1, aB, aA);
    }
  } else {
  for (int step = c_begin * TPC; step < step_end; step += 2) {
    DIR_STEP(step, aA, aB);
    if (step + 1 < step_end) DIR_STEP(step + 1, aB, aA);
  }
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
  if constexpr (F16) {
    if (bad && d.guard) atomicOr(d.guard, 1);
  }
  auto pixmap = [&](int j, int& b) {
    const int oy = oy0 + j / DIR_TW, ox = ox0 + j % DIR_TW;
    b = tb;
    return (oy < d.OH && ox < d.OW) ? oy * d.OW + ox : -1;
  };
  if constexpr (TG) {
    conv_epilogue_tapgemm<F16>(d, acc, cblk0, wave, lane, tid, OHW, pixmap, reinterpret_cast<float*>(Pst));
    return;
  }
  if (gridDim.z > 1) {  // raw partial sums of this K-part; conv_ksplit_reduce_kernel applies bias / act / epilogue
    accflow_conv_desc e = d;
    e.out = d.kws + (long long)blockIdx.z * d.B * d.Cout * OHW;
    e.out_bs = (long long)d.Cout * OHW;
    e.bias = nullptr;
    e.wscale16 = nullptr;  // (applied by the reduce kernel)
    e.out16 = nullptr;     // (written by the reduce kernel)
    e.cb = 0;              // (the partial sums are plain (B, Cout, OH, OW))
    conv_epilogue_impl<ACCFLOW_EPI_STORE, ACCFLOW_ACT_NONE, WC, WP, TCW, TP>(e, acc, cblk0, wc, wp, lane, OHW, pixmap);
    return;
  }
  if constexpr (S16) {
    // the update block's plain-store convolutions (convc1 / convc2 / convf1 / convf2 / the motion conv / the flow head's first
    // conv: store + ReLU into an S16 tensor) take the lean epilogue (conv_common.h) - a wave whose 32 * TCW rows all exist
    if ((ACCFLOW_DIRECT_LEAN || d.p32 || nstep <= ACCFLOW_DIRECT_LEAN_MAXSTEPS) && d.epi == ACCFLOW_EPI_STORE && !d.cb && !d.stats && cblk0 + (wc + 1) * TCW * 32 <= d.Cout &&
        (d.act == ACCFLOW_ACT_NONE || d.act == ACCFLOW_ACT_RELU)) {
      if (d.act == ACCFLOW_ACT_RELU) conv_epilogue_lean<ACCFLOW_ACT_RELU, WC, WP, TCW, TP>(d, acc, cblk0, wc, wp, lane, OHW, pixmap);
      else conv_epilogue_lean<ACCFLOW_ACT_NONE, WC, WP, TCW, TP>(d, acc, cblk0, wc, wp, lane, OHW, pixmap);
      return;
    }
  }
#ifdef ACCFLOW_DIRECT_ABL_NOEPI
  // (measurement builds, results INVALID: what do the GRU epilogues cost?  tools: profiles/r06_gru_epilogue_ablation.txt)
  if ((d.epi == ACCFLOW_EPI_GRU_ZR || d.epi == ACCFLOW_EPI_GRU_Q) && acc[0][0][0] != 12345.678f) return;
#endif
  if constexpr (S16 && W4 && KT == 5) {
    // the refinement loop's GRU epilogues on packed operands (pre-split state, pixel-major z / context addend): conv_common.h
    if (d.e0_fmt) {
      if (d.epi == ACCFLOW_EPI_GRU_ZR) conv_epilogue_gru16<true, TP>(d, acc, cblk0, wc, lane, OHW, pixmap);
      else conv_epilogue_gru16<false, TP>(d, acc, cblk0, wc, lane, OHW, pixmap);
      return;
    }
  }
  conv_epilogue_px<WC, WP, TCW, TP, decltype(pixmap), F16>(d, acc, cblk0, wc, wp, lane, OHW, pixmap, tb, trem * WP + wp);
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


}  // namespace

// launch one instantiation group (conv2d_direct_v*.hip); returns 0 or a hipError_t
int accflow_direct_launch_s16(const accflow_conv_desc& d, int tc, dim3 grid, hipStream_t st);
int accflow_direct_launch_s16tg(const accflow_conv_desc& d, dim3 grid, hipStream_t st);
int accflow_direct_launch_s16k9(const accflow_conv_desc& d, int tc, bool tapgemm, dim3 grid, hipStream_t st);   // 3x3 forms of the loop below
int accflow_direct_launch_s16k(const accflow_conv_desc& d, int kt, dim3 grid, hipStream_t st);   // tap-specialised loop (KT = 5), 128-channel kernel
int accflow_direct_launch_f16(const accflow_conv_desc& d, int tc, bool w4, dim3 grid, hipStream_t st);
int accflow_direct_launch_f16_norm(const accflow_conv_desc& d, int tc, dim3 grid, hipStream_t st);
int accflow_direct_launch_bf16(const accflow_conv_desc& d, int tc, int nt, bool w4, dim3 grid, hipStream_t st);
int accflow_direct_launch_bf16_norm(const accflow_conv_desc& d, int tc, int nt, dim3 grid, hipStream_t st);
