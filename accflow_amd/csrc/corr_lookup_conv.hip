// CorrBlock lookup FUSED with the motion encoder's first convolution (raft/corr.py:24-45 -> raft/update.py:89-91:
// cor = relu(convc1(corr)), a 1x1 convolution of the 4 x 81 lookup channels into 256).
//
// Why: as two launches the lookup writes 1 408 B per query pixel (the S16 form of its 324 taps) that convc1 reads back at
// once - 2 x 119 MB per refinement iteration at B = 11, 60 x 128 - and convc1, a 1x1 convolution with a 22-step
// reduction, spends its time staging those chunks and synchronising (179 TFLOP/s where the 3x3 shapes reach 400).  The
// lookup's output IS convc1's reduction axis, so the blended taps go from the lanes that computed them through 16 KB of
// LDS straight into the MFMA B operand; only relu(convc1) - 256 channels, pre-split - is written.
//
// Shape of the work.  Workgroup = 64 consecutive query pixels of one pair (half a 128-pixel block of the displaced
// pyramid, corr_disp.hip) x all 256 output channels, 8 waves in two roles:
//   * waves 0-3 SAMPLE: wave w reads pyramid LEVEL w for the 64 pixels, lane = pixel - the same loads, blend and rounding
//     as corr_lookup_disp_kernel (one range-checked buffer_load_dword per window cell, 256 contiguous bytes per wave and
//     tap when the lanes agree on the window origin), window rows requested PF rows ahead of the row pair being blended;
//   * waves 4-7 MULTIPLY: wave 4 + w owns output channels 64w .. 64w+63 for all 64 pixels (2 x 2 accumulator tiles of
//     32 x 32), A fragments straight from the pack in L2 two 16-deep steps ahead, B fragments from the LDS stage.
// Reduction order: super-step c = 0..9 is 32 deep = taps 8c .. 8c+7 (compute order n = j*9 + i: i along x, j along y) of
// each of the 4 levels, k = 32c + 8*level + t; then one 16-deep tail step whose first 4 entries are tap 80 of the 4
// levels (k = 320 + level) - 336 instead of the 352 the per-level padding of the S16 lookup costs.  The weight pack is
// accflow_conv_pack_patch16's over the re-indexed (256, 336, 1, 1) weight, i.e. the MFMA A fragments as they lie in L2.
// Every 8 taps a sampling wave writes one 16-byte hi and one 16-byte lo chunk per lane into the LDS stage of the
// super-step (double buffered: ONE workgroup barrier per 32-deep super-step); the multiplying waves run that
// super-step's 24 MFMAs each while the samplers blend the next one.
//
// Why two roles.  gfx950 counts every vector-memory load of a wave in ONE in-order counter, so a wave that requests
// window rows PF rows ahead (HBM, 1-2 us) AND weight fragments one step ahead (L2, ~0.25 us) waits for the rows whenever
// it waits for the fragments.  The first form of this kernel (4 waves, every wave sampling its level AND multiplying its
// 64 channels; git history) measured 83-90 us whatever PF (profiles/r05_lc1_unified_variants.txt).  Split, a sampling
// wave's counter holds only row requests and a multiplying wave's only weight fragments.
//
// Measured (B = 11, 60 x 128, profiles/r05_lc1_*.txt): 81 us against 56 + 72 = 128 us for the two launches it replaces.
// Compile-time ablations: without the output stores 54 us, without the window loads 54, without both 19 (MFMAs: free) -
// loads and stores do not overlap: the in-kernel timeline shows a workgroup living 24.6 us = 5.7 us until its first
// window rows have arrived + 10 super-steps of 1.2 us + 6.9 us of epilogue, with 2 workgroups per CU.  A persistent
// form (a workgroup walking over 2-3 tiles, its samplers starting the next tile underneath the epilogue of the last)
// measured 86-95 us and was dropped (profiles/r05_lc1_persistent.txt); so was a DECOUPLED form - the stage hand-over through a
// 6-stage LDS ring guarded by per-stage counters (ds_add / polling with s_sleep) instead of the workgroup barrier, samplers
// running up to 6 super-steps ahead across tiles: correct, 122-132 us (profiles/r05_lc1_decoupled.txt): the polling and the
// run-time stage addressing cost more than the barrier's lock-step.
#include "conv_common.h"
#include <utility>

namespace {

constexpr int R = 4, WIN = 2 * R + 2;
constexpr unsigned OOB = 0x40000000u;
// measurement builds only (-DACCFLOW_LC1_ABL=bits): 1 no output stores, 2 no MFMAs, 4 no window loads, 8 one A step only
#ifndef ACCFLOW_LC1_ABL
#define ACCFLOW_LC1_ABL 0
#endif
constexpr int ABL = ACCFLOW_LC1_ABL;

struct lc1_params {
  const float* l0; const float* l1; const float* l2; const float* l3;
  const float* coords;
  const void* wpatch16; const float* wscale16; const float* bias;
  void* out16; long long out16_bs;
  float* out; long long out_bs;
  int* guard;
  int B, H8, W8, Cout, CoutPad, act;
  unsigned long long* prof;   // tools only (accflow_debug_lc1_prof): 16 stamps per wave of the PROF instantiation
};

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

constexpr int NSUPER = 10, NSTEP = 21;

// split8_f16 (conv_common.h) with the range check kept in the VECTOR domain: the largest |x * s| seen so far as an integer
// maximum over the sign-cleared bit patterns (NaN > inf > every finite value), tested once at the end of the kernel.  The
// compare-per-value form leaves a 64-bit SGPR mask per compare; in this kernel's 2 000-instruction straight-line body the
// scheduler let those pile up (85 SGPR spills into VGPR lanes).
__device__ __forceinline__ void split8_f16_mx(const float (&x)[8], u32x4 (&out)[2], unsigned& mx, float s) {
  unsigned hi[4], lo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = x[2 * j] * s, b = x[2 * j + 1] * s;
    mx = max(mx, max(__builtin_bit_cast(unsigned, a) & 0x7FFFFFFFu, __builtin_bit_cast(unsigned, b) & 0x7FFFFFFFu));
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
  // (pin the running maximum HERE: nothing consumes it before the end of the kernel, and the scheduler otherwise sinks the
  // whole max chain there - keeping all 81 scaled taps alive, 70 of them in scratch)
  asm volatile("" : "+v"(mx));
}

// PF: window rows requested ahead of the row pair being blended; PROF: the stamped instantiation of tools/lc1_prof.py
template <int PF, bool PROF = false>
__global__ __launch_bounds__(512, 4) void corr_lookup_convc1_ws_kernel(const lc1_params a) {
  // PROF: stamp[0] = entry (s_memrealtime, 100 MHz), stamp[1..11] = after barrier 0..10, stamp[12] = exit, stamp[13] = the
  // role's set-up done (first window rows requested / first A fragments requested)
  unsigned long long stamp[14];
  if constexpr (PROF) stamp[0] = __builtin_amdgcn_s_memrealtime();
#define LC1_STAMP(i) do { if constexpr (PROF) stamp[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define LC1_STAMPS_OUT() do { if constexpr (PROF) { if (lane == 0 && a.prof) { \
    unsigned long long* q = a.prof + ((long long)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 16; \
    _Pragma("unroll") for (int i = 0; i < 14; ++i) q[i] = stamp[i]; } } } while (0)
  __shared__ u32x4 Bst[2 * 4 * 2 * 64];   // [stage][level][term][pixel]
  __shared__ float Tail[4 * 64];          // tap 80 of [level][pixel]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H8 = a.H8, W8 = a.W8, P = H8 * W8;
  const int b = blockIdx.y;
  const int pix0 = blockIdx.x * 64;
  constexpr float ASC = (float)(1 << ACCFLOW_F16_ASHIFT);

  if (wave < 4) {
    // ================= lookup role: level `wave` of the 64 pixels, lane = pixel =================
    const int lvl = wave;
    const int pix = pix0 + lane;
    const bool active = pix < P;
    const int pc = active ? pix : 0;
    const float* vol = lvl == 0 ? a.l0 : lvl == 1 ? a.l1 : lvl == 2 ? a.l2 : a.l3;
    const int Hl = H8 >> lvl, Wl = W8 >> lvl;
    const int PB = (P + 127) >> 7, pblk = pix0 >> 7;
    const long long slab = (long long)Hl * Wl * 128;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(vol + ((long long)b * PB + pblk) * slab), 0, (int)(slab * 4), 0x00020000);
    const float inv = 1.0f / (float)(1 << lvl);
    float cx = a.coords[((long long)b * 2 + 0) * P + pc] * inv;
    float cy = a.coords[((long long)b * 2 + 1) * P + pc] * inv;
    cx = fminf(fmaxf(cx, -1.0e6f), 1.0e6f);
    cy = fminf(fmaxf(cy, -1.0e6f), 1.0e6f);
    const float fx0 = floorf(cx), fy0 = floorf(cy);
    const float ax = cx - fx0, ay = cy - fy0;
    const int xs = (int)fx0 - R, ys = (int)fy0 - R;
    const float w00 = (1.0f - ax) * (1.0f - ay), w01 = ax * (1.0f - ay), w10 = (1.0f - ax) * ay, w11 = ax * ay;
    const int y1 = pc / W8, x1 = pc - y1 * W8;
    const int y1l = y1 >> lvl, x1l = x1 >> lvl;
    unsigned coloff[WIN];
#pragma unroll
    for (int q = 0; q < WIN; ++q) {
      const int x = xs + q;
      int m = x - x1l;
      if (m < 0) m += Wl;
      coloff[q] = (active && (unsigned)x < (unsigned)Wl) ? (unsigned)(m * 128 + (pc & 127)) * 4u : OOB;
    }
    const unsigned rowstride = (unsigned)Wl * 512u;
    auto rowoff = [&](int r) -> unsigned {
      const int y = ys + r;
      int m = y - y1l;
      if (m < 0) m += Hl;
      return (unsigned)y < (unsigned)Hl ? (unsigned)m * rowstride : OOB;
    };
    auto blend4 = [&](float a00, float a01, float a10, float a11) {
      return __builtin_fmaf(a11, w11, __builtin_fmaf(a10, w10, __builtin_fmaf(a01, w01, a00 * w00)));
    };
    float rows[PF + 1][WIN];
    auto load_row = [&](int r, float (&dst)[WIN]) {
      const unsigned ro = rowoff(r);
#pragma unroll
      for (int q = 0; q < WIN; ++q)
        dst[q] = (ABL & 4) ? __builtin_bit_cast(float, ro + coloff[q])
                           : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, ro + coloff[q], 0, 0));
    };
    unsigned mx = 0u;
    float buf[8];
#pragma unroll
    for (int r = 0; r < PF; ++r) load_row(r, rows[r]);
    LC1_STAMP(13);
    static_for<2 * R + 1>([&](auto j_) {
      constexpr int j = decltype(j_)::value;
      if constexpr (j + PF < WIN) load_row(j + PF, rows[(j + PF) % (PF + 1)]);
      const float (&r0)[WIN] = rows[j % (PF + 1)];
      const float (&r1)[WIN] = rows[(j + 1) % (PF + 1)];
      static_for<2 * R + 1>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        constexpr int n = j * 9 + i;
        buf[n & 7] = blend4(r0[i], r0[i + 1], r1[i], r1[i + 1]);
        if constexpr ((n & 7) == 7) {
          constexpr int c = n >> 3;
          u32x4 terms[2];
          split8_f16_mx(buf, terms, mx, ASC);
          Bst[(((c & 1) * 4 + lvl) * 2 + 0) * 64 + lane] = terms[0];
          Bst[(((c & 1) * 4 + lvl) * 2 + 1) * 64 + lane] = terms[1];
          __syncthreads();   // barrier c: stage c & 1 is complete
          LC1_STAMP(c + 1);
        }
      });
    });
    Tail[lvl * 64 + lane] = buf[0];
    // the tail value's range check (the multiplying waves split it): |x * 2^4| as a bit pattern
    mx = max(mx, __builtin_bit_cast(unsigned, buf[0] * ASC) & 0x7FFFFFFFu);
    __syncthreads();         // barrier 10: Tail is complete
    LC1_STAMP(11);
    if (!(mx < 0x477FF000u) && a.guard) atomicOr(a.guard, 1);   // 0x477FF000 = 65520.0f
    LC1_STAMP(12);
    LC1_STAMPS_OUT();
    return;
  }

  // ================= GEMM role: output channels 64 * (wave - 4) .. + 63 for the 64 pixels =================
  const int wc = wave - 4;
  const int l31 = lane & 31, kh = lane >> 5;
  constexpr int COUTPAD = 256;   // (host-checked)
  constexpr unsigned step_bytes = 2u * COUTPAD * 16u, term_bytes = (unsigned)NSTEP * step_bytes;
  const __amdgpu_buffer_rsrc_t rsrcw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wpatch16), 0,
                                                                        (int)(3u * term_bytes), 0x00020000);
  const unsigned avoff = (unsigned)((kh * COUTPAD + wc * 64 + l31) * 16);
  bf16x8 aF[3][2][2];   // [16-deep step mod 3][term][channel tile]: requested two steps ahead
  auto load_a_step = [&](auto s_) {
    constexpr int s = decltype(s_)::value;
    if constexpr (s < NSTEP) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
          aF[s % 3][t][tc] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
              rsrcw, (int)(avoff + tc * 512), (int)(t * term_bytes + ((ABL & 8) ? 0 : s) * step_bytes), 0));
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int tc = 0; tc < 2; ++tc)
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tc][tp][r] = 0.0f;
  auto mfma3 = [&](const bf16x8 (&A)[2][2], const bf16x8 (&Bf)[2][2]) {
    constexpr int PA[3] = {1, 0, 0}, PBI[3] = {0, 1, 0};
    if constexpr ((ABL & 2) != 0) {
      acc[0][0][0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, A[0][0])[0] ^ __builtin_bit_cast(u32x4, Bf[0][0])[0] ^
                                                __builtin_bit_cast(u32x4, A[1][1])[1] ^ __builtin_bit_cast(u32x4, Bf[1][1])[1]);
      return;
    }
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
      for (int tc = 0; tc < 2; ++tc)
#pragma unroll
        for (int tp = 0; tp < 2; ++tp)
          acc[tc][tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[PA[pr]][tc]),
                                                              __builtin_bit_cast(f16x8, Bf[PBI[pr]][tp]), acc[tc][tp], 0, 0, 0);
  };
  load_a_step(std::integral_constant<int, 0>{});
  load_a_step(std::integral_constant<int, 1>{});
  LC1_STAMP(13);
  static_for<NSUPER>([&](auto c_) {
    constexpr int c = decltype(c_)::value;
    __syncthreads();         // barrier c
    LC1_STAMP(c + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks == 0) load_a_step(std::integral_constant<int, 2 * c + 2>{});
      else load_a_step(std::integral_constant<int, 2 * c + 3>{});
      bf16x8 Bf[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int tp = 0; tp < 2; ++tp)
          Bf[t][tp] = __builtin_bit_cast(bf16x8, Bst[(((c & 1) * 4 + 2 * ks + kh) * 2 + t) * 64 + tp * 32 + l31]);
      mfma3(aF[(2 * c + ks) % 3], Bf);
    }
  });
  __syncthreads();           // barrier 10
  LC1_STAMP(11);
  {
    unsigned mxd = 0u;       // (range-checked by the sampling waves)
    bf16x8 Bf[2][2];
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
      float x[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) x[q] = kh == 0 ? Tail[q * 64 + tp * 32 + l31] : 0.0f;
#pragma unroll
      for (int q = 4; q < 8; ++q) x[q] = 0.0f;
      u32x4 terms[2];
      split8_f16_mx(x, terms, mxd, ASC);
      Bf[0][tp] = __builtin_bit_cast(bf16x8, terms[0]);
      Bf[1][tp] = __builtin_bit_cast(bf16x8, terms[1]);
    }
    mfma3(aF[(NSTEP - 1) % 3], Bf);
  }
  accflow_conv_desc e = {};
  e.B = a.B; e.Cout = a.Cout; e.CoutPad = COUTPAD;
  e.bias = a.bias; e.wscale16 = a.wscale16;
  e.out = a.out; e.out_bs = a.out_bs;
  e.out16 = a.out16; e.out16_bs = a.out16_bs;
  e.guard = a.guard;
  auto pixmap = [&](int jj, int& bb) {
    bb = b;
    return pix0 + jj < P ? pix0 + jj : -1;
  };
  if ((ABL & 1) && acc[0][0][0] != 12345.678f) return;
  // (one 32-channel tile at a time: the epilogue's scale / bias vectors of both tiles at once cost 31 spilled registers at
  // this kernel's 128-register budget)
#pragma unroll
  for (int tc = 0; tc < 2; ++tc) {
    f32x16 (&at)[1][2] = *reinterpret_cast<f32x16 (*)[1][2]>(&acc[tc]);
    if (a.act == ACCFLOW_ACT_RELU) conv_epilogue_lean<ACCFLOW_ACT_RELU, 4, 1, 1, 2>(e, at, wc * 64 + tc * 32, 0, 0, lane, P, pixmap);
    else conv_epilogue_lean<ACCFLOW_ACT_NONE, 4, 1, 1, 2>(e, at, wc * 64 + tc * 32, 0, 0, lane, P, pixmap);
  }
  if constexpr (PROF) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  LC1_STAMP(12);
  LC1_STAMPS_OUT();
#undef LC1_STAMP
#undef LC1_STAMPS_OUT
}


}  // namespace

#ifdef ACCFLOW_LC1_PROF
// TOOLS BUILDS ONLY (tools/lc1_prof.py builds its own library with -DACCFLOW_LC1_PROF into another directory: python -m
// accflow_amd.build --libdir=... --unit-define=corr_lookup_conv:ACCFLOW_LC1_PROF): device buffer of 16 x 8 x workgroups uint64
// that the stamped instantiation of the fused kernel fills.  The product library has neither the global nor the export.
namespace { unsigned long long* g_lc1_prof = nullptr; }
extern "C" int accflow_debug_lc1_prof(unsigned long long* buf) { g_lc1_prof = buf; return 0; }
#endif

// reduction length of the fused kernel's weight pack (see the header): 10 x 32 + 16
extern "C" int accflow_corr_lookup_convc1_kpad(void) { return 16 * NSTEP; }

extern "C" int accflow_corr_lookup_convc1_s16(const float* lvl0, const float* lvl1, const float* lvl2, const float* lvl3,
                                              const float* coords, const void* wpatch16, const float* wscale16,
                                              const float* bias, void* out16, long long out16_bs, float* out, long long out_bs,
                                              int act, int* guard, int B, int H8, int W8, int Cout, void* stream) {
  if (!lvl0 || !lvl1 || !lvl2 || !lvl3 || !coords || !wpatch16 || !wscale16 || (!out16 && !out) || B <= 0 ||
      !accflow_corr_disp_supported(H8, W8) || Cout != 256 || accflow_conv_coutpad(Cout) != 256 ||
      (act != ACCFLOW_ACT_NONE && act != ACCFLOW_ACT_RELU))
    return 1;
  lc1_params a;
  a.l0 = lvl0; a.l1 = lvl1; a.l2 = lvl2; a.l3 = lvl3; a.coords = coords;
  a.wpatch16 = wpatch16; a.wscale16 = wscale16; a.bias = bias;
  a.out16 = out16; a.out16_bs = out16_bs; a.out = out; a.out_bs = out_bs; a.guard = guard;
  a.B = B; a.H8 = H8; a.W8 = W8; a.Cout = Cout; a.CoutPad = accflow_conv_coutpad(Cout); a.act = act;
  a.prof = nullptr;
  const dim3 grid(cdiv((long long)H8 * W8, 64), B), block(512);
  hipStream_t st = as_stream(stream);
#ifdef ACCFLOW_LC1_PROF
  a.prof = g_lc1_prof;
  if (a.prof) {
    hipLaunchKernelGGL((corr_lookup_convc1_ws_kernel<3, true>), grid, block, 0, st, a);
    ACCFLOW_RETURN_LAUNCH_STATUS();
  }
#endif
  // (window rows requested 3 rows ahead; 2 and 4 measured the same within 2 %, profiles/r05_lc1_unified_variants.txt)
  hipLaunchKernelGGL((corr_lookup_convc1_ws_kernel<3>), grid, block, 0, st, a);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
