// The encoders' stem: 7x7, stride 2, pad 3 convolution of the 3-channel image into 64 channels (extractor.py:140,201-205),
// fp16 split arithmetic (ACCFLOW_CONV_F16X3), on the matrix cores.
//
// It ran on the im2col kernel at 83-98 TFLOP/s (0.54 ms per C3 step for the 20 images of the three encoders,
// profiles/r03_conv_shapes.txt): that kernel gathers every one of the 147 reduction elements of a pixel from global memory
// with its own bounds logic.  Here a workgroup stages the 13 x 69 x 3 input patch of its 4 x 32 output pixels ONCE in LDS
// - columns de-interleaved by parity so that the stride-2 reads of 32 neighbouring output pixels are consecutive words; since
// round 5 every element is split into fp16 hi / lo (x 2^4, range-checked) ONCE while it is staged, one word = {hi, lo} - and
// every lane builds its B fragments (8 consecutive k of one pixel) from it with 8 conflict-free ds_read_b32 at compile-time
// offsets and 8 v_perm (the hi halves, the lo halves) and feeds the same 3-MFMA product as the direct kernel.  The weights are the im2col kernel's fp16 pack (accflow_conv_pack_split16: k = c*49 + ky*7 + kx,
// padded to 160 = 10 steps), read as A fragments straight from L2.  The epilogue is the shared one, so the result can leave
// pre-split (accflow_conv_desc.out16: the S16 tensor layer1 stages by LDS DMA - no fp32 round trip and no to_s16 pass) or
// raw with InstanceNorm statistics (fnet).  Same products, same fp32 accumulation order along k as the im2col form.
#include "conv_common.h"

namespace {

constexpr int ST_TH = 4, ST_TW = 32, ST_PH = 13, ST_PWH = 36;   // patch: 13 rows x 2 parities x 36 (35 / 34 used) columns
constexpr int ST_K = 147, ST_STEPS = 10;

constexpr int ST_PWORDS = 3 * ST_PH * 2 * ST_PWH;               // patch words; word ST_PWORDS.. : zeros for the padded k (147..159)
__host__ __device__ __forceinline__ constexpr int st_koff(int k) {   // word offset of reduction element k inside the patch of pixel (0, 0)
  return (((k / 49) * ST_PH + (k % 49) / 7) * 2 + ((k % 7) & 1)) * ST_PWH + ((k % 7) >> 1);
}

// waves per SIMD the register budget is sized for (a lower bound for the compiler).  Round 5, before the patch was split at staging:
// 226 VGPRs at 2; forcing 3 / 4 (168 / 128 registers) cost 332 / 464 us per launch against 147 (tools/stem_bench.py, 7 images of
// 480 x 1024, S16 + ReLU output) - spills.  With the split moved to the staging pass the kernel needs 85 registers (5 waves per
// SIMD fit) and takes 100 us (143 instead of 196 with raw output + InstanceNorm statistics).
#ifndef ACCFLOW_STEM_WAVES
#define ACCFLOW_STEM_WAVES 2
#endif
__global__ __launch_bounds__(256, ACCFLOW_STEM_WAVES) void conv_stem7_kernel(const accflow_conv_desc d) {
  constexpr int WC = 2, WP = 2, TCW = 1, TP = 2;
  // round 5: the patch is split ONCE, when it is staged - a word holds fp16 hi (low half) and lo (high half) of x * 2^4 - instead
  // of in every fragment that uses the element (each of the 2 691 patch elements feeds ~12 (pixel, tap) pairs of the tile: the
  // loop spent ~100 VALU instructions per step on conversions next to 6 MFMAs); a fragment is now 8 LDS words and 8 v_perm.
  __shared__ unsigned P[ST_PWORDS + 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave / WP, wp = wave % WP;
  const int l31 = lane & 31, kh = lane >> 5;
  const int OHW = d.OH * d.OW;
  const int tilesX = (d.OW + ST_TW - 1) / ST_TW, tilesY = (d.OH + ST_TH - 1) / ST_TH;
  const int tb = blockIdx.x / (tilesX * tilesY), trem = blockIdx.x - tb * tilesX * tilesY;
  const int oy0 = (trem / tilesX) * ST_TH, ox0 = (trem % tilesX) * ST_TW;
  // ---- the patch: input rows 2*oy0 - 3 .. + 12, columns 2*ox0 - 3 .. + 68 of the 3 channels, zero outside the image ----
  {
    const float* src = d.in0 + (long long)tb * d.in0_bs;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    // (all 11 loads of a thread are in flight before the first one is waited for: a rolled loop paid an HBM round trip per
    // element and made the kernel 2x slower than the im2col form it replaces)
    constexpr int NE = 3 * ST_PH * 69, NIT = (NE + 255) / 256;
    const __amdgpu_buffer_rsrc_t rsrci = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(src), 0, (int)(unsigned)(3LL * d.H * d.W * 4), 0x00020000);
    float v[NIT];
    int slot[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + 256 * it;
      const int c = i / (ST_PH * 69), r = (i - c * ST_PH * 69) / 69, col = i - (c * ST_PH + r) * 69;
      const int iy = iy0 + r, ix = ix0 + col;
      const bool ok = i < NE && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
      slot[it] = i < NE ? ((c * ST_PH + r) * 2 + (col & 1)) * ST_PWH + (col >> 1) : -1;
      v[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                            rsrci, ok ? (int)(((long long)c * d.H * d.W + (long long)iy * d.W + ix) * 4) : -1, 0, 0));
    }
    constexpr float ASC0 = (float)(1 << ACCFLOW_F16_ASHIFT);
    bool bad0 = false;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (slot[it] >= 0) {      // split8_f16's arithmetic, one element: hi = fp16(x 2^4), lo = fp16(x 2^4 - hi)
        const float a = v[it] * ASC0;
        bad0 |= !(fabsf(a) < 65520.0f);
        const _Float16 hq = (_Float16)a;
        const _Float16 lq = (_Float16)(a - (float)hq);
        P[slot[it]] = (unsigned)__builtin_bit_cast(unsigned short, hq) | ((unsigned)__builtin_bit_cast(unsigned short, lq) << 16);
      }
    if (tid < 64) P[ST_PWORDS + tid] = 0u;
    if (bad0 && d.guard) atomicOr(d.guard, 1);
  }
  // ---- A fragments: [term][k/8][CoutPad][8] fp16 (accflow_conv_pack_split16) ----
  const long long oct_bytes = (long long)d.CoutPad * 16, term_bytes = (long long)(d.Kpad / 8) * oct_bytes;
  const __amdgpu_buffer_rsrc_t rsrcw =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.wsplit16), 0, (int)(unsigned)(2 * term_bytes), 0x00020000);
  const unsigned avoff = (unsigned)((kh * d.CoutPad + wc * 32 + l31) * 16);
  f32x16 acc[TCW][TP];
#pragma unroll
  for (int tp = 0; tp < TP; ++tp)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][tp][r] = 0.0f;
  int pbase[TP];
#pragma unroll
  for (int tp = 0; tp < TP; ++tp) pbase[tp] = (2 * (wp * TP + tp)) * 2 * ST_PWH + l31;   // output row -> patch row 2 * row
  __syncthreads();
#pragma unroll
  for (int step = 0; step < ST_STEPS; ++step) {
    bf16x8 a[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
      a[t] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                            rsrcw, (int)avoff, (int)(unsigned)(t * term_bytes + (long long)step * 2 * oct_bytes), 0));
    bf16x8 b[2][TP];
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) {
      unsigned wq[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {     // lane half 0: k = 16 step + j, half 1: k = 16 step + 8 + j; padded k read the zero words
        const int k0 = step * 16 + j, k1 = step * 16 + 8 + j;
        const int o0 = k0 < ST_K ? pbase[tp] + st_koff(k0 < ST_K ? k0 : 0) : ST_PWORDS + l31;
        const int o1 = k1 < ST_K ? pbase[tp] + st_koff(k1 < ST_K ? k1 : 0) : ST_PWORDS + l31;
        wq[j] = P[kh ? o1 : o0];
      }
      unsigned hi[4], lo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        hi[j] = __builtin_amdgcn_perm(wq[2 * j + 1], wq[2 * j], 0x05040100u);   // low halves of the two words
        lo[j] = __builtin_amdgcn_perm(wq[2 * j + 1], wq[2 * j], 0x07060302u);   // high halves
      }
      { const u32x4 t = {hi[0], hi[1], hi[2], hi[3]}; b[0][tp] = __builtin_bit_cast(bf16x8, t); }
      { const u32x4 t = {lo[0], lo[1], lo[2], lo[3]}; b[1][tp] = __builtin_bit_cast(bf16x8, t); }
    }
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
      for (int tp = 0; tp < TP; ++tp)
        acc[0][tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[PA[pr]]), __builtin_bit_cast(f16x8, b[PB[pr]][tp]),
                                                            acc[0][tp], 0, 0, 0);
  }
  auto pixmap = [&](int j, int& b) {
    const int oy = oy0 + j / ST_TW, ox = ox0 + j % ST_TW;
    b = tb;
    return (oy < d.OH && ox < d.OW) ? oy * d.OW + ox : -1;
  };
  if (d.epi == ACCFLOW_EPI_STORE && !d.stats && (wc + 1) * 32 <= d.Cout &&
      (d.act == ACCFLOW_ACT_NONE || d.act == ACCFLOW_ACT_RELU)) {
    if (d.act == ACCFLOW_ACT_RELU) conv_epilogue_lean<ACCFLOW_ACT_RELU, WC, WP, TCW, TP>(d, acc, 0, wc, wp, lane, OHW, pixmap);
    else conv_epilogue_lean<ACCFLOW_ACT_NONE, WC, WP, TCW, TP>(d, acc, 0, wc, wp, lane, OHW, pixmap);
    return;
  }
  conv_epilogue_px<WC, WP, TCW, TP, decltype(pixmap), true>(d, acc, 0, wc, wp, lane, OHW, pixmap, tb, trem * WP + wp);
}

}  // namespace

bool accflow_conv_stem_eligible(const accflow_conv_desc& d) {
  return d.mode == ACCFLOW_CONV_F16X3 && d.wsplit16 && d.wscale16 && !d.in1 && !d.in_fmt && !d.nsrc && d.C0 == 3 && d.KH == 7 &&
         d.KW == 7 && d.stride == 2 && d.padH == 3 && d.padW == 3 && d.Cout > 32 && d.Cout <= 64 && !d.offset && !d.in_norm &&
         !d.wsplit_bs && !d.cb && d.Kpad == 160 && d.OH == (d.H + 6 - 7) / 2 + 1 && d.OW == (d.W + 6 - 7) / 2 + 1;
}

int accflow_launch_conv_stem(const accflow_conv_desc& d, hipStream_t st) {
  const int tiles = cdiv(d.OW, ST_TW) * cdiv(d.OH, ST_TH);
  ACCFLOW_DRY_RUN(tiles * 2);   // one statistics slot per wave along the pixels (WP = 2)
  hipLaunchKernelGGL(conv_stem7_kernel, dim3((unsigned)((long long)d.B * tiles)), dim3(256), 0, st, d);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
