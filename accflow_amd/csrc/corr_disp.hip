// Displacement-indexed correlation pyramid: the hot-path layout of the all-pairs volume.
//
// Why: in the reference layout (raft/corr.py:8-22: one (Hl x Wl) plane per query pixel) a lookup reads a private
// 10x10 window per pixel and level - 10 row segments of 40 B, each dragging in one or two 128-B lines that no
// other pixel ever uses (PMC: ~500 MB of HBM traffic per launch against 245 MB algorithmic), and the 64 lanes
// of a wave touch 64 different lines per load.  Neighbouring query pixels, however, look at neighbouring TARGET
// pixels (flow is piecewise smooth), i.e. at the same DISPLACEMENT.  So level l is stored as
//
//   E_l[b][p / 128][dy][dx][p % 128],   p = y1*W8 + x1 the query pixel (blocks of 128, the last one padded),
//   dy = (y' - (y1 >> l)) mod Hl,  dx = (x' - (x1 >> l)) mod Wl   for target cell (y', x') of level l,
//
// a permutation of the reference volume V_l[b][p][y'][x'] (padded to a multiple of 128 query pixels).  A lookup tap
// (row r, column q of the 10x10 window) then reads, for the 64 consecutive query pixels of a wave, 64 consecutive
// floats whenever their integer window origins agree relative to the pixel - two full 128-B lines per load
// instruction, every fetched byte used.  Incoherent flow (noise) degrades to one line per lane and tap; results
// are identical either way.  Blocking p by 128 keeps the displacement cells of one block of query pixels adjacent
// in memory (neighbouring dx are 512 B apart, a whole level of one block is <= 3.9 MB), so the GEMM's tile stores,
// the pooling passes and the window reads all stay inside a few DRAM pages instead of striding by 4*P bytes.
//
// Level 0 is written in this layout straight from the matrix-core GEMM (conv2d.hip, corr_disp_store: the
// 128x128 accumulator tile is sheared through LDS so that stores run along p); levels 1..3 are pooled in
// displacement space with F.avg_pool2d(2, 2) semantics (floor sizes, ((a+b)+c)+d then * 0.25, pool of pool).
// The lookup math is corr_lookup.hip's (CorrBlock.__call__, raft/corr.py:24-45 + bilinear_sampler).
#include "common.h"

namespace {

constexpr int R = 4, WIN = 2 * R + 2;
constexpr unsigned OOB = 0x40000000u;  // >= every level's byte size (host-checked): a buffer load there returns 0

// out level l+1 from level l: thread = query pixel p, workgroup row = one (dy, dx) cell of the output level
__global__ __launch_bounds__(256) void corr_disp_pool_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             int Hi, int Wi, int W8, int P, int lsrc) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int Ho = Hi >> 1, Wo = Wi >> 1;
  const int dy = blockIdx.y / Wo, dx = blockIdx.y - dy * Wo;
  const int y1 = p / W8, x1 = p - y1 * W8;
  int yo = dy + (y1 >> (lsrc + 1)), xo = dx + (x1 >> (lsrc + 1));  // target cell of the output level
  if (yo >= Ho) yo -= Ho;
  if (xo >= Wo) xo -= Wo;
  int sy0 = 2 * yo - (y1 >> lsrc), sx0 = 2 * xo - (x1 >> lsrc);     // displaced index of source cell (2yo, 2xo)
  int sy1 = sy0 + 1, sx1 = sx0 + 1;
  if (sy0 < 0) sy0 += Hi;
  if (sy1 < 0) sy1 += Hi;
  if (sx0 < 0) sx0 += Wi;
  if (sx1 < 0) sx1 += Wi;
  const int PB = (P + 127) >> 7;
  const float* src = in + (((long long)blockIdx.z * PB + (p >> 7)) * Hi * Wi) * 128 + (p & 127);
  const float a = src[(long long)(sy0 * Wi + sx0) * 128], b = src[(long long)(sy0 * Wi + sx1) * 128];
  const float c = src[(long long)(sy1 * Wi + sx0) * 128], d = src[(long long)(sy1 * Wi + sx1) * 128];
  out[((((long long)blockIdx.z * PB + (p >> 7)) * Ho * Wo) + blockIdx.y) * 128 + (p & 127)] = (((a + b) + c) + d) * 0.25f;
}

// One wave = 64 consecutive query pixels of one pair at ONE pyramid level (blockIdx.z), lane = pixel.
// OUT16: the result goes out PRE-SPLIT (accflow_corr_lookup_disp_s16): a level's 81 taps in compute order n = j*9 + i
// fill 11 octets (88 channels, the last 7 zero); every 8 blended values a lane writes one 16-byte hi and one 16-byte lo
// chunk - 1 KB contiguous per wave and store instruction instead of 256 B.
typedef unsigned lu32x4 __attribute__((ext_vector_type(4)));
template <int PF, bool OUT16 = false>
__global__ __launch_bounds__(64) void corr_lookup_disp_kernel(const float* __restrict__ l0, const float* __restrict__ l1,
                                                               const float* __restrict__ l2, const float* __restrict__ l3,
                                                               const float* __restrict__ coords, float* __restrict__ out,
                                                               long long out_bs, int H8, int W8, int* guard = nullptr) {
  const int P = H8 * W8;
  const int lane = threadIdx.x & 63;
  const int lvl = blockIdx.z;
  const int b = blockIdx.y;
  const int pix = blockIdx.x * 64 + lane;
  const bool active = pix < P;
  const int pc = active ? pix : 0;

  const float* vol = lvl == 0 ? l0 : lvl == 1 ? l1 : lvl == 2 ? l2 : l3;
  const int Hl = H8 >> lvl, Wl = W8 >> lvl;
  // this wave's 64 pixels lie in ONE 128-pixel block: the descriptor covers that block's (Hl, Wl, 128) slab
  const int PB = (P + 127) >> 7, pblk = (blockIdx.x * 64) >> 7;
  const long long slab = (long long)Hl * Wl * 128;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(vol + ((long long)b * PB + pblk) * slab), 0, (int)(slab * 4), 0x00020000);

  const float inv = 1.0f / (float)(1 << lvl);
  float cx = coords[((long long)b * 2 + 0) * P + pc] * inv;
  float cy = coords[((long long)b * 2 + 1) * P + pc] * inv;
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

  // one explicit fma chain: the fp32 and the S16 instantiation must blend with the SAME roundings (left to the
  // compiler's contraction they did not)
  auto blend4 = [&](float a00, float a01, float a10, float a11) {
    return __builtin_fmaf(a11, w11, __builtin_fmaf(a10, w10, __builtin_fmaf(a01, w01, a00 * w00)));
  };
  float* o = out + (long long)b * out_bs + (long long)(lvl * 81) * P + pc;
  // OUT16: `out` is the S16 tensor (out_bs in 4-byte words); chunk (octet, term) of this pixel
  lu32x4* o16 = reinterpret_cast<lu32x4*>(out + (long long)b * out_bs) + pc;
  float buf[8];
  bool bad = false;
  auto flush = [&](int oct, int count) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    unsigned h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a = (2 * k < count) ? buf[2 * k] * (float)(1 << ACCFLOW_F16_ASHIFT) : 0.0f;
      const float c = (2 * k + 1 < count) ? buf[2 * k + 1] * (float)(1 << ACCFLOW_F16_ASHIFT) : 0.0f;
      bad |= !(fabsf(a) < 65520.0f) | !(fabsf(c) < 65520.0f);
      const f2 v = {a, c};
      const h2 hq = __builtin_convertvector(v, h2);
      const f2 back = __builtin_convertvector(hq, f2);
      const f2 r = {a - back[0], c - back[1]};
      const h2 lq = __builtin_convertvector(r, h2);
      h[k] = __builtin_bit_cast(unsigned, hq);
      l[k] = __builtin_bit_cast(unsigned, lq);
    }
    if (active) {
      o16[(long long)((lvl * 11 + oct) * 2 + 0) * P] = lu32x4{h[0], h[1], h[2], h[3]};
      o16[(long long)((lvl * 11 + oct) * 2 + 1) * P] = lu32x4{l[0], l[1], l[2], l[3]};
    }
  };
  // window rows are streamed PF rows ahead of the row pair being blended (a rotating set of PF + 1 row buffers)
  float rows[PF + 1][WIN];
  auto load_row = [&](int r, float (&dst)[WIN]) {
    const unsigned ro = rowoff(r);
#pragma unroll
    for (int q = 0; q < WIN; ++q)
      dst[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, ro + coloff[q], 0, 0));
  };
#pragma unroll
  for (int r = 0; r < PF; ++r) load_row(r, rows[r]);
#pragma unroll
  for (int j = 0; j < 2 * R + 1; ++j) {
    if (j + PF < WIN) load_row(j + PF, rows[(j + PF) % (PF + 1)]);
    const float (&r0)[WIN] = rows[j % (PF + 1)];
    const float (&r1)[WIN] = rows[(j + 1) % (PF + 1)];
    if constexpr (OUT16) {
#pragma unroll
      for (int i = 0; i < 2 * R + 1; ++i) {
        const int n = j * 9 + i;
        buf[n & 7] = blend4(r0[i], r0[i + 1], r1[i], r1[i + 1]);
        if ((n & 7) == 7 || n == 80) flush(n >> 3, (n & 7) + 1);
      }
    } else if (active) {
#pragma unroll
      for (int i = 0; i < 2 * R + 1; ++i) {
        o[(long long)(i * 9 + j) * P] = blend4(r0[i], r0[i + 1], r1[i], r1[i + 1]);
      }
    }
  }
  if constexpr (OUT16) {
    if (bad && guard) atomicOr(guard, 1);
  }
}

}  // namespace

extern "C" int accflow_corr_disp_supported(int H8, int W8) {
  // one 128-pixel block's level-0 slab must stay below the out-of-range marker of the range-checked buffer loads
  return H8 >= 8 && W8 >= 8 && H8 < 65536 && W8 < 65536 && (long long)H8 * W8 * 512 <= (long long)OOB;
}

// floats per pair of level l: the query pixels are padded to a multiple of 128
extern "C" long long accflow_corr_disp_level_elems(int H8, int W8, int level) {
  return (long long)(((long long)H8 * W8 + 127) / 128) * 128 * (H8 >> level) * (W8 >> level);
}

// levels first+1..3 from the displaced level `first` (0: everything below level 0; 1: level 1 already written)
int accflow_corr_disp_pool_from(const float* lvl0, float* lvl1, float* lvl2, float* lvl3, int B, int H8, int W8, int first,
                                hipStream_t st) {
  const int P = H8 * W8;
  const float* lv[4] = {lvl0, lvl1, lvl2, lvl3};
  float* dst[4] = {nullptr, lvl1, lvl2, lvl3};
  for (int l = first; l < 3; ++l) {
    const int Hi = H8 >> l, Wi = W8 >> l;
    hipLaunchKernelGGL(corr_disp_pool_kernel, dim3(cdiv(P, 256), (Hi >> 1) * (Wi >> 1), B), dim3(256), 0, st, lv[l],
                       dst[l + 1], Hi, Wi, W8, P, l);
  }
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

// levels 1..3 from a displaced level 0 (exposed for tests; accflow_corr_volume_disp_f32 pools from level 1 on when
// its GEMM has emitted level 1)
extern "C" int accflow_corr_disp_pool_f32(const float* lvl0, float* lvl1, float* lvl2, float* lvl3, int B, int H8,
                                          int W8, void* stream) {
  if (!lvl0 || !lvl1 || !lvl2 || !lvl3 || B <= 0 || !accflow_corr_disp_supported(H8, W8)) return 1;
  return accflow_corr_disp_pool_from(lvl0, lvl1, lvl2, lvl3, B, H8, W8, 0, as_stream(stream));
}

extern "C" int accflow_corr_lookup_disp_f32(const float* lvl0, const float* lvl1, const float* lvl2,
                                            const float* lvl3, const float* coords, float* out, long long out_bs,
                                            int B, int H8, int W8, void* stream) {
  if (!lvl0 || !lvl1 || !lvl2 || !lvl3 || !coords || !out || B <= 0 || !accflow_corr_disp_supported(H8, W8)) return 1;
  // Measured (B = 11, 60x128; zero / mixed flow): rows prefetched ahead of the blend 1, 2, 3, 5: the same (the kernel
  // is bound by the lines it fetches, not by latency); one 64-thread workgroup per (64 pixels, level) instead of one
  // 256-thread workgroup with a wave per level: 44.2 -> 41.2 / 64.1 -> 59.9 us (a level's wave no longer holds its
  // workgroup's slot until the slowest level is done; level 0, the heaviest, is dispatched first); non-temporal loads:
  // 42.8 / 88.9 us (adjacent taps re-read the same lines through L1 / L2 when lanes disagree on the window origin).
  hipLaunchKernelGGL(corr_lookup_disp_kernel<1>, dim3(cdiv((long long)H8 * W8, 64), B, 4), dim3(64), 0, as_stream(stream),
                     lvl0, lvl1, lvl2, lvl3, coords, out, out_bs, H8, W8);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_corr_lookup_disp_s16(const float* lvl0, const float* lvl1, const float* lvl2, const float* lvl3,
                                            const float* coords, void* out16, long long out16_bs, int* guard, int B, int H8,
                                            int W8, void* stream) {
  if (!lvl0 || !lvl1 || !lvl2 || !lvl3 || !coords || !out16 || B <= 0 || !accflow_corr_disp_supported(H8, W8)) return 1;
  hipLaunchKernelGGL((corr_lookup_disp_kernel<1, true>), dim3(cdiv((long long)H8 * W8, 64), B, 4), dim3(64), 0,
                     as_stream(stream), lvl0, lvl1, lvl2, lvl3, coords, reinterpret_cast<float*>(out16), out16_bs, H8, W8, guard);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
