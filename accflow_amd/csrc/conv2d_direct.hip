#include "conv2d_direct_kernel.h"

namespace {

// out = epilogue(act(sum_z part[z] + bias)): the K-parts are added in the fixed order z = 0, 1, ... (deterministic);
// one thread per output element (the form used when no S16 copy is requested: measured 9.6 vs 13.3 us per launch for
// the octet-per-thread form below on the batch-1 fusion chain)
__global__ __launch_bounds__(256) void conv_ksplit_reduce_kernel(const accflow_conv_desc d, int Z) {
  const int OHW = d.OH * d.OW;
  const long long n = (long long)d.B * d.Cout * OHW;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int px = (int)(i % OHW);
  const int ch = (int)((i / OHW) % d.Cout);
  const int b = (int)(i / ((long long)OHW * d.Cout));
  float v = d.kws[i];
  for (int z = 1; z < Z; ++z) v += d.kws[(long long)z * n + i];
  v = fmaf(v, d.wscale16 ? d.wscale16[(d.wsplit_bs ? (long long)b * d.CoutPad : 0) + ch] : 1.0f, d.bias ? d.bias[ch] : 0.0f);
  if (d.pre && (d.epi == ACCFLOW_EPI_GRU_ZR || d.epi == ACCFLOW_EPI_GRU_Q)) v += d.pre[b * d.pre_bs + (long long)ch * OHW + px];
  v = apply_act(v, d.act);
  const long long o = (long long)ch * OHW + px;
  const int half = d.Cout >> 1;
  if (d.cb) {  // channel-block scatter (STORE / ACCUM): see accflow_conv_desc.cb
    const int blk = ch / d.cb;
    const long long oc = (long long)(ch - blk * d.cb) * OHW + px;
    const float r = d.epi == ACCFLOW_EPI_ACCUM ? d.e0[b * d.e0_bs + blk * d.e0_cbs + oc] + v : v;
    d.out[b * d.out_bs + blk * d.out_cbs + oc] = r;
    return;
  }
  switch (d.epi) {
    case ACCFLOW_EPI_RES_RELU: d.out[b * d.out_bs + o] = fmaxf(d.e0[b * d.e0_bs + o] + v, 0.0f); break;
    case ACCFLOW_EPI_GRU_ZR:
      if (ch < half) d.out[b * d.out_bs + o] = v;
      else d.out2[b * d.out2_bs + o - (long long)half * OHW] = v * d.e0[b * d.e0_bs + o - (long long)half * OHW];
      break;
    case ACCFLOW_EPI_GRU_Q: {
      const float z = d.e1[b * d.e1_bs + o], h = d.e0[b * d.e0_bs + o];
      d.out[b * d.out_bs + o] = gru_blend(z, h, v);
    } break;
    case ACCFLOW_EPI_ACCUM: d.out[b * d.out_bs + o] = d.e0[b * d.e0_bs + o] + v; break;
    default: d.out[b * d.out_bs + o] = v;
  }
}


// out = epilogue(act(sum_z part[z] + bias)): the K-parts are added in the fixed order z = 0, 1, ... (deterministic).
// Thread = (batch item, octet of 8 output channels, pixel): the fp32 stores of the 8 channels are 8 coalesced stores
// along the pixels, and the 8 results are exactly one chunk of the S16 copy (accflow_conv_desc.out16) when requested.
__global__ __launch_bounds__(256) void conv_ksplit_reduce_s16_kernel(const accflow_conv_desc d, int Z) {
  const int OHW = d.OH * d.OW;
  const int O = (d.Cout + 7) >> 3;
  const long long n = (long long)d.B * d.Cout * OHW;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)d.B * O * OHW) return;
  const int px = (int)(i % OHW);
  const int oct = (int)((i / OHW) % O);
  const int b = (int)(i / ((long long)OHW * O));
  const int half = d.Cout >> 1;
  const bool zr = d.epi == ACCFLOW_EPI_GRU_ZR;
  float res[8];
  bool in16[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = oct * 8 + j;
    res[j] = 0.0f;
    in16[j] = false;
    if (ch >= d.Cout) continue;
    const long long e = ((long long)b * d.Cout + ch) * OHW + px;
    float v = d.kws[e];
    for (int z = 1; z < Z; ++z) v += d.kws[(long long)z * n + e];
    v = fmaf(v, d.wscale16 ? d.wscale16[(d.wsplit_bs ? (long long)b * d.CoutPad : 0) + ch] : 1.0f, d.bias ? d.bias[ch] : 0.0f);
    if (d.pre && (zr || d.epi == ACCFLOW_EPI_GRU_Q)) v += d.pre[b * d.pre_bs + (long long)ch * OHW + px];
    v = apply_act(v, d.act);
    const long long o = (long long)ch * OHW + px;
    float r = v;
    if (d.cb) {  // channel-block scatter (STORE / ACCUM)
      const int blk = ch / d.cb;
      const long long oc = (long long)(ch - blk * d.cb) * OHW + px;
      r = d.epi == ACCFLOW_EPI_ACCUM ? d.e0[b * d.e0_bs + blk * d.e0_cbs + oc] + v : v;
      if (d.out) d.out[b * d.out_bs + blk * d.out_cbs + oc] = r;
      res[j] = r;
      in16[j] = true;
      continue;
    }
    switch (d.epi) {
      case ACCFLOW_EPI_RES_RELU: r = fmaxf(d.e0[b * d.e0_bs + o] + v, 0.0f); break;
      case ACCFLOW_EPI_GRU_ZR:
        if (ch >= half) r = v * d.e0[b * d.e0_bs + o - (long long)half * OHW];
        break;
      case ACCFLOW_EPI_GRU_Q: {
        const float z = d.e1[b * d.e1_bs + o], h = d.e0[b * d.e0_bs + o];
        r = gru_blend(z, h, v);
      } break;
      case ACCFLOW_EPI_ACCUM: r = d.e0[b * d.e0_bs + o] + v; break;
      default: break;
    }
    if (zr && ch >= half) {
      if (d.out2) d.out2[b * d.out2_bs + o - (long long)half * OHW] = r;
    } else if (d.out) {
      d.out[b * d.out_bs + o] = r;
    }
    res[j] = r;
    in16[j] = !zr || ch >= half;
  }
  if (!d.out16) return;
  // S16 copy: GRU_ZR -> the r*h channels (octets counted from Cout/2, which is a multiple of 8), else all channels
  const int c16 = oct * 8 - (zr ? half : 0);
  if (c16 < 0) return;
  const int n16 = zr ? half : d.Cout;
  const int nvalid = min(8, ((n16 + 1) & ~1) - c16);   // whole channel pairs only (accflow_conv_desc.out16)
  constexpr float ASC = (float)(1 << ACCFLOW_F16_ASHIFT);
  unsigned hi[4], lo[4];
  bool bad = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float a = in16[2 * k] ? res[2 * k] * ASC : 0.0f, c = in16[2 * k + 1] ? res[2 * k + 1] * ASC : 0.0f;
    bad |= !(fabsf(a) < 65520.0f) | !(fabsf(c) < 65520.0f);
    const f32x2 v2 = {a, c};
    const f16x2 hq = __builtin_convertvector(v2, f16x2);
    const f32x2 back = __builtin_convertvector(hq, f32x2);
    const f32x2 rest = {a - back[0], c - back[1]};
    const f16x2 lq = __builtin_convertvector(rest, f16x2);
    hi[k] = __builtin_bit_cast(unsigned, hq);
    lo[k] = __builtin_bit_cast(unsigned, lq);
  }
  unsigned* base = reinterpret_cast<unsigned*>(d.out16) + b * d.out16_bs;
  int o16 = c16 >> 3;
  if (d.cb) {
    base += (long long)(c16 / d.cb) * d.out16_cbs;
    o16 = (c16 % d.cb) >> 3;
  }
  unsigned* ph = base + (((long long)(o16 * 2 + 0)) * OHW + px) * 4;
  unsigned* pl = base + (((long long)(o16 * 2 + 1)) * OHW + px) * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (2 * k < nvalid) { ph[k] = hi[k]; pl[k] = lo[k]; }
  if (bad && d.guard) atomicOr(d.guard, 1);
}

// The same for the common case of the batch-1 fusion chain on pre-split tensors (and the GMA aggregation: per-item row scales,
// channel-block scatter) - every octet complete, not the GRU_ZR form - with a thread per (batch item, 4-channel group, pixel): half the serial loads per thread and
// twice the threads of the octet form above (480 workgroups for 128 channels x 7 680 pixels left most CUs idle: 15 us per
// launch, 0.9 ms per sequence for the chain's 60 reduces, profiles/r04_kernel_stats_bench_1stream.txt).  A lane writes its 4
// channels' 8 bytes into each term's chunk, the conv epilogue's store pattern.
__global__ __launch_bounds__(256) void conv_ksplit_reduce_s16q_kernel(const accflow_conv_desc d, int Z) {
  const int OHW = d.OH * d.OW;
  const int Q = d.Cout >> 2;
  const long long n = (long long)d.B * d.Cout * OHW;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)d.B * Q * OHW) return;
  const int px = (int)(i % OHW);
  const int q = (int)((i / OHW) % Q);
  const int b = (int)(i / ((long long)OHW * Q));
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = d.kws[((long long)b * d.Cout + q * 4 + j) * OHW + px];
  for (int z = 1; z < Z; ++z)
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += d.kws[(long long)z * n + ((long long)b * d.Cout + q * 4 + j) * OHW + px];
  float res[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = q * 4 + j;
    float t = fmaf(v[j], d.wscale16 ? d.wscale16[(d.wsplit_bs ? (long long)b * d.CoutPad : 0) + ch] : 1.0f, d.bias ? d.bias[ch] : 0.0f);
    if (d.pre && d.epi == ACCFLOW_EPI_GRU_Q) t += d.pre[b * d.pre_bs + (long long)ch * OHW + px];
    t = apply_act(t, d.act);
    const long long o = (long long)ch * OHW + px;
    if (d.cb) {  // channel-block scatter (STORE / ACCUM; cb % 32 == 0, so the 4 channels share a block): accflow_conv_desc.cb
      const int blk = ch / d.cb;
      const long long oc = (long long)(ch - blk * d.cb) * OHW + px;
      if (d.epi == ACCFLOW_EPI_ACCUM) t = d.e0[b * d.e0_bs + blk * d.e0_cbs + oc] + t;
      if (d.out) d.out[b * d.out_bs + blk * d.out_cbs + oc] = t;
      res[j] = t;
      continue;
    }
    switch (d.epi) {
      case ACCFLOW_EPI_RES_RELU: t = fmaxf(d.e0[b * d.e0_bs + o] + t, 0.0f); break;
      case ACCFLOW_EPI_GRU_Q: {
        const float zz = d.e1[b * d.e1_bs + o], h = d.e0[b * d.e0_bs + o];
        t = gru_blend(zz, h, t);
      } break;
      case ACCFLOW_EPI_ACCUM: t = d.e0[b * d.e0_bs + o] + t; break;
      default: break;
    }
    if (d.out) d.out[b * d.out_bs + o] = t;
    res[j] = t;
  }
  constexpr float ASC = (float)(1 << ACCFLOW_F16_ASHIFT);
  unsigned hi[2], lo[2];
  bool bad = false;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float a = res[2 * k] * ASC, c = res[2 * k + 1] * ASC;
    bad |= !(fabsf(a) < 65520.0f) | !(fabsf(c) < 65520.0f);
    const f32x2 v2 = {a, c};
    const f16x2 hq = __builtin_convertvector(v2, f16x2);
    const f32x2 back = __builtin_convertvector(hq, f32x2);
    const f32x2 rest = {a - back[0], c - back[1]};
    const f16x2 lq = __builtin_convertvector(rest, f16x2);
    hi[k] = __builtin_bit_cast(unsigned, hq);
    lo[k] = __builtin_bit_cast(unsigned, lq);
  }
  unsigned* base = reinterpret_cast<unsigned*>(d.out16) + b * d.out16_bs;
  int oct = q >> 1;
  const int w0 = (q & 1) * 2;
  if (d.cb) {
    base += (long long)((q * 4) / d.cb) * d.out16_cbs;
    oct = ((q * 4) % d.cb) >> 3;
  }
  unsigned* ph = base + (((long long)(oct * 2 + 0)) * OHW + px) * 4 + w0;
  unsigned* pl = base + (((long long)(oct * 2 + 1)) * OHW + px) * 4 + w0;
  typedef unsigned u2_ __attribute__((ext_vector_type(2)));
  *reinterpret_cast<u2_*>(ph) = u2_{hi[0], hi[1]};
  *reinterpret_cast<u2_*>(pl) = u2_{lo[0], lo[1]};
  if (bad && d.guard) atomicOr(d.guard, 1);
}

}  // namespace

int conv_ksplit_reduce_launch(const accflow_conv_desc& d, int Z, hipStream_t st) {
  if (d.out16 && !(d.Cout & 7) && !(d.cb & 7) && d.epi != ACCFLOW_EPI_GRU_ZR) {
    const long long nthr = (long long)d.B * (d.Cout / 4) * d.OH * d.OW;
    hipLaunchKernelGGL(conv_ksplit_reduce_s16q_kernel, dim3(cdiv(nthr, 256)), dim3(256), 0, st, d, Z);
  } else if (d.out16) {
    const long long nthr = (long long)d.B * ((d.Cout + 7) / 8) * d.OH * d.OW;
    hipLaunchKernelGGL(conv_ksplit_reduce_s16_kernel, dim3(cdiv(nthr, 256)), dim3(256), 0, st, d, Z);
  } else {
    const long long nout = (long long)d.B * d.Cout * d.OH * d.OW;
    hipLaunchKernelGGL(conv_ksplit_reduce_kernel, dim3(cdiv(nout, 256)), dim3(256), 0, st, d, Z);
  }
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

namespace {

template <int TC>
int launch_conv_direct(const accflow_conv_desc& d, hipStream_t st) {
  const int tiles = cdiv(d.OW, DIR_TW) * cdiv(d.OH, DIR_TH);
  const long long nb = (long long)d.B * tiles * cdiv(d.Cout, 2 * TC * 32);
  // split-K for grids that leave most of the 256 CUs idle (the batch-1 fusion chain): 2-4 parts of >= 2 chunks
  static const bool w4 = [] { const char* e = getenv("ACCFLOW_DIRECT_W4"); return !e || atoi(e) != 0; }();
  // normalise-on-load (desc.in_norm): single-source input of <= 256 channels, the kernels instantiated for it
  const bool norm_ok = !d.in1 && d.C0 <= DIR_NORM_MAXC && (TC == 1 || w4);
  if (accflow_tls_dry_route) *accflow_tls_dry_route = norm_ok ? 1 : 0;
  ACCFLOW_DRY_RUN(tiles * ((TC == 2 && w4) ? 1 : 2));  // one slot per wave along the pixels
  if (d.in_norm && !norm_ok) return 1;
  int Z = 1;
  const long long nout = (long long)d.B * d.Cout * d.OH * d.OW;
  const int nchunk = (d.C0 + d.C1 + 15) / 16;
  if (d.kws && nchunk >= 256 && nb < 700 && !d.stats) {
    // very deep reductions (GMA aggregation: 14 400 channels at 720x1280): the part count that best fills whole
    // rounds of the 768 workgroup slots
    static const int zenv = [] { const char* e = getenv("ACCFLOW_DIRECT_KSPLIT"); return e ? atoi(e) : 0; }();
    double best = 0.0;
    for (int z = 1; z <= 8; ++z) {
      const double fill = (double)(nb * z) / (double)(cdiv(nb * z, 768) * 768);
      if (fill > best + 0.03) { best = fill; Z = z; }
    }
    if (zenv > 0) Z = zenv;
  } else if (d.kws && nb < 320 && !d.stats) {
    // (a fill-based part count up to 8 here, too, measured no gain on the batch-1 fusion chain: 30.9 vs 31.0 ms)
    Z = (int)((512 + nb - 1) / nb);
    if (Z > 4) Z = 4;
    if (Z > nchunk / 2) Z = nchunk / 2;
  }
  if (Z > 1 && (long long)Z * nout > d.kws_elems) Z = (int)(d.kws_elems / nout);
  if (Z < 1) Z = 1;
  if (d.out16 && d.epi == ACCFLOW_EPI_GRU_ZR && ((d.Cout >> 1) & 7)) Z = 1;
  dim3 grid((unsigned)((long long)d.B * tiles), cdiv(d.Cout, 2 * TC * 32), Z);
  // ACCFLOW_DIRECT_W4=0 selects the 2 x 2 wave layout of the 128-channel kernel (A/B measurements)
  constexpr bool CAN_W4 = TC == 2;
  const bool f16 = d.mode == ACCFLOW_CONV_F16X3 && d.wpatch16;
  int rc;
  if (d.epi == ACCFLOW_EPI_TAPGEMM) {   // (validated by accflow_conv2d_f32)
    if (TC != 2 || !f16 || !d.in_fmt || Z != 1 || !(CAN_W4 && w4)) return 1;
    static const bool kt9_tg = [] { const char* e = getenv("ACCFLOW_DIRECT_KT9"); return !e || atoi(e) != 0; }();
    rc = (kt9_tg && d.KH == 3 && d.KW == 3 && (d.C0 + d.C1) % 16 == 0) ? accflow_direct_launch_s16k9(d, 2, true, grid, st)
                                                                        : accflow_direct_launch_s16tg(d, grid, st);
  } else if (d.in_fmt) {  // S16 sources: the fp16 kernel with the DMA loader (every source must be S16; no normalise-on-load)
    if (!f16 || d.in_norm || d.in_fmt != (d.in1 ? 3 : 1)) return 1;
    // the tap-specialised K loop of the 128-channel kernel for 5-tap convolutions (the GRU's 1x5 / 5x1; ACCFLOW_DIRECT_KT=0: the
    // generic loop, A/B runs): 23.80 / 23.72 vs 23.91 / 23.98 ms per step on one box
    static const bool kt_on = [] { const char* e = getenv("ACCFLOW_DIRECT_KT"); return !e || atoi(e) != 0; }();
    // ... and for 3x3 (9 taps; round 6, ACCFLOW_DIRECT_KT9=0: the generic loop): 23.27-23.31 vs 23.44-23.54 ms per step
    static const bool kt9_on = [] { const char* e = getenv("ACCFLOW_DIRECT_KT9"); return !e || atoi(e) != 0; }();
    if (kt_on && TC == 2 && CAN_W4 && w4 && d.KH * d.KW == 5) rc = accflow_direct_launch_s16k(d, 5, grid, st);
    else if (kt9_on && (TC == 1 || (CAN_W4 && w4)) && d.KH == 3 && d.KW == 3) rc = accflow_direct_launch_s16k9(d, TC, false, grid, st);
    else rc = accflow_direct_launch_s16(d, TC, grid, st);
  } else if (d.in_norm) {
    rc = f16 ? accflow_direct_launch_f16_norm(d, TC, grid, st)
             : accflow_direct_launch_bf16_norm(d, TC, d.mode == ACCFLOW_CONV_BF16X3 ? 2 : 3, grid, st);
  } else {
    rc = f16 ? accflow_direct_launch_f16(d, TC, CAN_W4 && w4, grid, st)
             : accflow_direct_launch_bf16(d, TC, d.mode == ACCFLOW_CONV_BF16X3 ? 2 : 3, CAN_W4 && w4, grid, st);
  }
  if (rc) return rc;
  if (Z > 1) return conv_ksplit_reduce_launch(d, Z, st);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

}  // namespace

// Level 0 of the displaced correlation pyramid for one pair: E_0 = shear(F1^T F2 / sqrt(C)) with BOTH feature maps
// pre-split into [term][k/8][pixel][8] packs (d.wpatch / d.wpatch16 = fmap1 incl. the 1/sqrt(C) scale, d.in0 reused as
// the pack of fmap2), so that every MFMA fragment is one 16-byte buffer load straight from L2: no LDS, no barrier and no
// operand split in the K loop (the 1x1-convolution form of the direct kernel re-split the same fmap2 tile in each of
// the P/128 workgroups of a tile column and spent 2300 cycles per 16-deep step, 384 of them in its 12 MFMAs).
// Workgroup = 128 query pixels (rows) x 128 target pixels (columns): TWO image rows (2*yo, 2*yo + 1) x 64 columns
// (xc*64 ..), so that the tile holds whole 2x2 pooling cells and level 1 is emitted with level 0 (corr_disp_store2;
// d.out2 = this pair's level 1); LDS only for the displaced store.
template <int NT, bool F16>
__global__ __launch_bounds__(256, 2) void corr_disp_gemm_kernel(const accflow_conv_desc d) {
  constexpr int TC = 2, TP = 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[DISP2_LDS_BYTES];
#ifdef ACCFLOW_KPROF
  const unsigned long long tL0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave >> 1, wp = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int P = d.OH * d.OW;
  // XCD-aware tile order.  Workgroups b and b + 8 share an XCD (and its 4 MB L2): XCD j owns the query-pixel blocks
  // j, j + 8, ... and walks the target tiles with its own query blocks innermost, so the ~96 workgroups an XCD runs
  // at a time share <= 8 A tiles and ~12 B tiles (128 KB each).  In plain row-major order every workgroup streamed
  // its 256 KB of operands from beyond L2: 920 MB of reads per pair for 236 MB of output, 178 us per pair.
  const int npb = (P + 127) >> 7, percol = (npb + 7) >> 3;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int pb = xcd + 8 * (seq % percol), qt = seq / percol;
  if (pb >= npb) return;
  const int ncx = (d.OW + 63) >> 6;
  const int yo = qt / ncx, xc = qt - yo * ncx;
  const int cblk0 = pb * 128;
  const int nstep = d.Kpad / 16;
  const long long step_bytes = 2LL * d.CoutPad * 16, term_bytes = (long long)(d.Kpad / 8) * d.CoutPad * 16;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(F16 ? d.wpatch16 : d.wpatch), 0, (int)(unsigned)(3 * term_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.in0), 0, (int)(unsigned)(3 * term_bytes), 0x00020000);
  const unsigned avoff = (unsigned)((kh * d.CoutPad + cblk0 + wc * 64 + l31) * 16);
  // this wave's target pixels: row 2*yo + wp, columns xc*64 + i*32 + l31 (outside the image: masked, reads 0)
  unsigned bvoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int y2 = 2 * yo + wp, x2 = xc * 64 + i * 32 + l31;
    bvoff[i] = (y2 < d.OH && x2 < d.OW) ? (unsigned)((kh * d.CoutPad + y2 * d.OW + x2) * 16) : 0xFFFFFFFFu;
  }
#define CG_LOAD(STEP, A, Bf)                                                                                     \
  _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int i = 0; i < 2; ++i) {                 \
    const int so = (int)(unsigned)(t * term_bytes + (STEP) * step_bytes);                                        \
    A[t][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(avoff + i * 512), so, 0)); \
    Bf[t][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rb, (int)bvoff[i], so, 0));      \
  }
  f32x16 acc[TC][TP];
#pragma unroll
  for (int tc = 0; tc < TC; ++tc)
#pragma unroll
    for (int tp = 0; tp < TP; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tc][tp][r] = 0.0f;
  // fragments are requested PD steps ahead (an L2 round trip is longer than one 12-MFMA step): 2 with two terms
  // (three register sets, still 3 workgroups per CU), 1 with three terms
  constexpr int PD = NT == 2 ? 2 : 1;
  bf16x8 aA[NT][TC], bA[NT][TP], aB[NT][TC], bB[NT][TP], aC[NT][TC], bC[NT][TP];
  CG_LOAD(0, aA, bA);
  if (PD == 2 && nstep > 1) { CG_LOAD(1, aB, bB); }
#define CG_STEP(STEP, AC, BC, AN, BN)                                                                            \
  do {                                                                                                           \
    if ((STEP) + PD < nstep) { CG_LOAD((STEP) + PD, AN, BN); }                                                   \
    constexpr int NPAIR = NT == 3 ? 6 : 3;                                                                       \
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};                                        \
    _Pragma("unroll") for (int pr = 6 - NPAIR; pr < 6; ++pr) _Pragma("unroll") for (int tc = 0; tc < TC; ++tc)   \
        _Pragma("unroll") for (int tp = 0; tp < TP; ++tp) acc[tc][tp] =                                          \
            dir_mfma<F16>(AC[PA[pr]][tc], BC[PB[pr]][tp], acc[tc][tp]);                                          \
  } while (0)
  if constexpr (PD == 2) {
    for (int step = 0; step < nstep; step += 3) {
      CG_STEP(step, aA, bA, aC, bC);
      if (step + 1 < nstep) CG_STEP(step + 1, aB, bB, aA, bA);
      if (step + 2 < nstep) CG_STEP(step + 2, aC, bC, aB, bB);
    }
  } else {
    for (int step = 0; step < nstep; step += 2) {
      CG_STEP(step, aA, bA, aB, bB);
      if (step + 1 < nstep) CG_STEP(step + 1, aB, bB, aA, bA);
    }
  }
#undef CG_STEP
#undef CG_LOAD
#ifdef ACCFLOW_KPROF
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long tL1 = __builtin_amdgcn_s_memrealtime();
#endif
  corr_disp_store2(d, acc, reinterpret_cast<float*>(smem), reinterpret_cast<int*>(smem) + 64 * DISP_PITCH,
                   reinterpret_cast<float*>(smem + DISP_LDS_BYTES), d.out2, cblk0, yo, xc, wc, wp, lane, wave, tid);
#ifdef ACCFLOW_KPROF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) {
    const int slot = (blockIdx.x & 4095) * 16;
    g_kprof[slot + 8] = tL1 - tL0;
    g_kprof[slot + 11] = __builtin_amdgcn_s_memrealtime() - tL0;
    g_kprof[slot + 10] = 1;
    g_kprof[slot + 14] = 0;
  }
#endif
}

// GMA attention (gma/modules.py:54-76, heads = 1) straight into the pre-split S16 form, in two passes of ONE register-only
// GEMM over the fp16 hi/lo packs of k (rows j) and q (columns i) - K = 128 deep, so the tile (128 x 128) is recomputed
// rather than stored:
//   PASS 0: per column i the online-softmax pair (max, sum of exponentials) over this workgroup's 128 rows -> partial
//           statistics part[row block][i][2]; no logits ever reach HBM;
//   PASS 1: the same tile again, e = exp(x - m_i) / tot_i with the FINAL column statistics (gma_attn_stats_kernel), written
//           as S16 chunks: rows j are the "channels" of the aggregation GEMM's activation operand, and 4 consecutive
//           accumulator registers are 4 consecutive rows of one octet - the conv epilogue's S16 store pattern.
// Replaces a bf16x6 im2col GEMM that stored 4 P^2 bytes of logits plus a 2-read / 1-write softmax over them (1.55 ms per
// image1 at 90 x 160) by ~P^2 * 4 bytes of writes in all.
struct gma_attn_args {
  const void* kpack; const void* qpack;   // [2 terms][D/8][Ppad][8] fp16 (conv_pack_kmajor_kernel), Ppad = coutpad(P)
  float* part;                            // [nrb][P][2]
  const float* stats;                     // [P][2] = {max, 1 / sum}
  unsigned* out16;                        // S16 (1, P channels j, P pixels i)
  float acc_scale;                        // logits = acc * acc_scale
  int P, D, Ppad;
};

template <int PASS>
__global__ __launch_bounds__(256, 2) void gma_attn_gemm_kernel(const gma_attn_args a) {
  constexpr int TC = 2, TP = 2;
  __shared__ float redm[2][128], reds[2][128];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave >> 1, wp = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int nb = (a.P + 127) >> 7;
  const int rb = blockIdx.x / nb, cbk = blockIdx.x - rb * nb;   // row block (j), column block (i): columns fastest
  const int j0 = rb * 128, i0 = cbk * 128;
  const int nstep = a.D / 16;
  const long long step_bytes = 2LL * a.Ppad * 16, term_bytes = (long long)(a.D / 8) * a.Ppad * 16;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.kpack), 0, (int)(unsigned)(2 * term_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.qpack), 0, (int)(unsigned)(2 * term_bytes), 0x00020000);
  const unsigned avoff = (unsigned)((kh * a.Ppad + j0 + wc * 64 + l31) * 16);
  const unsigned bvoff = (unsigned)((kh * a.Ppad + i0 + wp * 64 + l31) * 16);
  f32x16 acc[TC][TP];
#pragma unroll
  for (int tc = 0; tc < TC; ++tc)
#pragma unroll
    for (int tp = 0; tp < TP; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tc][tp][r] = 0.0f;
  for (int step = 0; step < nstep; ++step) {
    bf16x8 A[2][TC], Bf[2][TP];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int so = (int)(unsigned)(t * term_bytes + step * step_bytes);
        A[t][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(avoff + i * 512), so, 0));
        Bf[t][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rq, (int)(bvoff + i * 512), so, 0));
      }
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
      for (int tc = 0; tc < TC; ++tc)
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) acc[tc][tp] = dir_mfma<true>(A[PA[pr]][tc], Bf[PB[pr]][tp], acc[tc][tp]);
  }
  const int lh4 = kh * 4;
  if constexpr (PASS == 0) {
    // column statistics over this wave's 64 rows (registers, then the other half-wave), then over the two row halves (LDS)
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) {
      float m = -INFINITY;
#pragma unroll
      for (int tc = 0; tc < TC; ++tc)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = j0 + wc * 64 + tc * 32 + (r & 3) + 8 * (r >> 2) + lh4;
          if (j < a.P) m = fmaxf(m, acc[tc][tp][r] * a.acc_scale);
        }
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float s = 0.0f;
#pragma unroll
      for (int tc = 0; tc < TC; ++tc)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = j0 + wc * 64 + tc * 32 + (r & 3) + 8 * (r >> 2) + lh4;
          if (j < a.P) s += expf(acc[tc][tp][r] * a.acc_scale - m);
        }
      s += __shfl_xor(s, 32, 64);
      if (kh == 0) {
        redm[wc][wp * 64 + tp * 32 + l31] = m;
        reds[wc][wp * 64 + tp * 32 + l31] = s;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int i = i0 + tid;
      if (i < a.P) {
        const float m0 = redm[0][tid], m1 = redm[1][tid];
        const float m = fmaxf(m0, m1);
        float s = 0.0f;
        if (m0 > -INFINITY) s += reds[0][tid] * expf(m0 - m);
        if (m1 > -INFINITY) s += reds[1][tid] * expf(m1 - m);
        float* p = a.part + ((long long)rb * a.P + i) * 2;
        p[0] = m; p[1] = s;
      }
    }
  } else {
    const int O = (a.P + 7) >> 3;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out16, 0, (int)(unsigned)((long long)O * 2 * a.P * 16), 0x00020000);
    constexpr float ASC = (float)(1 << ACCFLOW_F16_ASHIFT);
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) {
      const int i = i0 + wp * 64 + tp * 32 + l31;
      const bool iok = i < a.P;
      const float m = iok ? a.stats[2 * i] : 0.0f, inv = iok ? a.stats[2 * i + 1] * ASC : 0.0f;
      const unsigned vo = iok ? (unsigned)(i * 16 + lh4 * 2) : 0xFFFFFFFFu;
#pragma unroll
      for (int tc = 0; tc < TC; ++tc)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int jb = j0 + wc * 64 + tc * 32 + 8 * g;   // first row of the octet (wave-uniform); this lane: jb + lh4 ..+3
          const int oct = jb >> 3;
          if (oct >= O) continue;
          float e[4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            e[q] = (jb + lh4 + q < a.P) ? expf(acc[tc][tp][4 * g + q] * a.acc_scale - m) * inv : 0.0f;
          unsigned hi2[2], lo2[2];
#pragma unroll
          for (int k = 0; k < 2; ++k) {   // values in [0, 16]
            const f32x2 v2 = {e[2 * k], e[2 * k + 1]};
            const f16x2 hq = __builtin_convertvector(v2, f16x2);
            const f32x2 back = __builtin_convertvector(hq, f32x2);
            const f32x2 rest = {v2[0] - back[0], v2[1] - back[1]};
            const f16x2 lq = __builtin_convertvector(rest, f16x2);
            hi2[k] = __builtin_bit_cast(unsigned, hq);
            lo2[k] = __builtin_bit_cast(unsigned, lq);
          }
          typedef unsigned u2 __attribute__((ext_vector_type(2)));
          const u2 hv = {hi2[0], hi2[1]}, lv = {lo2[0], lo2[1]};
          const int so = (int)(unsigned)((long long)oct * 2 * a.P * 16);
          __builtin_amdgcn_raw_buffer_store_b64(hv, ro, (int)vo, so, 0);
          __builtin_amdgcn_raw_buffer_store_b64(lv, ro, (int)vo, so + a.P * 16, 0);
        }
    }
  }
}

// final column statistics {max, 1 / sum} from the per-row-block partials, combined in the fixed order of the blocks
__global__ __launch_bounds__(256) void gma_attn_stats_kernel(const float* __restrict__ part, float* __restrict__ stats, int P, int nrb) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  float m = -INFINITY;
  for (int r = 0; r < nrb; ++r) m = fmaxf(m, part[((long long)r * P + i) * 2]);
  float s = 0.0f;
  for (int r = 0; r < nrb; ++r) {
    const float mr = part[((long long)r * P + i) * 2];
    if (mr > -INFINITY) s += part[((long long)r * P + i) * 2 + 1] * expf(mr - m);
  }
  stats[2 * i] = m;
  stats[2 * i + 1] = 1.0f / s;
}

int accflow_launch_gma_attn(const void* kpack, const void* qpack, float* part, float* stats, void* out16, float acc_scale, int P,
                            int D, int Ppad, hipStream_t st) {
  gma_attn_args a;
  a.kpack = kpack; a.qpack = qpack; a.part = part; a.stats = stats; a.out16 = reinterpret_cast<unsigned*>(out16);
  a.acc_scale = acc_scale; a.P = P; a.D = D; a.Ppad = Ppad;
  const int nb = cdiv(P, 128);
  hipLaunchKernelGGL((gma_attn_gemm_kernel<0>), dim3(nb * nb), dim3(256), 0, st, a);
  hipLaunchKernelGGL(gma_attn_stats_kernel, dim3(cdiv(P, 256)), dim3(256), 0, st, part, stats, P, nb);
  hipLaunchKernelGGL((gma_attn_gemm_kernel<1>), dim3(nb * nb), dim3(256), 0, st, a);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

// LDS-ring form of corr_disp_gemm_kernel (fp16 split, two terms).  The register-only loop above lets every wave pull its
// own A and B fragments from L2 - 8 KB per 12 MFMAs - and is bound by the vector memory pipe, not by the matrix cores or
// the stores (profiles/r03_corr_gemm_store_ab.txt).  Here each operand byte enters the CU once per WORKGROUP: a 16-deep
// step's A tile (128 query pixels) and B tile (2 image rows x 64 target pixels), 2 terms x 2 octets x 128 x 16 B = 8 KB
// each, are DMA'd (buffer_load_dwordx4 ... lds, 16 one-KB pieces per step, 4 per wave) into a 3-slot LDS ring two steps
// ahead; the waves read their fragments with conflict-free ds_read_b128 (32 lanes = 512 contiguous bytes).  Half the
// vector-memory traffic, none of it through VGPRs.  One barrier per step: it publishes step s (every wave has waited for
// ITS pieces with a counted vmcnt) and retires step s-1's reads, whose slot the DMA of step s+2 then overwrites.
// The ring memory is reused by the displaced store afterwards.
// (A 16-byte-store epilogue with a 2-slot ring and 4 workgroups per CU was built, measured and removed again: no faster -
// profiles/r03_corr_gemm_store_ab.txt; in-kernel stamps, tools/kprof_corr.py: K loop 13.8 us, store phase 8.5 us.)
constexpr int CRING_SLOT_CHUNKS = 2 * 2 * 2 * 128;   // [A|B][term][octet][128] 16-byte chunks
// measurement builds only (tools/corr_ablation.sh, -DACCFLOW_CORR_ABL=bits): 1 no B-fragment reads (A's registers reused),
// 2 no MFMAs, 4 no operand DMA (stale LDS), 8 no displaced store, 16 no A-fragment reads either
#ifndef ACCFLOW_CORR_ABL
#define ACCFLOW_CORR_ABL 0
#endif
__global__ __launch_bounds__(256, 2) void corr_disp_ring_kernel(const accflow_conv_desc d) {
  constexpr int TC = 2, TP = 2;
  constexpr int CRING_SLOTS = 3;
  constexpr int RING_BYTES = CRING_SLOTS * CRING_SLOT_CHUNKS * 16;
  constexpr int ST_BYTES = DISP2_LDS_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING_BYTES > ST_BYTES ? RING_BYTES : ST_BYTES];
  u32x4* ring = reinterpret_cast<u32x4*>(smem);
#ifdef ACCFLOW_KPROF
  const unsigned long long tL0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave >> 1, wp = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int P = d.OH * d.OW;
  const int npb = (P + 127) >> 7, percol = (npb + 7) >> 3;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int pb = xcd + 8 * (seq % percol), qt = seq / percol;
  if (pb >= npb) return;
  const int ncx = (d.OW + 63) >> 6;
  const int yo = qt / ncx, xc = qt - yo * ncx;
  const int cblk0 = pb * 128;
  const int nstep = d.Kpad / 16;
  const unsigned oct_bytes = (unsigned)d.CoutPad * 16u, term_bytes = (unsigned)(d.Kpad / 8) * oct_bytes;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.wpatch16), 0, (int)(2 * term_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.in0), 0, (int)(2 * term_bytes), 0x00020000);
  // DMA piece i of this wave: piece id = wave*4 + i -> operand (A: 0-7, B: 8-15), (term, octet) row, half of the 128 items
  unsigned pvoff[4];
  int prow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = wave * 4 + i;
    const int isb = id >> 3, row = (id >> 1) & 3, half = id & 1;
    prow[i] = row;
    if (!isb) {
      pvoff[i] = (unsigned)(cblk0 + half * 64 + lane) * 16u;       // (rows beyond P are zero in the pack)
    } else {
      const int y2 = 2 * yo + half, x2 = xc * 64 + lane;           // B: image row `half` of the tile
      pvoff[i] = (y2 < d.OH && x2 < d.OW) ? (unsigned)(y2 * d.OW + x2) * 16u : 0xFFFFFFFFu;
    }
  }
  auto issue = [&](int step) {
    const int slot = step % CRING_SLOTS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = wave * 4 + i;
      const int isb = id >> 3, row = prow[i], half = id & 1;
      const int t = row >> 1, o = row & 1;
      const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)t * term_bytes + (unsigned)(2 * step + o) * oct_bytes));
      const int dst = __builtin_amdgcn_readfirstlane(slot * CRING_SLOT_CHUNKS + isb * 512 + row * 128 + half * 64);
      if (isb)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)&ring[dst], 16, (int)pvoff[i], (int)soff, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)&ring[dst], 16, (int)pvoff[i], (int)soff, 0, 0);
    }
  };
  f32x16 acc[TC][TP];
#pragma unroll
  for (int tc = 0; tc < TC; ++tc)
#pragma unroll
    for (int tp = 0; tp < TP; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tc][tp][r] = 0.0f;
  constexpr int AHEAD = CRING_SLOTS - 1;        // steps of DMA in flight beyond the one being consumed
  issue(0);
  if (AHEAD > 1 && nstep > 1) issue(1);
  bf16x8 A[2][TC], Bf[2][TP];
  for (int step = 0; step < nstep; ++step) {
    // this wave's 4 pieces of `step` have landed once at most the pieces of the later steps are still in flight
    if (AHEAD > 1 && step + 1 < nstep) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (step + AHEAD < nstep && !(ACCFLOW_CORR_ABL & 4)) issue(step + AHEAD);   // into the slot whose reads (step - 1) every wave finished before the barrier
    const u32x4* sl = ring + (step % CRING_SLOTS) * CRING_SLOT_CHUNKS;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (!(ACCFLOW_CORR_ABL & 16) || step == 0) A[t][i] = __builtin_bit_cast(bf16x8, sl[(t * 2 + kh) * 128 + wc * 64 + i * 32 + l31]);
        if (ACCFLOW_CORR_ABL & 1) Bf[t][i] = A[t][i];
        else Bf[t][i] = __builtin_bit_cast(bf16x8, sl[512 + (t * 2 + kh) * 128 + wp * 64 + i * 32 + l31]);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
      for (int tc = 0; tc < TC; ++tc)
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) {
          if (ACCFLOW_CORR_ABL & 2) acc[tc][tp][pr] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, A[PA[pr]][tc])[0] ^ __builtin_bit_cast(u32x4, Bf[PB[pr]][tp])[1]);
          else acc[tc][tp] = dir_mfma<true>(A[PA[pr]][tc], Bf[PB[pr]][tp], acc[tc][tp]);
        }
  }
  __syncthreads();   // the ring is dead: its memory becomes the displaced store's staging tile
  if ((ACCFLOW_CORR_ABL & 8) && acc[0][0][0] != 12345.678f) return;
#ifdef ACCFLOW_KPROF
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long tL1 = __builtin_amdgcn_s_memrealtime();
#endif
  corr_disp_store2(d, acc, reinterpret_cast<float*>(smem), reinterpret_cast<int*>(smem) + 64 * DISP_PITCH,
                   reinterpret_cast<float*>(smem + DISP_LDS_BYTES), d.out2, cblk0, yo, xc, wc, wp, lane, wave, tid);
#ifdef ACCFLOW_KPROF
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long tL2 = __builtin_amdgcn_s_memrealtime();   // store instructions issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) {
    const int slot = (blockIdx.x & 4095) * 16;
    g_kprof[slot + 8] = tL1 - tL0;
    g_kprof[slot + 9] = tL2 - tL1;
    g_kprof[slot + 11] = __builtin_amdgcn_s_memrealtime() - tL0;
    g_kprof[slot + 10] = 1;
  }
#endif
}

int accflow_launch_corr_disp_direct(const accflow_conv_desc& d, hipStream_t st) {
  const int P = d.OH * d.OW, npb = cdiv(P, 128);
  if (!d.out2) return 1;
  const int nqt = cdiv(d.OH, 2) * cdiv(d.OW, 64);  // target tiles: 2 rows x 64 columns
  dim3 grid(8 * ((npb + 7) / 8) * nqt);  // (XCD, its query blocks, target tiles): see the kernel
  // ACCFLOW_CORR_GEMM=regs: the register-only operand loop (A/B); default: the LDS-ring form
  static const bool ringk = [] { const char* e = getenv("ACCFLOW_CORR_GEMM"); return !(e && e[0] == 'r'); }();
  if (d.mode == ACCFLOW_CONV_F16X3 && d.wpatch16 && ringk) {
    hipLaunchKernelGGL(corr_disp_ring_kernel, grid, dim3(256), 0, st, d);
  } else if (d.mode == ACCFLOW_CONV_F16X3 && d.wpatch16) hipLaunchKernelGGL((corr_disp_gemm_kernel<2, true>), grid, dim3(256), 0, st, d);
  else if (d.mode == ACCFLOW_CONV_BF16X3) hipLaunchKernelGGL((corr_disp_gemm_kernel<2, false>), grid, dim3(256), 0, st, d);
  else hipLaunchKernelGGL((corr_disp_gemm_kernel<3, false>), grid, dim3(256), 0, st, d);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

int accflow_launch_conv_direct(const accflow_conv_desc& d, int tc, hipStream_t st) {
  // Cout = 128 m + 64 (convc2: 256 -> 192): with 128-channel workgroups only, the last channel block would run all its
  // MFMAs with half of its rows on padding (25 % of the launch's matrix work wasted at 192).  Pointwise epilogues let
  // the launch be cut in two along the channels: 128 m channels on the 128-channel kernel, the last 64 on the
  // 64-channel kernel, through a descriptor whose channel-indexed pointers are advanced by ch0.
  const bool pointwise = d.epi == ACCFLOW_EPI_STORE || d.epi == ACCFLOW_EPI_RES_RELU || d.epi == ACCFLOW_EPI_ACCUM;
  if (tc == 2 && d.Cout > 128 && d.Cout % 128 > 0 && d.Cout % 128 <= 64 && pointwise && !d.stats && !accflow_tls_dry_slots && !(d.kws && (long long)d.B * cdiv(d.OW, DIR_TW) * cdiv(d.OH, DIR_TH) * cdiv(d.Cout, 128) < 320)) {
    const int ch0 = d.Cout / 128 * 128;
    const long long OHW = (long long)d.OH * d.OW;
    accflow_conv_desc a = d, b = d;
    a.Cout = ch0;
    b.Cout = d.Cout - ch0;
    b.out = d.out ? d.out + ch0 * OHW : nullptr;
    if (d.bias) b.bias = d.bias + ch0;
    if (d.wscale16) b.wscale16 = d.wscale16 + ch0;
    if (d.e0) b.e0 = d.e0 + ch0 * OHW;
    if (d.out16) b.out16 = reinterpret_cast<char*>(d.out16) + (long long)(ch0 / 8) * 2 * OHW * 16;
    // packs are [term][step][octet][CoutPad][8]: the same CoutPad pitch, the channel origin moved by ch0 16-byte rows
    if (d.wpatch) b.wpatch = reinterpret_cast<const char*>(d.wpatch) + (long long)ch0 * 16;
    if (d.wpatch16) b.wpatch16 = reinterpret_cast<const char*>(d.wpatch16) + (long long)ch0 * 16;
    const int rc = launch_conv_direct<2>(a, st);
    if (rc) return rc;
    return launch_conv_direct<1>(b, st);
  }
  return tc == 2 ? launch_conv_direct<2>(d, st) : launch_conv_direct<1>(d, st);
}

bool accflow_conv_direct_eligible(const accflow_conv_desc& d) {
  if (!d.wpatch || d.wsplit_bs || d.mode == ACCFLOW_CONV_F32 || d.offset || d.stride != 1 || d.Cout <= 4) return false;
  if (d.OH != d.H || d.OW != d.W) return false;                                // "same" convolutions only
  if ((DIR_TH + d.KH - 1) * (DIR_TW + d.KW - 1) > DIR_NPMAX) return false;
  if (d.C0 + d.C1 < 16) return false;                                          // 2 / 3-channel stems: im2col kernel
  if (d.in1 && (d.C0 % (d.KH * d.KW == 1 ? 32 : 16))) return false;            // a chunk must not straddle the sources
  return true;
}

#ifdef ACCFLOW_KPROF
#define ACCFLOW_DIRECT_UNITY
#include "conv2d_direct_v_s16.hip"
#include "conv2d_direct_v_s16tg.hip"
#include "conv2d_direct_v_s16k.hip"
#include "conv2d_direct_v_s16k9.hip"
#include "conv2d_direct_v_f16.hip"
#include "conv2d_direct_v_f16n.hip"
#include "conv2d_direct_v_bf16x6.hip"
#include "conv2d_direct_v_bf16x3.hip"
#include "conv2d_direct_v_bf16n.hip"
extern "C" int accflow_debug_occupancy(int* out) {
  int n = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[n++], conv2d_direct_bf16s_kernel<2, 3>, 256, 0);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[n++], conv2d_direct_bf16s_kernel<2, 2, true>, 256, 0);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[n++], conv2d_direct_bf16s_kernel<1, 3>, 256, 0);
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

