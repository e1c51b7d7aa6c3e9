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
#include "conv_common.h"

namespace {

// weights for the patch kernel: [3 terms][nchunk*T steps][2 octets][CoutPad][8] bf16, element (t, step = cc*T + tap,
// o, ch, q) = term t of w[ch][cc*16 + o*8 + q][tap] (* scale[ch]), zero beyond Cin / Cout
// f16: two fp16 terms of val * 2^k[ch], the row scale recovered from wscale16[ch] = 2^-(k[ch] + ACCFLOW_F16_ASHIFT)
// octs: octets per channel chunk (2 = the direct kernel's 16-channel chunks)
__global__ void conv_pack_patch_kernel(const float* __restrict__ w, const float* __restrict__ scale, int Cout, int Cin,
                                       int T, int CoutPad, unsigned short* __restrict__ wp, int f16,
                                       const float* __restrict__ wscale16, int octs = 2,
                                       const float* __restrict__ gptr = nullptr) {
  const int nstep = (Cin + 8 * octs - 1) / (8 * octs) * T;
  const long long per_term = (long long)nstep * octs * CoutPad * 8;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= per_term) return;
  const int q = (int)(idx & 7);
  long long r = idx >> 3;
  const int ch = (int)(r % CoutPad); r /= CoutPad;
  const int o = (int)(r % octs); r /= octs;
  const int step = (int)r, cc = step / T, tap = step % T;
  const int c = cc * 8 * octs + o * 8 + q;
  float val = 0.0f;
  if (c < Cin && ch < Cout) {
    val = w[((long long)ch * Cin + c) * T + tap];
    if (scale) val *= scale[ch];
    if (gptr) val *= gptr[0];
  }
  float rr = val;
  if (f16) {  // two fp16 terms (third slot zero) of the row-scaled weight: |rr| < 2^11, exact power-of-two scaling
    rr = val * (ldexpf(1.0f, -ACCFLOW_F16_ASHIFT) / wscale16[ch]);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const _Float16 hq = t < 2 ? (_Float16)rr : (_Float16)0.0f;
      wp[t * per_term + idx] = __builtin_bit_cast(unsigned short, hq);
      rr -= (float)hq;
    }
    return;
  }
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
                                       int kmajor, float cscale, const float* __restrict__ gptr,
                                       const float* __restrict__ wscale16 = nullptr) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)Kpad * CoutPad) return;
  w += (long long)blockIdx.y * Cout * Cin * KH * KW;          // batched use: one weight matrix per blockIdx.y
  ws += (long long)blockIdx.y * 3 * Kpad * CoutPad;
  if (wscale16) wscale16 += (long long)blockIdx.y * CoutPad;
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
  if (wscale16) {  // fp16 hi + lo of the row-scaled weight (third slot zero), see conv_pack_patch_kernel
    r = val * (ldexpf(1.0f, -ACCFLOW_F16_ASHIFT) / wscale16[o]);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const _Float16 hq = t < 2 ? (_Float16)r : (_Float16)0.0f;
      ws[t * per_term + dst] = __builtin_bit_cast(unsigned short, hq);
      r -= (float)hq;
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const __bf16 b = (__bf16)r;
    ws[t * per_term + dst] = __builtin_bit_cast(unsigned short, b);
    r -= (float)b;
  }
}

// The same pack for a K-MAJOR matrix (a feature map (C, P) used as weights: correlation level 0), one thread per
// (8 consecutive k, output column): coalesced reads along the columns and ONE 16-byte store per term (the generic
// kernel above writes 2-byte elements 16 bytes apart).  f16 != 0: fp16 hi + lo, the third slot is not written.
__global__ __launch_bounds__(256) void conv_pack_kmajor_kernel(const float* __restrict__ w, int Cout, int K, int Kpad,
                                                               int CoutPad, u32x4* __restrict__ ws, float cscale, int f16,
                                                               int* guard) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)(Kpad / 8) * CoutPad) return;
  const int k8 = (int)(idx / CoutPad), o = (int)(idx % CoutPad);
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k8 * 8 + j;
    x[j] = (k < K && o < Cout) ? w[(long long)k * Cout + o] * cscale : 0.0f;
  }
  const long long per_term = (long long)(Kpad / 8) * CoutPad;
  if (f16) {
    u32x4 t2[2];
    bool bad = false;
    split8_f16<0>(x, t2, bad, 1.0f);  // (cscale already carries 2^ACCFLOW_F16_ASHIFT)
    if (bad && guard) atomicOr(guard, 1);
    ws[idx] = t2[0];
    ws[per_term + idx] = t2[1];
  } else {
    u32x4 t3[3];
    split8_bf16<3, 0>(x, t3);
#pragma unroll
    for (int t = 0; t < 3; ++t) ws[t * per_term + idx] = t3[t];
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

// wscale16[ch] = 2^-(k + ACCFLOW_F16_ASHIFT) with k such that max_j |w[ch][j] * scale[ch]| * 2^k lies in [2^10, 2^11)
// (k = 0 for an all-zero / non-finite row, 0 for the padding channels): one workgroup per output channel
// (blockIdx.y: one weight matrix per batch item, gptr: a device scalar factor - the GMA aggregation's per-pair v * gamma)
__global__ __launch_bounds__(256) void conv_row_scale16_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                               int Cout, int rowlen, float* __restrict__ wscale16,
                                                               const float* __restrict__ gptr = nullptr) {
  __shared__ float red[256];
  const int ch = blockIdx.x;
  w += (long long)blockIdx.y * Cout * rowlen;
  wscale16 += (long long)blockIdx.y * gridDim.x;
  float m = 0.0f;
  if (ch < Cout)
    for (int j = threadIdx.x; j < rowlen; j += 256) m = fmaxf(m, fabsf(w[(long long)ch * rowlen + j]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x) return;
  m = red[0];
  if (ch < Cout && scale) m *= fabsf(scale[ch]);
  if (gptr) m *= fabsf(gptr[0]);
  int k = 0;
  if (m > 0.0f && m < 3.0e38f) {
    int e;
    frexpf(m, &e);  // m = f * 2^e, f in [0.5, 1)  ->  m * 2^(11 - e) in [2^10, 2^11)
    k = 11 - e;
    if (k > 100) k = 100;
    if (k < -100) k = -100;
  }
  wscale16[ch] = ldexpf(1.0f, -(k + ACCFLOW_F16_ASHIFT));
}

// conv_row_scale16_kernel + conv_pack_patch_kernel in ONE launch for 1x1 weight matrices that change every call (the GMA
// aggregation's v * gamma: 72 packs per sequence): one workgroup per output row - the row maximum, then that row's fp16
// hi / lo chunks in the patch layout [term][step][octet][CoutPad][8] (T = 1, 2 octets per step).
// Rows of up to ROWS16_REG * 2048 elements are read ONCE: every thread keeps its 8-element chunks (c8 = tid + 256 i) in
// registers between the maximum and the pack (the two-pass form read the 22 MB of a 3-item v twice: 31 us per launch, 72
// launches per C5 sequence); longer rows take the two-pass loop.
constexpr int ROWS16_REG = 8;
__global__ __launch_bounds__(256) void conv_pack_rows16_kernel(const float* __restrict__ w, int Cout, int Cin, int CoutPad,
                                                               unsigned short* __restrict__ wp, float* __restrict__ wscale16,
                                                               const float* __restrict__ gptr) {
  __shared__ float red[256];
  const int ch = blockIdx.x;
  const float gmul = gptr ? gptr[0] : 1.0f;
  const int nstep = (Cin + 15) / 16, nchunk = nstep * 2;
  const bool inreg = nchunk <= ROWS16_REG * 256 && !(Cin & 3);       // (rows start 16-byte aligned when Cin % 4 == 0)
  const float* row = w + (long long)ch * Cin;
  float keep[ROWS16_REG][8];
  float m = 0.0f;
  if (inreg) {
#pragma unroll
    for (int i = 0; i < ROWS16_REG; ++i) {
      const int c0 = (threadIdx.x + 256 * i) * 8;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = c0 + 4 * h;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (ch < Cout && c + 4 <= Cin) v = *reinterpret_cast<const f32x4*>(row + c);
        else if (ch < Cout) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = c + q < Cin ? row[c + q] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { keep[i][4 * h + q] = v[q]; m = fmaxf(m, fabsf(v[q])); }
      }
    }
  } else if (ch < Cout) {
    for (int j = threadIdx.x; j < Cin; j += 256) m = fmaxf(m, fabsf(row[j]));
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (threadIdx.x < s2) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s2]);
    __syncthreads();
  }
  m = red[0] * fabsf(gmul);
  int k = 0;
  if (m > 0.0f && m < 3.0e38f) {
    int e;
    frexpf(m, &e);
    k = 11 - e;
    if (k > 100) k = 100;
    if (k < -100) k = -100;
  }
  const float sc = ldexpf(1.0f, -(k + ACCFLOW_F16_ASHIFT));    // the same value conv_row_scale16_kernel stores
  if (threadIdx.x == 0) wscale16[ch] = sc;
  const float mul = ldexpf(1.0f, -ACCFLOW_F16_ASHIFT) / sc;
  const long long per_term = (long long)nstep * 2 * CoutPad * 8;
  auto emit = [&](int c8, const float (&x)[8]) {              // one 8-channel chunk (step, octet) of this row
    unsigned short hi[8], lo[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float rr = x[q] * gmul * mul;
      const _Float16 h = (_Float16)rr;
      rr -= (float)h;
      const _Float16 l = (_Float16)rr;
      hi[q] = __builtin_bit_cast(unsigned short, h);
      lo[q] = __builtin_bit_cast(unsigned short, l);
    }
    const long long base = ((long long)c8 * CoutPad + ch) * 8;   // (16-byte aligned: one vector store per term)
    u32x4 hv, lv;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      hv[q] = (unsigned)hi[2 * q] | ((unsigned)hi[2 * q + 1] << 16);
      lv[q] = (unsigned)lo[2 * q] | ((unsigned)lo[2 * q + 1] << 16);
    }
    *reinterpret_cast<u32x4*>(wp + base) = hv;
    *reinterpret_cast<u32x4*>(wp + per_term + base) = lv;
  };
  if (inreg) {
#pragma unroll
    for (int i = 0; i < ROWS16_REG; ++i) {
      const int c8 = threadIdx.x + 256 * i;
      if (c8 < nchunk) emit(c8, keep[i]);
    }
    return;
  }
  for (int c8 = threadIdx.x; c8 < nchunk; c8 += 256) {
    float x[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = c8 * 8 + q;
      x[q] = (c < Cin && ch < Cout) ? row[c] : 0.0f;
    }
    emit(c8, x);
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

// Every pack of one convolution in ONE launch (accflow_conv_pack_all_f32): what accflow_conv_pack_f32, _pack_bf16s, _pack_patch,
// _pack_patch16 and _pack_split16 write with up to seven launches - a training step rebuilds the packs of ~40 trainable
// convolutions and of their transposed forms after every optimizer step, 570 launches of ~5 us inside the step's graph.  One
// workgroup per output row: the row's largest magnitude (the fp16 packs' power-of-two scale, conv_row_scale16_kernel's value),
// then the row's elements in the im2col order (wpack, ktab, wsplit, wsplit16) and in the patch order (wpatch, wpatch16); every
// value is computed by the expressions of the single-purpose kernels above, so the packs are bit-identical to theirs.
// tflip: `w` is the (Cin, Cout, KH, KW) weight of the FORWARD convolution and the logical weight is its transposed, flipped
// form  W[o][c][ky][kx] = w[c][o][KH-1-ky][KW-1-kx]  - the input-gradient convolution's weights without materialising them.
__global__ __launch_bounds__(256) void conv_pack_all_kernel(const float* __restrict__ w, const float* __restrict__ scale, int Cout,
                                                            int Cin, int KH, int KW, int C0, int tap_major, int tflip, int Kpad,
                                                            int CoutPad, float* __restrict__ wpack, int4* __restrict__ ktab,
                                                            unsigned short* __restrict__ wsplit, unsigned short* __restrict__ wpatch,
                                                            unsigned short* __restrict__ wpatch16, float* __restrict__ wscale16,
                                                            unsigned short* __restrict__ wsplit16) {
  __shared__ float red[256];
  const int ch = blockIdx.x, T = KH * KW, K = Cin * T;
  auto wat = [&](int c, int t) -> float {     // logical w[ch][c][tap t]
    if (!tflip) return w[((long long)ch * Cin + c) * T + t];
    const int ky = t / KW, kx = t - ky * KW;
    return w[(((long long)c * Cout + ch) * KH + (KH - 1 - ky)) * KW + (KW - 1 - kx)];
  };
  float rscale = 1.0f;    // 2^-ASHIFT / wscale16[ch]
  if (wscale16) {
    float m = 0.0f;
    if (ch < Cout)
      for (int j = threadIdx.x; j < K; j += 256) m = fmaxf(m, fabsf(wat(j / T, j % T)));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
      if (threadIdx.x < s2) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s2]);
      __syncthreads();
    }
    m = red[0];
    if (ch < Cout && scale) m *= fabsf(scale[ch]);
    int k = 0;
    if (m > 0.0f && m < 3.0e38f) {
      int e;
      frexpf(m, &e);
      k = 11 - e;
      if (k > 100) k = 100;
      if (k < -100) k = -100;
    }
    const float ws = ldexpf(1.0f, -(k + ACCFLOW_F16_ASHIFT));
    if (threadIdx.x == 0) wscale16[ch] = ws;
    rscale = ldexpf(1.0f, -ACCFLOW_F16_ASHIFT) / ws;
  }
  const float sc = (scale && ch < Cout) ? scale[ch] : 1.0f;
  const bool has_sc = scale != nullptr;
  // ---- im2col order: k = (c, tap) (tap_major: (tap, c)), Kpad entries ----
  const long long per_term = (long long)Kpad * CoutPad;
  for (int k = threadIdx.x; k < Kpad; k += 256) {
    int c = 0, t = 0;
    if (k < K) {
      if (tap_major) { t = k / Cin; c = k % Cin; } else { c = k / T; t = k % T; }
    }
    float val = 0.0f;
    if (k < K && ch < Cout) {
      val = wat(c, t);
      if (has_sc) val *= sc;
    }
    wpack[(long long)k * CoutPad + ch] = val;
    if (ch == 0) {
      const int ky = t / KW, kx = t % KW;
      int4 e;
      if (k < K) { e.x = c < C0 ? c : c - C0; e.y = ky; e.z = kx; e.w = c < C0 ? 0 : 1; }
      else { e.x = 0; e.y = 1 << 20; e.z = 1 << 20; e.w = 0; }
      ktab[k] = e;
    }
    if (wsplit || wsplit16) {     // (never tap_major: k = c * T + t) - conv_pack_bf16s_kernel's value: * scale, * cscale (= 1)
      float v2 = 0.0f;
      if (k < K && ch < Cout) {
        v2 = wat(k / T, k % T);
        if (has_sc) v2 *= sc;
        v2 *= 1.0f;
      }
      const long long dst = ((long long)(k / 8) * CoutPad + ch) * 8 + (k % 8);
      if (wsplit) {
        float r = v2;
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
          const __bf16 b = (__bf16)r;
          wsplit[tt * per_term + dst] = __builtin_bit_cast(unsigned short, b);
          r -= (float)b;
        }
      }
      if (wsplit16) {
        float r = v2 * rscale;
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
          const _Float16 hq = tt < 2 ? (_Float16)r : (_Float16)0.0f;
          wsplit16[tt * per_term + dst] = __builtin_bit_cast(unsigned short, hq);
          r -= (float)hq;
        }
      }
    }
  }
  // ---- patch order: (16-channel chunk, tap) steps, 2 octets ----
  if (wpatch || wpatch16) {
    const int nstep = (Cin + 15) / 16 * T;
    const long long pterm = (long long)nstep * 2 * CoutPad * 8;
    for (int i = threadIdx.x; i < nstep * 16; i += 256) {
      const int q = i & 7, o = (i >> 3) & 1, step = i >> 4;
      const int cc = step / T, tap = step % T;
      const int c = cc * 16 + o * 8 + q;
      float val = 0.0f;
      if (c < Cin && ch < Cout) {
        val = wat(c, tap);
        if (has_sc) val *= sc;
      }
      const long long idx = (((long long)step * 2 + o) * CoutPad + ch) * 8 + q;
      if (wpatch) {
        float rr = val;
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
          const __bf16 bq = (__bf16)rr;
          wpatch[tt * pterm + idx] = __builtin_bit_cast(unsigned short, bq);
          rr -= (float)bq;
        }
      }
      if (wpatch16) {
        float rr = val * rscale;
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
          const _Float16 hq = tt < 2 ? (_Float16)rr : (_Float16)0.0f;
          wpatch16[tt * pterm + idx] = __builtin_bit_cast(unsigned short, hq);
          rr -= (float)hq;
        }
      }
    }
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
                     KH * KW, accflow_conv_coutpad(Cout), reinterpret_cast<unsigned short*>(wpatch), 0, nullptr);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_conv_pack_patch16(const float* w, const float* scale, int Cout, int Cin, int KH, int KW,
                                         void* wpatch16, float* wscale16, void* stream) {
  if (!w || !wpatch16 || !wscale16 || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return 1;
  const long long n = accflow_conv_patch_elems(Cout, Cin, KH, KW) / 3;
  const int CoutPad = accflow_conv_coutpad(Cout);
  hipLaunchKernelGGL(conv_row_scale16_kernel, dim3(CoutPad), dim3(256), 0, as_stream(stream), w, scale, Cout,
                     Cin * KH * KW, wscale16);
  hipLaunchKernelGGL(conv_pack_patch_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), w, scale, Cout, Cin,
                     KH * KW, CoutPad, reinterpret_cast<unsigned short*>(wpatch16), 1, wscale16);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_conv_pack_split16(const float* w, const float* scale, int Cout, int Cin, int KH, int KW,
                                         void* wsplit16, float* wscale16, void* stream) {
  if (!w || !wsplit16 || !wscale16 || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return 1;
  const int Kpad = accflow_conv_kpad(Cin, KH, KW), CoutPad = accflow_conv_coutpad(Cout);
  const long long n = (long long)Kpad * CoutPad;
  hipLaunchKernelGGL(conv_row_scale16_kernel, dim3(CoutPad), dim3(256), 0, as_stream(stream), w, scale, Cout,
                     Cin * KH * KW, wscale16);
  hipLaunchKernelGGL(conv_pack_bf16s_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), w, scale, Cout, Cin, KH,
                     KW, Kpad, CoutPad, reinterpret_cast<unsigned short*>(wsplit16), 0, 1.0f, nullptr, wscale16);
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
// guard: device flag of the f16x3 mode (caller-owned, may be NULL).
// lvl1 (displaced layout only): the register-only GEMM also emits level 1; *lvl1_done tells the caller so.
int accflow_corr_level0_bf16s(const float* fmap1, const float* fmap2, float* lvl0, void* ws, int B, int C, int H8,
                              int W8, int mode, int disp, int* guard, hipStream_t st, float* lvl1, int* lvl1_done) {
  if (lvl1_done) *lvl1_done = 0;
  const int P = H8 * W8;
  const int Kpad = accflow_conv_kpad(C, 1, 1), CoutPad = accflow_conv_coutpad(P);
  unsigned short* wsplit = reinterpret_cast<unsigned short*>(ws);
  int* ktab = reinterpret_cast<int*>(wsplit + 3LL * Kpad * CoutPad);
  const float cscale = 1.0f / sqrtf((float)C);
  // the displaced layout comes out of the direct kernel (a 1x1 convolution's [term][step][octet][ch][8] weight pack
  // is the [term][k/8][ch][8] split itself); in f16x3 mode fmap1 is packed as fp16 hi + lo
  const bool direct = disp && C >= 16 && C % 16 == 0 && W8 % 2 == 0 && lvl1;
  const bool f16 = direct && mode == ACCFLOW_CONV_F16X3;
  if (mode == ACCFLOW_CONV_F16X3 && !f16) mode = ACCFLOW_CONV_BF16X6;
  // f16x3: both feature maps are split as fp16 hi + lo of x * 2^ACCFLOW_F16_ASHIFT (exact scaling; keeps lo a normal
  // number down to |x| = 2^-7) and the accumulator is multiplied by 2^-2*ASHIFT in the store
  const float fs = f16 ? ldexpf(1.0f, ACCFLOW_F16_ASHIFT) : 1.0f;
  for (int b = 0; b < B; ++b) {
    const long long n = (long long)Kpad * CoutPad;
    hipLaunchKernelGGL(conv_pack_kmajor_kernel, dim3(cdiv(n / 8, 256)), dim3(256), 0, st, fmap1 + (long long)b * C * P, P, C,
                       Kpad, CoutPad, reinterpret_cast<u32x4*>(wsplit), cscale * fs, f16 ? 1 : 0, guard);
    if (b == 0) hipLaunchKernelGGL(conv_ktab_kernel, dim3(cdiv(Kpad, 256)), dim3(256), 0, st, C, Kpad, reinterpret_cast<int4*>(ktab));
    accflow_conv_desc d = {};
    d.in0 = fmap2 + (long long)b * C * P; d.in0_bs = (long long)C * P; d.C0 = C; d.C1 = 0;
    d.B = 1; d.H = H8; d.W = W8; d.OH = H8; d.OW = W8; d.KH = 1; d.KW = 1; d.stride = 1; d.padH = 0; d.padW = 0;
    d.Cout = P; d.wpack = reinterpret_cast<const float*>(wsplit) /* unused in split modes */; d.ktab = ktab;
    const long long pair_elems = disp ? (long long)((P + 127) / 128) * 128 * P : (long long)P * P;  // displaced: p padded to 128
    d.Kpad = Kpad; d.CoutPad = CoutPad; d.out = lvl0 + (long long)b * pair_elems; d.out_bs = pair_elems;
    d.act = ACCFLOW_ACT_NONE; d.epi = ACCFLOW_EPI_STORE; d.wsplit = wsplit; d.mode = mode;
    if (direct) {  // register-only GEMM: fmap2 packed the same way (no scale) right behind the k-table
      unsigned short* bsplit = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(ws) + 3LL * Kpad * CoutPad * 2 +
                                                                 (long long)Kpad * 16);
      hipLaunchKernelGGL(conv_pack_kmajor_kernel, dim3(cdiv(n / 8, 256)), dim3(256), 0, st, fmap2 + (long long)b * C * P, P,
                         C, Kpad, CoutPad, reinterpret_cast<u32x4*>(bsplit), fs, f16 ? 1 : 0, guard);
      d.in0 = reinterpret_cast<const float*>(bsplit);
      d.wpatch = wsplit;
      if (f16) { d.wpatch16 = wsplit; d.guard = guard; d.acc_scale = 1.0f / (fs * fs); }
      if (!lvl1) return 1;
      d.out2 = lvl1 + (long long)b * ((P + 127) / 128) * 128 * (H8 >> 1) * (W8 >> 1);
      const int rc = accflow_launch_corr_disp_direct(d, st);
      if (rc) return rc;
      if (lvl1_done) *lvl1_done = 1;
      continue;
    }
    if (disp) {
      const int rc = accflow_launch_corr_disp_bf16s(d, st);
      if (rc) return rc;
      continue;
    }
    const int rc = accflow_conv2d_f32(&d, st);
    if (rc) return rc;
  }
  return (int)hipGetLastError();
}

// Per-FRAME operand packs of the displaced correlation GEMM.  A 7-frame sequence evaluates 11 pairs over 7 feature maps
// (frame 0 is the target of 6 of them): packing per pair split every map up to 6 times and needed the pair-major copies
// torch.cat made of them.  One pack per frame, x * 2^ACCFLOW_F16_ASHIFT as fp16 hi + lo (or 3 bf16 terms), serves as
// A (queries) and as B (targets); the 1/sqrt(C) of corr.py:55 moves into the accumulator scale.
extern "C" long long accflow_corr_pack_bytes(int C, int H8, int W8) {
  return 3LL * accflow_conv_kpad(C, 1, 1) * accflow_conv_coutpad(H8 * W8) * 2;
}

extern "C" int accflow_corr_pack_f32(const float* fmaps, void* packs, int mode, int* guard, int F, int C, int H8, int W8,
                                     void* stream) {
  if (!fmaps || !packs || F <= 0 || C < 16 || (C % 16) || (W8 & 1) || !accflow_corr_disp_supported(H8, W8)) return 1;
  if (mode != ACCFLOW_CONV_BF16X3 && mode != ACCFLOW_CONV_BF16X6 && mode != ACCFLOW_CONV_F16X3) return 1;
  const int P = H8 * W8, Kpad = accflow_conv_kpad(C, 1, 1), CoutPad = accflow_conv_coutpad(P);
  const bool f16 = mode == ACCFLOW_CONV_F16X3;
  const float fs = f16 ? ldexpf(1.0f, ACCFLOW_F16_ASHIFT) : 1.0f;
  const long long n = (long long)Kpad * CoutPad, bytes = accflow_corr_pack_bytes(C, H8, W8);
  for (int f = 0; f < F; ++f)
    hipLaunchKernelGGL(conv_pack_kmajor_kernel, dim3(cdiv(n / 8, 256)), dim3(256), 0, as_stream(stream),
                       fmaps + (long long)f * C * P, P, C, Kpad, CoutPad,
                       reinterpret_cast<u32x4*>(reinterpret_cast<char*>(packs) + f * bytes), fs, f16 ? 1 : 0, guard);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

int accflow_corr_disp_pool_from(const float* lvl0, float* lvl1, float* lvl2, float* lvl3, int B, int H8, int W8, int first,
                                hipStream_t st);

extern "C" int accflow_corr_volume_disp_packed_f32(const void* packs, int F, const int* idx1, const int* idx2, float* lvl0,
                                                   float* lvl1, float* lvl2, float* lvl3, int mode, int* guard, int B,
                                                   int C, int H8, int W8, void* stream) {
  if (!packs || F <= 0 || !idx1 || !idx2 || !lvl0 || !lvl1 || !lvl2 || !lvl3 || B <= 0 || C < 16 || (C % 16) || (W8 & 1) ||
      !accflow_corr_disp_supported(H8, W8))
    return 1;
  for (int b = 0; b < B; ++b)   // (host arrays) every pair inside the pack buffer, checked before the first launch
    if (idx1[b] < 0 || idx2[b] < 0 || idx1[b] >= F || idx2[b] >= F) return 1;
  if (mode != ACCFLOW_CONV_BF16X3 && mode != ACCFLOW_CONV_BF16X6 && mode != ACCFLOW_CONV_F16X3) return 1;
  hipStream_t st = as_stream(stream);
  const int P = H8 * W8, Kpad = accflow_conv_kpad(C, 1, 1), CoutPad = accflow_conv_coutpad(P);
  const bool f16 = mode == ACCFLOW_CONV_F16X3;
  const float fs = f16 ? ldexpf(1.0f, ACCFLOW_F16_ASHIFT) : 1.0f;
  const long long bytes = accflow_corr_pack_bytes(C, H8, W8);
  const long long pair0 = (long long)((P + 127) / 128) * 128 * P, pair1 = (long long)((P + 127) / 128) * 128 * (H8 >> 1) * (W8 >> 1);
  for (int b = 0; b < B; ++b) {
    const char* a = reinterpret_cast<const char*>(packs) + idx1[b] * bytes;
    const char* t = reinterpret_cast<const char*>(packs) + idx2[b] * bytes;
    accflow_conv_desc d = {};
    d.in0 = reinterpret_cast<const float*>(t);
    d.C0 = C; d.B = 1; d.H = H8; d.W = W8; d.OH = H8; d.OW = W8; d.KH = 1; d.KW = 1; d.stride = 1;
    d.Cout = P; d.Kpad = Kpad; d.CoutPad = CoutPad;
    d.out = lvl0 + b * pair0; d.out_bs = pair0; d.out2 = lvl1 + b * pair1;
    d.mode = mode; d.wpatch = a;
    if (f16) { d.wpatch16 = a; d.guard = guard; }
    d.acc_scale = (1.0f / sqrtf((float)C)) / (fs * fs);   // corr / sqrt(dim) (corr.py:55) and the operands' 2^ASHIFT
    const int rc = accflow_launch_corr_disp_direct(d, st);
    if (rc) return rc;
  }
  return accflow_corr_disp_pool_from(lvl0, lvl1, lvl2, lvl3, B, H8, W8, 1, st);
}

// GMA aggregation (gma/modules.py:102-115) as B independent 1x1 convolutions on the split-bf16 matrix cores:
// out[b][d][i] = fmap[b][d][i] + gamma * sum_j v[b][d][j] * attnT[b][j][i].  The TRANSPOSED attention (j-major) is
// exactly a (1, P channels, h, w) activation tensor, v[b] the (D x P) weight matrix (re-split every call, gamma folded
// in), and the residual add is the conv's accumulate epilogue.  ws as accflow_corr_volume_ws_bytes-style scratch:
// 3*Kpad*CoutPad uint16 + Kpad int4.
int accflow_gma_aggregate_conv(const float* attnT, const float* v, const float* fmap, const float* gamma, float* out,
                               long long out_bs, void* ws, int mode, int* guard, int B, int D, int H, int W, hipStream_t st) {
  const int P = H * W;
  const int Kpad = accflow_conv_kpad(P, 1, 1), CoutPad = accflow_conv_coutpad(D);
  if (mode == ACCFLOW_CONV_F16X3 && B == 1 && P >= 16) {
    // one item (or a stacked run of items that share the attention): the LDS-patch kernel (about twice the im2col
    // kernel's rate) with v * gamma packed as its fp16 weights; split-K fills the chip (115 pixel tiles at 720x1280)
    unsigned short* wpatch16 = reinterpret_cast<unsigned short*>(ws);
    float* wscale16 = reinterpret_cast<float*>(wpatch16 + accflow_conv_patch_elems(D, P, 1, 1));
    float* kws = wscale16 + CoutPad;
    const long long n = accflow_conv_patch_elems(D, P, 1, 1) / 3;
    hipLaunchKernelGGL(conv_row_scale16_kernel, dim3(CoutPad), dim3(256), 0, st, v, nullptr, D, P, wscale16, gamma);
    hipLaunchKernelGGL(conv_pack_patch_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, v, nullptr, D, P, 1, CoutPad, wpatch16, 1,
                       wscale16, 2, gamma);
    accflow_conv_desc d = {};
    d.in0 = attnT; d.in0_bs = (long long)P * P; d.C0 = P;
    d.B = 1; d.H = H; d.W = W; d.OH = H; d.OW = W; d.KH = 1; d.KW = 1; d.stride = 1;
    d.Cout = D; d.Kpad = Kpad; d.CoutPad = CoutPad;
    d.out = out; d.out_bs = out_bs;
    d.act = ACCFLOW_ACT_NONE; d.epi = ACCFLOW_EPI_ACCUM; d.e0 = fmap; d.e0_bs = (long long)D * P;
    d.wpatch = wpatch16; d.wpatch16 = wpatch16; d.wscale16 = wscale16; d.mode = mode; d.guard = guard;
    d.kws = kws; d.kws_elems = 8LL * D * P;
    if (!accflow_conv_direct_eligible(d)) return 1;
    return accflow_launch_conv_direct(d, D > 64 ? 2 : 1, st);
  }
  unsigned short* wsplit = reinterpret_cast<unsigned short*>(ws);
  int* ktab = reinterpret_cast<int*>(wsplit + 3LL * Kpad * CoutPad * B);
  float* wscale16 = reinterpret_cast<float*>(reinterpret_cast<char*>(ktab) + (long long)Kpad * 16);  // [B][CoutPad]
  float* kws = wscale16 + (((long long)B * CoutPad + 3) & ~3LL);   // B == 1: 8 * D * P floats of split-K partials
  hipLaunchKernelGGL(conv_ktab_kernel, dim3(cdiv(Kpad, 256)), dim3(256), 0, st, P, Kpad, reinterpret_cast<int4*>(ktab));
  const long long n = (long long)Kpad * CoutPad;
  // f16x3: v[b] * gamma as fp16 hi + lo with per-row (and per-pair) power-of-two scales; the attention (values in
  // [0, 1]) is split by the kernel with the usual 2^ACCFLOW_F16_ASHIFT.  Half the MFMAs of the bf16x6 form.
  const bool f16 = mode == ACCFLOW_CONV_F16X3 && (P % 64) == 0;
  if (mode == ACCFLOW_CONV_F16X3 && !f16) mode = ACCFLOW_CONV_BF16X6;
  if (f16)
    hipLaunchKernelGGL(conv_row_scale16_kernel, dim3(CoutPad, B), dim3(256), 0, st, v, nullptr, D, P, wscale16, gamma);
  hipLaunchKernelGGL(conv_pack_bf16s_kernel, dim3(cdiv(n, 256), B), dim3(256), 0, st, v, nullptr, D, P, 1, 1, Kpad, CoutPad,
                     wsplit, 0, 1.0f, gamma, f16 ? wscale16 : nullptr);
  accflow_conv_desc d = {};
  d.in0 = attnT; d.in0_bs = (long long)P * P; d.C0 = P; d.C1 = 0;
  d.B = B; d.H = H; d.W = W; d.OH = H; d.OW = W; d.KH = 1; d.KW = 1; d.stride = 1;
  d.Cout = D; d.wpack = reinterpret_cast<const float*>(wsplit); d.ktab = ktab; d.Kpad = Kpad; d.CoutPad = CoutPad;
  d.out = out; d.out_bs = out_bs;
  d.act = ACCFLOW_ACT_NONE; d.epi = ACCFLOW_EPI_ACCUM; d.e0 = fmap; d.e0_bs = (long long)D * P;
  d.wsplit = wsplit; d.wsplit_bs = 3LL * Kpad * CoutPad * 2; d.mode = mode;
  if (f16) { d.wsplit16 = wsplit; d.wscale16 = wscale16; d.guard = guard; }
  if (B == 1) { d.kws = kws; d.kws_elems = 8LL * D * P; }
  if (P % 64 == 0) {  // 64-pixel tiles never straddle two pairs: as few launches as 32-bit buffer offsets allow
    const long long per_pair = (long long)P * P * 4;
    const int chunk = (int)(((1LL << 32) - 1) / per_pair);
    if (chunk < 1) return 1;
    for (int b0 = 0; b0 < B; b0 += chunk) {
      accflow_conv_desc e = d;
      e.B = B - b0 < chunk ? B - b0 : chunk;
      e.in0 = attnT + (long long)b0 * P * P; e.out = out + (long long)b0 * out_bs; e.e0 = fmap + (long long)b0 * D * P;
      e.wsplit = wsplit + 3LL * Kpad * CoutPad * b0;
      if (f16) { e.wsplit16 = e.wsplit; e.wscale16 = wscale16 + (long long)b0 * CoutPad; }
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

// GMA aggregation on the pre-split attention (accflow_gma_attention_s16): ONE GEMM for the n batch items that share an
// attention matrix - rows = the n x D channels of v * gamma (packed per call as fp16 weights with their row scales), the
// S16 attention as the activation operand through the direct kernel's DMA loader (no split in the K loop: the matrix is
// constant over the refinement iterations), split-K over the P = h*w deep reduction - whose epilogue reads each item's
// residual and writes each item's slice (fp32 and / or S16) through the channel-block scatter: no copies around it.
int accflow_gma_aggregate_s16_impl(const void* attn16, const float* v, const float* fmap, long long fmap_bs, const float* gamma,
                                   float* out, long long out_bs, void* out16, long long out16_bs, void* ws, int* guard, int n,
                                   int D, int H, int W, hipStream_t st) {
  const int P = H * W, Cout = n * D;
  const int Kpad = accflow_conv_kpad(P, 1, 1), CoutPad = accflow_conv_coutpad(Cout);
  unsigned short* wpatch16 = reinterpret_cast<unsigned short*>(ws);
  float* wscale16 = reinterpret_cast<float*>(wpatch16 + accflow_conv_patch_elems(Cout, P, 1, 1));
  float* kws = wscale16 + CoutPad;
  hipLaunchKernelGGL(conv_pack_rows16_kernel, dim3(CoutPad), dim3(256), 0, st, v, Cout, P, CoutPad, wpatch16, wscale16, gamma);
  accflow_conv_desc d = {};
  d.in0 = reinterpret_cast<const float*>(attn16); d.in0_bs = accflow_s16_item_words(P, H, W); d.C0 = P; d.in_fmt = 1;
  d.B = 1; d.H = H; d.W = W; d.OH = H; d.OW = W; d.KH = 1; d.KW = 1; d.stride = 1;
  d.Cout = Cout; d.Kpad = Kpad; d.CoutPad = CoutPad;
  d.out = out; d.out_bs = (long long)n * out_bs;
  d.out16 = out16; d.out16_bs = (long long)n * out16_bs;
  d.act = ACCFLOW_ACT_NONE; d.epi = ACCFLOW_EPI_ACCUM; d.e0 = fmap; d.e0_bs = (long long)n * fmap_bs;
  d.cb = D; d.out_cbs = out_bs; d.e0_cbs = fmap_bs; d.out16_cbs = out16_bs;
  d.wpatch = wpatch16; d.wpatch16 = wpatch16; d.wscale16 = wscale16; d.mode = ACCFLOW_CONV_F16X3; d.guard = guard;
  d.kws = kws; d.kws_elems = 8LL * Cout * P;
  if (!accflow_conv_direct_eligible(d)) return 1;
  return accflow_launch_conv_direct(d, Cout > 64 ? 2 : 1, st);
}

// GMA attention -> S16 (gma_attn_gemm_kernel, conv2d_direct.hip): q and k are packed as fp16 hi / lo of x * 2^ASHIFT (k
// carries GMA's scale dim_head^-1/2), two passes of the register-only GEMM per item.
// ws: accflow_gma_attention_s16_ws_bytes(D, H, W) = two packs + partial / final column statistics.
int accflow_launch_gma_attn(const void* kpack, const void* qpack, float* part, float* stats, void* out16, float acc_scale, int P,
                            int D, int Ppad, hipStream_t st);

extern "C" long long accflow_gma_attention_s16_ws_bytes(int D, int H, int W) {
  const long long P = (long long)H * W, Ppad = accflow_conv_coutpad((int)P), nb = (P + 127) / 128;
  return 2 * (2LL * (D / 8) * Ppad * 16) + (nb * P * 2 + P * 2) * 4 + 256;
}

extern "C" int accflow_gma_attention_s16(const float* qk, void* attn16, void* ws, int* guard, int B, int D, int H, int W,
                                         float scale, void* stream) {
  if (!qk || !attn16 || !ws || B <= 0 || D < 16 || (D % 16) || H <= 0 || W <= 0) return 1;
  const int P = H * W, Ppad = accflow_conv_coutpad(P), Kpad = accflow_conv_kpad(D, 1, 1);
  if (Kpad != D) return 1;
  hipStream_t st = as_stream(stream);
  const long long pack_bytes = 2LL * (D / 8) * Ppad * 16;
  char* kp = reinterpret_cast<char*>(ws);
  char* qp = kp + pack_bytes;
  float* part = reinterpret_cast<float*>(qp + pack_bytes);
  float* stats = part + (long long)cdiv(P, 128) * P * 2;
  const float fs = ldexpf(1.0f, ACCFLOW_F16_ASHIFT);
  const long long n8 = (long long)(Kpad / 8) * Ppad, item = accflow_s16_item_words(P, H, W);
  for (int b = 0; b < B; ++b) {
    const float* q = qk + (long long)b * 2 * D * P;   // q = qk[:, :D], k = qk[:, D:] (modules.py:63)
    hipLaunchKernelGGL(conv_pack_kmajor_kernel, dim3(cdiv(n8, 256)), dim3(256), 0, st, q + (long long)D * P, P, D, Kpad, Ppad,
                       reinterpret_cast<u32x4*>(kp), scale * fs, 1, guard);
    hipLaunchKernelGGL(conv_pack_kmajor_kernel, dim3(cdiv(n8, 256)), dim3(256), 0, st, q, P, D, Kpad, Ppad,
                       reinterpret_cast<u32x4*>(qp), fs, 1, guard);
    const int rc = accflow_launch_gma_attn(kp, qp, part, stats, reinterpret_cast<unsigned*>(attn16) + b * item, 1.0f / (fs * fs),
                                           P, D, Ppad, st);
    if (rc) return rc;
  }
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" long long accflow_s16_item_words(int C, int H, int W) {
  return (long long)((C + 7) / 8) * 2 * H * W * 4;   // octets x 2 terms x pixels x 16 bytes, in 4-byte words
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

extern "C" int accflow_conv_pack_all_f32(const float* w, const float* scale, int Cout, int Cin, int KH, int KW, int C0, int tap_major,
                                         int transpose_flip, float* wpack, int* ktab, void* wsplit, void* wpatch, void* wpatch16,
                                         float* wscale16, void* wsplit16, void* stream) {
  if (!w || !wpack || !ktab || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return 1;
  if ((wpatch16 || wsplit16) && !wscale16) return 1;
  if (tap_major && (wsplit || wpatch || wpatch16 || wsplit16)) return 1;      // (the matrix-core packs are (c, tap)-ordered)
  const int Kpad = accflow_conv_kpad(Cin, KH, KW), CoutPad = accflow_conv_coutpad(Cout);
  hipLaunchKernelGGL(conv_pack_all_kernel, dim3(CoutPad), dim3(256), 0, as_stream(stream), w, scale, Cout, Cin, KH, KW, C0, tap_major,
                     transpose_flip, Kpad, CoutPad, wpack, reinterpret_cast<int4*>(ktab), reinterpret_cast<unsigned short*>(wsplit),
                     reinterpret_cast<unsigned short*>(wpatch), reinterpret_cast<unsigned short*>(wpatch16), wscale16,
                     reinterpret_cast<unsigned short*>(wsplit16));
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

thread_local int* accflow_tls_dry_slots = nullptr;
thread_local int* accflow_tls_dry_route = nullptr;

extern "C" int accflow_conv_in_norm_supported(const accflow_conv_desc* desc) {
  if (!desc) return 0;
  int slots = 0, route = 0;
  accflow_conv_desc probe = *desc;
  probe.in_norm = nullptr;
  probe.stats = nullptr;
  accflow_tls_dry_slots = &slots;
  accflow_tls_dry_route = &route;
  const int rc = accflow_conv2d_f32(&probe, nullptr);
  accflow_tls_dry_slots = nullptr;
  accflow_tls_dry_route = nullptr;
  return rc ? 0 : route;
}

extern "C" int accflow_conv_stat_slots(const accflow_conv_desc* desc) {
  if (!desc || desc->epi != ACCFLOW_EPI_STORE || desc->act != ACCFLOW_ACT_NONE) return 0;
  int slots = 0;
  accflow_tls_dry_slots = &slots;
  const int rc = accflow_conv2d_f32(desc, nullptr);
  accflow_tls_dry_slots = nullptr;
  return rc ? 0 : slots;
}

bool accflow_conv_stem_eligible(const accflow_conv_desc& d);              // conv_stem.hip
int accflow_launch_conv_stem(const accflow_conv_desc& d, hipStream_t st);

// (the pre-split GRU state rides on the tap-specialised 5-tap instantiations: both A/B switches of that route must be on)
static bool accflow_conv_gru16_supported() {
  static const bool ok = [] {
    const char* a = getenv("ACCFLOW_DIRECT_KT"); const char* b = getenv("ACCFLOW_DIRECT_W4"); const char* c = getenv("ACCFLOW_S16M");
    return !(a && atoi(a) == 0) && !(b && atoi(b) == 0) && !(c && atoi(c) != 0);
  }();
  return ok;
}

extern "C" int accflow_conv2d_f32(const accflow_conv_desc* desc, void* stream) {
  if (!desc) return 1;
  accflow_conv_desc dd = *desc;
  // the row / activation scales belong to the fp16 pack: every other kernel must not see them
  if (dd.mode == ACCFLOW_CONV_F16X3 && (dd.wpatch16 || dd.wsplit16) && !dd.wscale16) return 1;
  if (dd.mode != ACCFLOW_CONV_F16X3) { dd.wpatch16 = nullptr; dd.wsplit16 = nullptr; }
  if (!dd.wpatch16 && !dd.wsplit16) dd.wscale16 = nullptr;
  dd.acc_scale = 0.0f;
  if (dd.nsrc < 0 || dd.nsrc > ACCFLOW_CONV_MAX_SRC) return 1;
  // an S16 e0: the multi-source kernel's residual operand, or - round 6 - the GRU state of the 5-tap S16 direct convolutions on
  // PACKED operands (accflow_conv_desc.p32): GRU_ZR with p32 = 3, GRU_Q with p32 = 6, whole 128-channel blocks, no split-K
  const bool gru16 = !dd.nsrc && dd.e0_fmt;
  if (gru16 && !(dd.in_fmt == (dd.in1 ? 3 : 1) && dd.e0 && dd.pre && dd.out16 && dd.mode == ACCFLOW_CONV_F16X3 && !dd.cb && !dd.kws &&
                 dd.KH * dd.KW == 5 && dd.stride == 1 && accflow_conv_gru16_supported() && !dd.stats &&
                 ((dd.epi == ACCFLOW_EPI_GRU_ZR && dd.p32 == 3 && dd.out && dd.Cout == 256) ||
                  (dd.epi == ACCFLOW_EPI_GRU_Q && dd.p32 == 6 && dd.e1 && dd.Cout == 128))))
    return 1;
  // a pixel-major fp32 destination alone: a plain store of an S16-source direct convolution on whole 128-channel blocks
  if (!gru16 && dd.p32 && !(dd.p32 == 1 && !dd.nsrc && dd.in_fmt == (dd.in1 ? 3 : 1) && dd.mode == ACCFLOW_CONV_F16X3 && dd.out && !dd.out16 &&
                            dd.epi == ACCFLOW_EPI_STORE && (dd.act == ACCFLOW_ACT_NONE || dd.act == ACCFLOW_ACT_RELU) && !dd.cb &&
                            !dd.kws && !dd.stats && dd.stride == 1 && (dd.Cout % 128) == 0 && accflow_conv_gru16_supported()))
    return 1;
  const bool multi = dd.nsrc > 0;   // multi-source S16 form: src[] replaces in0 / in1 and the conv geometry fields
  if (multi) {
    if (dd.mode != ACCFLOW_CONV_F16X3 || !dd.wpatch16 || dd.in_norm || dd.offset || dd.wsplit_bs) return 1;
    dd.in_fmt = 0; dd.in0 = nullptr; dd.in1 = nullptr; dd.H = dd.OH; dd.W = dd.OW;
    dd.CoutPad = accflow_conv_coutpad(dd.Cout);
  }
  const accflow_conv_desc& d = dd;
  if ((!multi && (!d.in0 || !d.wpack || !d.ktab)) || d.B <= 0 || d.Cout <= 0 || d.OH <= 0 || d.OW <= 0) return 1;
  if (d.epi == ACCFLOW_EPI_TAPGEMM) {   // the result feeds the tap matrix of a following small-Cout conv inside the kernel
    if (multi || d.mode != ACCFLOW_CONV_F16X3 || !d.wpatch16 || !d.wscale16 || d.in_fmt != (d.in1 ? 3 : 1) || d.act != ACCFLOW_ACT_RELU ||
        (d.Cout & 127) || d.kws || d.stats || d.in_norm || d.cb || d.pre || d.offset || !d.tg_w16 || !d.tg_scale || !d.tg_out ||
        d.tg_rows < 1 || d.tg_rows > ACCFLOW_TAPGEMM_MAXROWS || d.tg_coutpad < 32 || (d.tg_coutpad & 31) ||
        d.tg_out_bs < (long long)d.tg_rows * d.OH * d.OW || (d.Cout > 128 && d.tg_out_ps <= 0) || !accflow_conv_direct_eligible(d) ||
        d.Kpad != accflow_conv_kpad(d.C0 + d.C1, d.KH, d.KW) || d.CoutPad != accflow_conv_coutpad(d.Cout))
      return 1;
    if ((((long long)(d.B - 1)) * d.in0_bs + accflow_s16_item_words(d.C0, d.H, d.W)) * 4 >= (1LL << 32)) return 1;
    if (d.in1 && (((long long)(d.B - 1)) * d.in1_bs + accflow_s16_item_words(d.C1, d.H, d.W)) * 4 >= (1LL << 32)) return 1;
    return accflow_launch_conv_direct(d, 2, as_stream(stream));
  }
  // the fp32 destination may be omitted only when the S16 copy is requested (GRU_ZR: that concerns out2 = r*h; z stays)
  if (!d.out && (!d.out16 || d.epi == ACCFLOW_EPI_GRU_ZR)) return 1;
  if (d.epi == ACCFLOW_EPI_GRU_ZR && !d.out2 && !d.out16) return 1;
  if ((d.in_fmt || d.out16) && d.mode != ACCFLOW_CONV_F16X3) return 1;     // S16 tensors hold the fp16 split
  if (d.in_fmt & ~3) return 1;
  if (d.out16) {   // the S16 copy exists in the epilogue's specialised (epilogue, activation) forms only
    const int ea = d.epi * 8 + d.act;
    if (ea != ACCFLOW_EPI_STORE * 8 + ACCFLOW_ACT_NONE && ea != ACCFLOW_EPI_STORE * 8 + ACCFLOW_ACT_RELU &&
        ea != ACCFLOW_EPI_STORE * 8 + ACCFLOW_ACT_SIGMOID && ea != ACCFLOW_EPI_RES_RELU * 8 + ACCFLOW_ACT_RELU &&
        ea != ACCFLOW_EPI_RES_RELU * 8 + ACCFLOW_ACT_NONE &&
        ea != ACCFLOW_EPI_GRU_ZR * 8 + ACCFLOW_ACT_SIGMOID && ea != ACCFLOW_EPI_GRU_Q * 8 + ACCFLOW_ACT_TANH &&
        ea != ACCFLOW_EPI_ACCUM * 8 + ACCFLOW_ACT_NONE)
      return 1;
  }
  if (d.cb && ((d.cb & 31) || (d.epi != ACCFLOW_EPI_STORE && d.epi != ACCFLOW_EPI_ACCUM) || d.stats || d.Cout % d.cb)) return 1;
  if (!multi && (d.Kpad != accflow_conv_kpad(d.C0 + d.C1, d.KH, d.KW) || d.CoutPad != accflow_conv_coutpad(d.Cout))) return 1;
  if ((d.epi == ACCFLOW_EPI_RES_RELU || d.epi == ACCFLOW_EPI_ACCUM) && !d.e0) return 1;
  if (d.epi == ACCFLOW_EPI_GRU_ZR && (!d.e0 || (d.Cout & 1))) return 1;
  if (d.epi == ACCFLOW_EPI_GRU_ZR && d.out16 && ((d.Cout >> 1) & 7)) return 1;
  if (d.epi == ACCFLOW_EPI_GRU_Q && (!d.e0 || !d.e1)) return 1;
  if (d.pre && d.epi != ACCFLOW_EPI_GRU_ZR && d.epi != ACCFLOW_EPI_GRU_Q) return 1;
  if (d.offset && !d.dmask) return 1;
  if ((long long)d.B * d.OH * d.OW >= (1LL << 31)) return 1;
  // sources are addressed through 32-bit buffer offsets: each must span < 4 GiB (callers chunk the batch)
  if (!multi) {
    const long long w0 = (d.in_fmt & 1) ? accflow_s16_item_words(d.C0, d.H, d.W) : (long long)d.C0 * d.H * d.W;
    const long long w1 = (d.in_fmt & 2) ? accflow_s16_item_words(d.C1, d.H, d.W) : (long long)d.C1 * d.H * d.W;
    if ((((long long)(d.B - 1)) * d.in0_bs + w0) * 4 >= (1LL << 32)) return 1;
    if (d.in1 && (((long long)(d.B - 1)) * d.in1_bs + w1) * 4 >= (1LL << 32)) return 1;
  }
  if (d.out16 && (((long long)(d.B - 1)) * d.out16_bs + accflow_s16_item_words(d.Cout, d.OH, d.OW)) * 4 >= (1LL << 32)) return 1;
  if (d.stats && (d.epi != ACCFLOW_EPI_STORE || d.act != ACCFLOW_ACT_NONE || d.stat_slots <= 0)) return 1;
  hipStream_t st = as_stream(stream);
  const long long Ptot = (long long)d.B * d.OH * d.OW;
  if (d.in_norm && !accflow_tls_dry_slots && !accflow_conv_in_norm_supported(&d)) return 1;
  if (d.stats && !accflow_tls_dry_slots) {  // the route must be the one accflow_conv_stat_slots reported for
    int want = 0;
    accflow_tls_dry_slots = &want;
    accflow_conv_desc probe = d;
    probe.stats = nullptr;
    const int prc = accflow_conv2d_f32(&probe, nullptr);
    accflow_tls_dry_slots = nullptr;
    if (prc || want != d.stat_slots) return 1;
  }
  if (multi) return accflow_launch_conv_s16m(d, -1, st);
  // the encoders' 7x7 stride-2 stem of the 3-channel image (conv_stem.hip; ACCFLOW_CONV_STEM=0: the im2col kernel, A/B)
  static const bool stem_on = [] { const char* e = getenv("ACCFLOW_CONV_STEM"); return !e || atoi(e) != 0; }();
  if (stem_on && accflow_conv_stem_eligible(d)) return accflow_launch_conv_stem(d, st);
  // S16 sources of the in0 / in1 form: the direct kernel's S16 instantiations; ACCFLOW_S16M=1 sends them to the multi-source
  // kernel instead (same results bit for bit; in the refinement loop it measured 2 - 6 % slower per launch on one box,
  // profiles/r04_ab_s16m_vs_direct.txt, so the update block stays where it was)
  static const bool s16m_on = [] { const char* e = getenv("ACCFLOW_S16M"); return e && atoi(e) != 0; }();
  if (d.in_fmt && s16m_on) {
    if (!accflow_conv_direct_eligible(d) || !d.wpatch16 || d.in_norm) return 1;
    if (d.in_fmt != (d.in1 ? 3 : 1)) return 1;
    accflow_conv_desc e = d;
    accflow_s16m_from_legacy(e);
    return accflow_launch_conv_s16m(e, -1, st);
  }
  if (d.in_fmt || d.out16 || d.cb) {
    // S16 tensors exist for the direct kernel only: its DMA loader reads them, its epilogue writes them.  No diversion
    // to another kernel whatever the grid size (the caller chose the format for this shape).
    if (!accflow_conv_direct_eligible(d) || !d.wpatch16 || d.stats || d.in_norm) return 1;
    if (d.in_fmt && d.in_fmt != (d.in1 ? 3 : 1)) return 1;
    return accflow_launch_conv_direct(d, d.Cout > 64 ? 2 : 1, st);
  }
  if (d.Cout <= 4 && !d.offset && (d.epi == ACCFLOW_EPI_STORE || d.epi == ACCFLOW_EPI_ACCUM || d.epi == ACCFLOW_EPI_RES_RELU)) {
    const bool same = d.stride == 1 && d.OH == d.H && d.OW == d.W && d.KH * d.KW >= 2 && d.C0 + d.C1 >= 16 &&
                      (8 + d.KH - 1) * (16 + d.KW - 1) <= 192 && d.KH * d.KW <= 25 && (!d.in1 || d.C0 % 16 == 0);
    ACCFLOW_DRY_RUN(0);
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
  if (accflow_conv_direct_eligible(d)) {
    const long long nb = (long long)d.B * cdiv(d.OW, DIR_TW) * cdiv(d.OH, DIR_TH);
    // with a split-K workspace small grids are split, not diverted (1x1 convolutions of < 512 channels: too few steps
    // per part to pay)
    const long long minb = (d.kws && (d.KH * d.KW >= 2 || d.C0 + d.C1 >= 512)) ? 0 : patch_min_blocks();
    if (d.Cout > 64 && nb * cdiv(d.Cout, 128) >= minb) return accflow_launch_conv_direct(d, 2, st);  // 128 ch
    if (nb * cdiv(d.Cout, 64) >= minb) return accflow_launch_conv_direct(d, 1, st);                  //  64 ch
  }
  // the im2col kernel's fp16 form needs its own pack; without it (per-batch weights, <= 32 output channels, deformable
  // fp32 kernel, ...) the call runs bf16x6 arithmetic
  if (d.mode == ACCFLOW_CONV_F16X3 && !(d.wsplit16 && !d.offset && d.Cout > 32)) {
    accflow_conv_desc e = d;
    e.mode = ACCFLOW_CONV_BF16X6;
    e.wpatch = nullptr;       // (and must not come back here)
    e.wpatch16 = nullptr; e.wsplit16 = nullptr; e.wscale16 = nullptr;
    return accflow_conv2d_f32(&e, stream);
  }
  if (d.mode == ACCFLOW_CONV_F16X3 && d.wsplit_bs) {  // per-batch-item fp16 packs (+ per-item row scales): GMA aggregation
    ACCFLOW_DRY_RUN(0);
    if ((d.OH * d.OW) % 64) return 1;
    return d.Cout > 64 ? accflow_launch_conv_bf16s(d, 2, 1, st) : accflow_launch_conv_bf16s(d, 1, 1, st);
  }
  if (d.mode == ACCFLOW_CONV_F16X3) {  // not direct-eligible (or too small a grid): im2col kernel on the fp16 pack
    accflow_conv_desc e = d;
    e.wpatch16 = nullptr;
    auto nb = [&](int bc, int bp) { return (long long)cdiv(Ptot, bp) * cdiv(e.Cout, bc); };
    if (e.Cout <= 64) return nb(64, 128) >= 384 ? accflow_launch_conv_bf16s(e, 1, 2, st) : accflow_launch_conv_bf16s(e, 1, 1, st);
    if (nb(128, 128) >= 384) return accflow_launch_conv_bf16s(e, 2, 2, st);
    if (nb(128, 64) >= 384) return accflow_launch_conv_bf16s(e, 2, 1, st);
    return accflow_launch_conv_bf16s(e, 1, 1, st);
  }
  if (d.wsplit_bs) {  // per-batch-item weights: 64-pixel tiles that never straddle items
    ACCFLOW_DRY_RUN(0);
    if (d.mode == ACCFLOW_CONV_F32 || !d.wsplit || d.offset || ((d.OH * d.OW) % 64) || d.Cout <= 32) return 1;
    return d.Cout > 64 ? accflow_launch_conv_bf16s(d, 2, 1, st) : accflow_launch_conv_bf16s(d, 1, 1, st);
  }
  if (d.mode != ACCFLOW_CONV_F32 && d.wsplit && !d.offset && d.Cout > 32) {
    // split-bf16 matrix-core path (k order must be (c, tap): the tap-major pack is deformable-only)
    auto nb = [&](int bc, int bp) { return (long long)cdiv(Ptot, bp) * cdiv(d.Cout, bc); };
    if (d.Cout <= 64) return nb(64, 128) >= 384 ? accflow_launch_conv_bf16s(d, 1, 2, st) : accflow_launch_conv_bf16s(d, 1, 1, st);
    // (a 192-channel x 128-pixel tile existed for Cout % 192 == 0 until round 5: the only such convolution of the workload,
    // convc2, runs on the direct kernel; its instantiation group took 5.6 minutes to compile - the longest unit of the build)
    if (nb(128, 128) >= 384) return accflow_launch_conv_bf16s(d, 2, 2, st);
    if (nb(128, 64) >= 384) return accflow_launch_conv_bf16s(d, 2, 1, st);
    return accflow_launch_conv_bf16s(d, 1, 1, st);
  }
  // Tile choice: the largest tile that still yields >= MIN_BLOCKS workgroups (256 CUs x ~1.5), since the
  // fusion chain runs at batch 1 (7 680 pixels) where 128x128 tiles would leave most CUs idle.
  ACCFLOW_DRY_RUN(0);  // (the fp32-MFMA kernels do not gather statistics)
  constexpr long long MIN_BLOCKS = 384;
  auto blocks = [&](int bc, int bp) { return (long long)cdiv(Ptot, bp) * cdiv(d.Cout, bc); };
  if (d.Cout <= 32) {
    if (blocks(32, 256) >= MIN_BLOCKS) return accflow_launch_conv_f32(d, 1, 4, 1, 2, st);   // 32 ch x 256 px
    return accflow_launch_conv_f32(d, 1, 4, 1, 1, st);                                       // 32 ch x 128 px
  }
  if (d.Cout <= 64) {
    if (blocks(64, 128) >= MIN_BLOCKS) return accflow_launch_conv_f32(d, 2, 2, 1, 2, st);   // 64 ch x 128 px
    return accflow_launch_conv_f32(d, 2, 2, 1, 1, st);                                       // 64 ch x 64 px
  }
  if (d.Cout % 96 == 0 && d.Cout % 128 != 0 && blocks(96, 128) >= MIN_BLOCKS)
    return accflow_launch_conv_f32(d, 1, 4, 3, 1, st);                                       // 96 ch x 128 px
  if (blocks(128, 128) >= MIN_BLOCKS) return accflow_launch_conv_f32(d, 2, 2, 2, 2, st);    // 128 ch x 128 px
  if (blocks(128, 64) >= MIN_BLOCKS) return accflow_launch_conv_f32(d, 2, 2, 2, 1, st);     // 128 ch x 64 px
  return accflow_launch_conv_f32(d, 2, 2, 1, 1, st);                                         // 64 ch x 64 px
}
