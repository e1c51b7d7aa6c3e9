#include "conv_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Direct-A patch kernel on the 16x16x32 matrix-core shape: stride-1 "same" convolutions with >= 3 taps, >= 32 input
// and > 64 output channels (the update block, AccPlus / decoder / context stacks, 96- and 128-channel encoder layers).
//
// Same design as conv2d_direct.hip (input patch of a channel chunk staged once in LDS, zero padding materialised, all
// taps read their B fragments from it; pre-split weight fragments straight from L2 into registers, double buffered;
// one barrier per chunk), re-cut for v_mfma_f32_16x16x32_{f16,bf16}:
//   * why: in-kernel timestamps on the 32x32x16 kernel (tools/kprof_conv.py, round 2) showed the chip holding
//     1.74-1.80 GHz in its loop (2.4 GHz nominal), and tools/mfma_shape_probe.hip - the same per-step operand traffic
//     and accumulator tile per wave, 3 workgroups per CU, random data - measured the 16x16x32 form 14 % faster than
//     32x32x16 at EQUAL cycles per step (1.66 vs 1.48 GHz; on all-zero data both hold 2.37 GHz): the shape decides the
//     clock the chip sustains (MI355X_MICROARCH.md, DVFS give-back item 7);
//   * K is ordered (32-channel chunk, tap, channel): one MFMA step = one tap x 32 channels, lane l supplies k-octet
//     l >> 4 of row / column l & 15, so the weight pack is [term][step][4 octets][CoutPad][8] and the LDS patch
//     [stage][term][4 octets][patch pixel] (16-B chunks; a 16-lane group reads 256 contiguous bytes: no conflicts);
//   * a workgroup = 128 output channels x 128 pixels, the 4 waves split the CHANNELS (32 each, no weight fragment is
//     loaded twice) and every wave covers all 128 pixels: 2 x 8 accumulator tiles of 16 x 16 = the same 64 registers;
//     per step 4 x 16-B weight loads and 16 x 16-B LDS reads per lane, 48 MFMAs;
//   * the pixel tile is 4 x 32, or 8 x 16 for kernels taller than 3 taps (5x1: patch 12 x 16 = 192 pixels), which
//     keeps every patch <= 208 pixels: 2 stages x 2 terms x 4 octets x 208 x 16 B = 52 KB, three workgroups per CU;
//   * the patch of the next chunk is gathered in two halves (taps 0 and 1) so that only 16 staging registers are live.
// Accumulator layout (16x16 tile, 4 registers): register r of lane l = row 4*(l >> 4) + r, column l & 15 - the
// epilogue below is written for it (stores are 4 channel rows x 16 consecutive pixels per instruction).

constexpr int D16_NPMAX = 208;

template <bool F16>
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// v = act(acc * wscale[ch] + bias[ch]) and the fused epilogues of accflow_hip.h on the 16x16 accumulator layout.
// chbase: first of this wave's 32 channels; tb: batch item of the tile (workgroup-uniform); rem[ct]: element offset of
// this lane's pixel of column tile ct inside an (OH, OW) plane, or -1.
template <int EPI, int ACT>
__device__ __forceinline__ void epilogue16(const accflow_conv_desc& d, f32x4 (&acc)[2][8], int chbase, int tb, const int (&rem)[8],
                                           int lane, int OHW, int stat_slot = 0) {
  constexpr unsigned MASKED = 0xFFFFFFFFu;
  const int lg4 = (lane >> 4) * 4;
  const int half = d.Cout >> 1;
  constexpr bool has_h = EPI != ACCFLOW_EPI_STORE, has_z = EPI == ACCFLOW_EPI_GRU_Q, zr = EPI == ACCFLOW_EPI_GRU_ZR;
  const int nout = zr ? half : d.Cout;
  auto span = [&](long long bs, int nch) { return (int)(unsigned)((((long long)(d.B - 1)) * bs + (long long)nch * OHW) * 4); };
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(d.out, 0, span(d.out_bs, nout), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_o2 =
      __builtin_amdgcn_make_buffer_rsrc(zr ? d.out2 : d.out, 0, zr ? span(d.out2_bs, half) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_e0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(has_h ? d.e0 : d.out), 0, has_h ? span(d.e0_bs, zr ? half : d.Cout) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_e1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(has_z ? d.e1 : d.out), 0, has_z ? span(d.e1_bs, d.Cout) : 0, 0x00020000);
  // the r-gate half of a GRU_ZR conv (channels >= Cout/2) is wave-uniform: Cout/2 is a multiple of 32
  const bool rgate = zr && chbase >= half;
  const unsigned OHW4 = (unsigned)OHW * 4u;
  // pre-activation addend of the GRU epilogues (accflow_conv_desc.pre)
  const bool has_pre = (zr || has_z) && d.pre != nullptr;
  const __amdgpu_buffer_rsrc_t r_pre = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(has_pre ? d.pre : d.out), 0, has_pre ? span(d.pre_bs, d.Cout) : 0, 0x00020000);
  // per group (rt, r): this lane's channel, its bias / scale, and the byte offsets of (tb, channel) in each tensor
  float hv[2][8], zv[2][8], pv[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) pv[i][ct] = 0.0f;
  auto fetch = [&](int g, float (&hh)[8], float (&zz)[8]) {
    if constexpr (has_h) {
      const int ch = chbase + (g >> 2) * 16 + lg4 + (g & 3);
      if (has_pre) {
        const unsigned bp = (unsigned)((long long)tb * d.pre_bs * 4) + (unsigned)ch * OHW4;
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
          pv[g & 1][ct] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
              r_pre, (int)((ch < d.Cout && rem[ct] >= 0) ? bp + (unsigned)rem[ct] * 4u : MASKED), 0, 0));
      }
      const int che = zr ? ch - half : ch;
      const bool live = ch < d.Cout && che >= 0;
      const unsigned b0 = (unsigned)((long long)tb * d.e0_bs * 4) + (unsigned)(live ? che : 0) * OHW4;
      const unsigned b1 = (unsigned)((long long)tb * d.e1_bs * 4) + (unsigned)ch * OHW4;
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) {
        const bool ok = live && rem[ct] >= 0;
        hh[ct] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_e0, (int)(ok ? b0 + (unsigned)rem[ct] * 4u : MASKED), 0, 0));
        if constexpr (has_z)
          zz[ct] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_e1, (int)(ok ? b1 + (unsigned)rem[ct] * 4u : MASKED), 0, 0));
      }
    }
  };
  float bias[8], scl[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int ch = min(chbase + (g >> 2) * 16 + lg4 + (g & 3), d.Cout - 1);
    bias[g] = d.bias ? d.bias[ch] : 0.0f;
    scl[g] = d.wscale16 ? d.wscale16[ch] : 1.0f;
  }
  fetch(0, hv[0], zv[0]);
  constexpr bool CAN_STATS = EPI == ACCFLOW_EPI_STORE && ACT == ACCFLOW_ACT_NONE;
  float stat_n = 0.0f;
  if constexpr (CAN_STATS) {
    if (d.stats) {
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) stat_n += rem[ct] >= 0 ? 1.0f : 0.0f;
      stat_n = row16_sum(stat_n);
    }
  }
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    if (g + 1 < 8) fetch(g + 1, hv[(g + 1) & 1], zv[(g + 1) & 1]);  // operands of the next group before this group's stores
    const int rt = g >> 2, r = g & 3;
    const int ch = chbase + rt * 16 + lg4 + r;
    const bool in = ch < d.Cout;
    if constexpr (CAN_STATS) {
      if (d.stats) {  // {sum, M2, n} over the tile's 128 pixels of this 16-lane row's channel
        float s = 0.0f;
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) s += rem[ct] >= 0 ? fmaf(acc[rt][ct][r], scl[g], bias[g]) : 0.0f;
        s = row16_sum(s);
        const float mean = s / fmaxf(stat_n, 1.0f);
        float q = 0.0f;
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
          const float dv = fmaf(acc[rt][ct][r], scl[g], bias[g]) - mean;
          q += rem[ct] >= 0 ? dv * dv : 0.0f;
        }
        q = row16_sum(q);
        if ((lane & 15) == 0 && in) stat_store(d, tb, ch, stat_slot, s, q, stat_n);
      }
    }
    const unsigned bo = (unsigned)((long long)tb * (rgate ? d.out2_bs : d.out_bs) * 4) + (unsigned)(rgate ? ch - half : ch) * OHW4;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      const float v = apply_act(fmaf(acc[rt][ct][r], scl[g], bias[g]) + pv[g & 1][ct], ACT);
      float hh = 0.0f, zz = 0.0f;
      if constexpr (has_h) hh = hv[g & 1][ct];
      if constexpr (has_z) zz = zv[g & 1][ct];
      float o = v;
      if constexpr (EPI == ACCFLOW_EPI_RES_RELU) o = fmaxf(hh + v, 0.0f);
      else if constexpr (EPI == ACCFLOW_EPI_GRU_Q) o = (1.0f - zz) * hh + zz * v;
      else if constexpr (EPI == ACCFLOW_EPI_ACCUM) o = hh + v;
      else if constexpr (zr) o = rgate ? v * hh : v;
      const unsigned off = (in && rem[ct] >= 0) ? bo + (unsigned)rem[ct] * 4u : MASKED;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rgate ? r_o2 : r_out, (int)off, 0, 0);
    }
  }
}

template <int NT, bool F16>
__global__ __launch_bounds__(256, 3) void conv2d_direct16_kernel(const accflow_conv_desc d) {
  static_assert(!F16 || NT == 2, "the fp16 split has two terms");
  extern __shared__ u32x4 Pst[];                // [2 stages][NT][4 octets][NP patch pixels], sized at launch

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, ko = lane >> 4;
  const int cblk0 = blockIdx.y * 128;
  const int chbase = cblk0 + wave * 32;
  const int OHW = d.OH * d.OW;
  const int TH = d.KH > 3 ? 8 : 4, TW = 128 / TH;   // 8 x 16 pixel tiles for tall kernels: the patch stays <= 208 pixels
  const int tilesX = (d.OW + TW - 1) / TW, tilesY = (d.OH + TH - 1) / TH;
  const int tb = blockIdx.x / (tilesX * tilesY), trem = blockIdx.x - tb * tilesX * tilesY;
  const int oy0 = (trem / tilesX) * TH, ox0 = (trem % tilesX) * TW;
  const int T = d.KH * d.KW;
  const int PW = TW + d.KW - 1, NP = (TH + d.KH - 1) * PW;
  const int PSTAGE = NT * 4 * NP;
  const int Cin = d.C0 + d.C1;
  const int nchunk = (Cin + 31) / 32, nstep = nchunk * T;
  const int HW = d.H * d.W;
  const int c_begin = (int)((long long)nchunk * blockIdx.z / gridDim.z);
  const int c_end = (int)((long long)nchunk * (blockIdx.z + 1) / gridDim.z);
  const int step_end = c_end * T;
  const bool busy = chbase < d.Cout;   // a wave whose 32 channels are all padding (Cout = 96) only helps staging

  // ---- patch staging: item it = tid + 256*i (i < 4) -> (octet = it / NP, patch pixel = it % NP); two items at a time
  const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.in0), 0, (int)(unsigned)((((long long)(d.B - 1)) * d.in0_bs + (long long)d.C0 * HW) * 4),
      0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.in1 ? d.in1 : d.in0), 0,
      (int)(unsigned)(d.in1 ? (((long long)(d.B - 1)) * d.in1_bs + (long long)d.C1 * HW) * 4 : 0), 0x00020000);
  unsigned voff[4];      // byte offset of (b, iy, ix) relative to its source's batch item 0, 0xFFFFFFFF in the zero padding
  int pinfo[4];          // (octet << 16) | patch pixel, -1 for a dead item
  const unsigned bb0 = (unsigned)((long long)tb * d.in0_bs * 4), bb1 = (unsigned)((long long)tb * d.in1_bs * 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int it = tid + 256 * i;
    const bool live = it < 4 * NP;
    const int oct = live ? it / NP : 0;
    const int pix = live ? it - oct * NP : 0;
    const int py = pix / PW, px = pix - py * PW;
    const int iy = oy0 - d.padH + py, ix = ox0 - d.padW + px;
    const bool ok = live && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
    voff[i] = ok ? (unsigned)((iy * d.W + ix) * 4) : 0xFFFFFFFFu;
    pinfo[i] = live ? (oct << 16) | pix : -1;
  }
  float xa[8], xb[8];
  auto gather2 = [&](int cc, int i0) {   // items i0, i0 + 1 of chunk cc
    const int c0 = cc * 32;              // a chunk never straddles the two sources (C0 % 32 == 0 when in1 is given)
    const bool second = c0 >= d.C0;
    const __amdgpu_buffer_rsrc_t rs = second ? rsrc1 : rsrc0;
    const int cs = second ? c0 - d.C0 : c0, cmax = second ? d.C1 : d.C0;
    const unsigned bb = second ? bb1 : bb0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int ca = cs + ((pinfo[i0] >> 16) & 3) * 8 + q, cb = cs + ((pinfo[i0 + 1] >> 16) & 3) * 8 + q;
      const unsigned oa = (ca < cmax && voff[i0] != 0xFFFFFFFFu) ? bb + voff[i0] + (unsigned)ca * (unsigned)HW * 4u : 0xFFFFFFFFu;
      const unsigned ob = (cb < cmax && voff[i0 + 1] != 0xFFFFFFFFu) ? bb + voff[i0 + 1] + (unsigned)cb * (unsigned)HW * 4u : 0xFFFFFFFFu;
      xa[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)oa, 0, 0));
      xb[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)ob, 0, 0));
    }
  };
  bool bad = false;
  constexpr float ASC = (float)(1 << ACCFLOW_F16_ASHIFT);
  auto store2 = [&](int stage, int i0) {
    u32x4 terms[NT];
    if constexpr (F16) split8_f16<0>(xa, terms, bad, ASC);
    else split8_bf16<NT, 0>(xa, terms);
    if (pinfo[i0] >= 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) Pst[stage * PSTAGE + (t * 4 + (pinfo[i0] >> 16)) * NP + (pinfo[i0] & 0xFFFF)] = terms[t];
    }
    if constexpr (F16) split8_f16<0>(xb, terms, bad, ASC);
    else split8_bf16<NT, 0>(xb, terms);
    if (pinfo[i0 + 1] >= 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        Pst[stage * PSTAGE + (t * 4 + (pinfo[i0 + 1] >> 16)) * NP + (pinfo[i0 + 1] & 0xFFFF)] = terms[t];
    }
  };

  // ---- A fragments: lane l -> row l & 15 of a 16-channel tile, k-octet l >> 4; 16 bytes per (term, row tile) ----
  const long long step_bytes = 4LL * d.CoutPad * 16, term_bytes = (long long)nstep * step_bytes;
  const __amdgpu_buffer_rsrc_t rsrcw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(F16 ? d.wpatch32_16 : d.wpatch32), 0, (int)(unsigned)(3 * term_bytes), 0x00020000);
  const unsigned avoff = (unsigned)((ko * d.CoutPad + chbase + l15) * 16);
#define D16_LOAD_A(STEP, A)                                                                                      \
  _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)                \
      A[t][rt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(                              \
          rsrcw, (int)(avoff + rt * 256), (int)(unsigned)(t * term_bytes + (STEP) * step_bytes), 0))

  // this lane's pixel of column tile ct inside the patch at tap (0, 0) = pb0 + ctoff(ct), the latter wave-uniform
  // (kept in scalar registers: the kernel sits at the 168-register limit of three waves per SIMD)
  const int pb0 = l15 + ko * NP;
  const int ctrow = TW == 32 ? 1 : 0;  // column tiles per tile row - 1
  auto ctoff = [&](int ct) { return ctrow ? (ct >> 1) * PW + (ct & 1) * 16 : ct * PW; };

  f32x4 acc[2][8];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[rt][ct][r] = 0.0f;

  bf16x8 aA[NT][2], aB[NT][2];
  D16_LOAD_A(c_begin * T, aA);
  gather2(c_begin, 0);
  store2(c_begin & 1, 0);
  gather2(c_begin, 2);
  store2(c_begin & 1, 2);
  __syncthreads();

  int cc = c_begin, tap = 0, ty = 0, tx = 0;
#define D16_STEP(STEP, ACUR, ANXT)                                                                               \
  do {                                                                                                           \
    const int pstage = cc & 1;                                                                                   \
    const bool next_chunk = cc + 1 < c_end;                                                                      \
    if ((STEP) + 1 < step_end) { D16_LOAD_A((STEP) + 1, ANXT); }                                                 \
    if (next_chunk) {                                                                                            \
      if (tap == 0) gather2(cc + 1, 0);                                                                          \
      else if (tap == 1) { store2(pstage ^ 1, 0); gather2(cc + 1, 2); }                                          \
      else if (tap == 2) store2(pstage ^ 1, 2);                                                                  \
    }                                                                                                            \
    if (busy) {                                                                                                  \
      const int toff = pstage * PSTAGE + ty * PW + tx;                                                           \
      _Pragma("unroll") for (int h = 0; h < 4; ++h) {  /* a quarter of the pixels at a time: 8 B registers */     \
        bf16x8 b[NT][2];                                                                                         \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int p = 0; p < 2; ++p)             \
            b[t][p] = __builtin_bit_cast(bf16x8, Pst[toff + t * (4 * NP) + ctoff(h * 2 + p) + pb0]);             \
        constexpr int NPAIR = NT == 3 ? 6 : 3;                                                                   \
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};                                    \
        _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) _Pragma("unroll") for (int pr = 6 - NPAIR; pr < 6; ++pr) \
            _Pragma("unroll") for (int p = 0; p < 2; ++p) acc[rt][h * 2 + p] = mfma16<F16>(                      \
                ACUR[PA[pr]][rt], b[PB[pr]][p], acc[rt][h * 2 + p]);                                             \
      }                                                                                                          \
    }                                                                                                            \
    if (++tx == d.KW) { tx = 0; ++ty; }                                                                          \
    if (++tap == T) {                                                                                            \
      __syncthreads();                                                                                           \
      tap = 0; ty = 0; tx = 0; ++cc;                                                                             \
    }                                                                                                            \
  } while (0)

#ifdef ACCFLOW_KPROF
  const unsigned long long tK0 = __builtin_readcyclecounter(), tR0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int step = c_begin * T; step < step_end; step += 2) {
    D16_STEP(step, aA, aB);
    if (step + 1 < step_end) D16_STEP(step + 1, aB, aA);
  }
#ifdef ACCFLOW_KPROF
  if (tid == 0) {
    KP_SLOT(0) = __builtin_readcyclecounter() - tK0;
    KP_SLOT(1) = __builtin_amdgcn_s_memrealtime() - tR0;
    KP_SLOT(2) = step_end - c_begin * T;
    KP_SLOT(10) = 1;
  }
#endif
#undef D16_STEP
#undef D16_LOAD_A
  if constexpr (F16) {
    if (bad && d.guard) atomicOr(d.guard, 1);
  }
  int rem[8];
#pragma unroll
  for (int ct = 0; ct < 8; ++ct) {
    const int j = ct * 16 + l15;
    const int oy = oy0 + j / TW, ox = ox0 + j % TW;
    rem[ct] = (oy < d.OH && ox < d.OW) ? oy * d.OW + ox : -1;
  }
  if (!busy) return;
  if (gridDim.z > 1) {  // raw partial sums of this K-part; conv_ksplit_reduce_kernel applies scale / bias / act / epilogue
    accflow_conv_desc e = d;
    e.out = d.kws + (long long)blockIdx.z * d.B * d.Cout * OHW;
    e.out_bs = (long long)d.Cout * OHW;
    e.bias = nullptr;
    e.wscale16 = nullptr;
    epilogue16<ACCFLOW_EPI_STORE, ACCFLOW_ACT_NONE>(e, acc, chbase, tb, rem, lane, OHW);
    return;
  }
#define D16_EPI_CASE(E, A) \
  case (E) * 8 + (A): epilogue16<E, A>(d, acc, chbase, tb, rem, lane, OHW, trem); break;
  switch (d.epi * 8 + d.act) {
    D16_EPI_CASE(ACCFLOW_EPI_STORE, ACCFLOW_ACT_NONE)
    D16_EPI_CASE(ACCFLOW_EPI_STORE, ACCFLOW_ACT_RELU)
    D16_EPI_CASE(ACCFLOW_EPI_STORE, ACCFLOW_ACT_SIGMOID)
    D16_EPI_CASE(ACCFLOW_EPI_STORE, ACCFLOW_ACT_TANH)
    D16_EPI_CASE(ACCFLOW_EPI_RES_RELU, ACCFLOW_ACT_RELU)
    D16_EPI_CASE(ACCFLOW_EPI_RES_RELU, ACCFLOW_ACT_NONE)
    D16_EPI_CASE(ACCFLOW_EPI_GRU_ZR, ACCFLOW_ACT_SIGMOID)
    D16_EPI_CASE(ACCFLOW_EPI_GRU_Q, ACCFLOW_ACT_TANH)
    D16_EPI_CASE(ACCFLOW_EPI_ACCUM, ACCFLOW_ACT_NONE)
    default: break;  // (the launcher only sends the combinations above)
  }
#undef D16_EPI_CASE
}

}  // namespace

bool accflow_conv_direct16_eligible(const accflow_conv_desc& d) {
  const bool f16 = d.mode == ACCFLOW_CONV_F16X3;
  if (f16 ? !(d.wpatch32_16 && d.wscale16) : !d.wpatch32) return false;
  if (d.wsplit_bs || d.mode == ACCFLOW_CONV_F32 || d.offset || d.stride != 1 || d.Cout <= 64) return false;
  if (d.OH != d.H || d.OW != d.W) return false;
  const int T = d.KH * d.KW, TH = d.KH > 3 ? 8 : 4, TW = 128 / TH;
  if (T < 3 || (TH + d.KH - 1) * (TW + d.KW - 1) > D16_NPMAX) return false;
  // (measured, same box, back to back: 3x3 +1 %, 1x5 +1..5 %, 96 channels +6 %, but 5x1 on the 8 x 16 tile -4 % against
  // the 32x32x16 kernel: tall kernels stay there; the 8 x 16 path is kept for ACCFLOW_DIRECT16=2)
  static const bool tall = [] { const char* e = getenv("ACCFLOW_DIRECT16"); return e && atoi(e) == 2; }();
  if (d.KH > 3 && !tall) return false;
  if (d.C0 + d.C1 < 32) return false;
  if (d.in1 && (d.C0 % 32)) return false;
  switch (d.epi * 8 + d.act) {
    case ACCFLOW_EPI_STORE * 8 + ACCFLOW_ACT_NONE: case ACCFLOW_EPI_STORE * 8 + ACCFLOW_ACT_RELU:
    case ACCFLOW_EPI_STORE * 8 + ACCFLOW_ACT_SIGMOID: case ACCFLOW_EPI_STORE * 8 + ACCFLOW_ACT_TANH:
    case ACCFLOW_EPI_RES_RELU * 8 + ACCFLOW_ACT_RELU: case ACCFLOW_EPI_RES_RELU * 8 + ACCFLOW_ACT_NONE:
    case ACCFLOW_EPI_GRU_ZR * 8 + ACCFLOW_ACT_SIGMOID: case ACCFLOW_EPI_GRU_Q * 8 + ACCFLOW_ACT_TANH:
    case ACCFLOW_EPI_ACCUM * 8 + ACCFLOW_ACT_NONE: break;
    default: return false;
  }
  if (d.epi == ACCFLOW_EPI_GRU_ZR && ((d.Cout >> 1) % 32)) return false;
  // byte offsets inside one tensor are 32-bit
  const long long OHW = (long long)d.OH * d.OW;
  if ((((long long)(d.B - 1)) * d.out_bs + (long long)d.Cout * OHW) * 4 >= (1LL << 32)) return false;
  return true;
}

int conv_ksplit_reduce_launch(const accflow_conv_desc& d, int Z, hipStream_t st);

int accflow_launch_conv_direct16(const accflow_conv_desc& d, hipStream_t st) {
  const int TH = d.KH > 3 ? 8 : 4, TW = 128 / TH;
  const int tiles = cdiv(d.OW, TW) * cdiv(d.OH, TH);
  const long long nb = (long long)d.B * tiles * cdiv(d.Cout, 128);
  ACCFLOW_DRY_RUN(tiles);
  int Z = 1;
  const long long nout = (long long)d.B * d.Cout * d.OH * d.OW;
  const int nchunk = (d.C0 + d.C1 + 31) / 32;
  if (d.kws && nb < 320 && !d.stats) {  // split-K for grids that leave most of the 256 CUs idle (the batch-1 fusion chain)
    Z = (int)((512 + nb - 1) / nb);
    if (Z > 4) Z = 4;
    if (Z > nchunk / 2) Z = nchunk / 2;
    if ((long long)Z * nout > d.kws_elems) Z = (int)(d.kws_elems / nout);
    if (Z < 1) Z = 1;
  }
  dim3 grid((unsigned)((long long)d.B * tiles), cdiv(d.Cout, 128), Z);
  const int NP = (TH + d.KH - 1) * (TW + d.KW - 1);
  // the three-term form needs more than the 64 KB a kernel may use without asking (set once per process: a constant)
  static const int big_lds = [] {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_direct16_kernel<3, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 3 * 4 * D16_NPMAX * 16);
  }();
  if (d.mode == ACCFLOW_CONV_F16X3) hipLaunchKernelGGL((conv2d_direct16_kernel<2, true>), grid, dim3(256), 2 * 2 * 4 * NP * 16, st, d);
  else if (d.mode == ACCFLOW_CONV_BF16X3) hipLaunchKernelGGL((conv2d_direct16_kernel<2, false>), grid, dim3(256), 2 * 2 * 4 * NP * 16, st, d);
  else {
    if (big_lds) return big_lds;
    hipLaunchKernelGGL((conv2d_direct16_kernel<3, false>), grid, dim3(256), 2 * 3 * 4 * NP * 16, st, d);
  }
  if (Z > 1) return conv_ksplit_reduce_launch(d, Z, st);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

#ifdef ACCFLOW_KPROF
extern "C" int accflow_debug_kprof16(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kprof), 4096 * 16 * 8);
  if (reset) { void* p; hipGetSymbolAddress(&p, HIP_SYMBOL(g_kprof)); hipMemset(p, 0, 4096 * 16 * 8); }
  return (int)hipGetLastError();
}
#endif
