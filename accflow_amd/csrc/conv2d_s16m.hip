// Host side of the multi-source S16 convolution kernel (conv_s16m_kernel.h): validation, wave-layout choice, split-K,
// and its weight pack (accflow_conv_pack_multi16).
#include "conv_s16m_kernel.h"

namespace {

struct multi_pack_args {
  const float* w[ACCFLOW_CONV_MAX_SRC];
  int rowlen[ACCFLOW_CONV_MAX_SRC];
  int nsrc;
};

// wscale16[ch] = 2^-(k + ACCFLOW_F16_ASHIFT), k such that the row's largest |w * scale| over ALL sources * 2^k lies in
// [2^10, 2^11) (conv_row_scale16_kernel's rule, conv2d.hip): one workgroup per output channel
__global__ __launch_bounds__(256) void conv_row_scale_multi_kernel(const multi_pack_args a, const float* __restrict__ scale, int Cout,
                                                                   float* __restrict__ wscale16) {
  __shared__ float red[256];
  const int ch = blockIdx.x;
  float m = 0.0f;
  if (ch < Cout)
    for (int s = 0; s < a.nsrc; ++s)
      for (int j = threadIdx.x; j < a.rowlen[s]; j += 256) m = fmaxf(m, fabsf(a.w[s][(long long)ch * a.rowlen[s] + j]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x) return;
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
  wscale16[ch] = ldexpf(1.0f, -(k + ACCFLOW_F16_ASHIFT));
}

// one source's steps [step0, step0 + ceil(C/16) * T) of the pack [term][step][2 octets][CoutPad][8]
__global__ __launch_bounds__(256) void conv_pack_multi16_kernel(const float* __restrict__ w, const float* __restrict__ scale, int Cout,
                                                                int C, int T, int CoutPad, long long per_term, long long step0,
                                                                unsigned short* __restrict__ wp,
                                                                const float* __restrict__ wscale16) {
  const long long n = (long long)((C + 15) / 16) * T * 2 * CoutPad * 8;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int q = (int)(idx & 7);
  long long r = idx >> 3;
  const int ch = (int)(r % CoutPad); r /= CoutPad;
  const int o = (int)(r & 1); r >>= 1;
  const int step = (int)r, g = step / T, tap = step - g * T;
  const int c = g * 16 + o * 8 + q;
  float val = 0.0f;
  if (c < C && ch < Cout) {
    val = w[((long long)ch * C + c) * T + tap];
    if (scale) val *= scale[ch];
  }
  float rr = val * (ldexpf(1.0f, -ACCFLOW_F16_ASHIFT) / wscale16[ch]);
  const long long dst = step0 * 2 * CoutPad * 8 + idx;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const _Float16 hq = t < 2 ? (_Float16)rr : (_Float16)0.0f;
    wp[t * per_term + dst] = __builtin_bit_cast(unsigned short, hq);
    rr -= (float)hq;
  }
}

long long multi_steps(int nsrc, const int* C, const int* KH, const int* KW) {
  long long n = 0;
  for (int s = 0; s < nsrc; ++s) n += (long long)((C[s] + 15) / 16) * KH[s] * KW[s];
  return n;
}

template <int TH>
bool s16m_fits(const accflow_conv_desc& d) {
  for (int s = 0; s < d.nsrc; ++s) {
    const accflow_conv_src& S = d.src[s];
    if ((TH + S.KH - 1) * (S16M_TW + S.KW - 1) > s16m_cap(TH) / 4) return false;
  }
  return true;
}

template <int LAY>
int s16m_launch(const accflow_conv_desc& d, hipStream_t st) {
  using L = s16m_lay<LAY>;
  constexpr int BC = L::WC * L::TCW * 32;
  const int tiles = cdiv(d.OW, S16M_TW) * cdiv(d.OH, L::TH);
  const long long nb = (long long)d.B * tiles * cdiv(d.Cout, BC);
  ACCFLOW_DRY_RUN(tiles * L::WP);  // one statistics slot per wave along the pixels
  int nchunk = 0;
  long long nstep = 0;
  for (int s = 0; s < d.nsrc; ++s) {
    const s16m_geom g = s16m_geometry<L::TH>(d.src[s]);
    nchunk += g.nch;
    nstep += (long long)g.n16 * g.T;
  }
  // split-K: the rules of launch_conv_direct (conv2d_direct.hip)
  int Z = 1;
  const long long nout = (long long)d.B * d.Cout * d.OH * d.OW;
  if (d.kws && nchunk >= 256 && nb < 700 && !d.stats) {
    double best = 0.0;
    for (int z = 1; z <= 8; ++z) {
      const double fill = (double)(nb * z) / (double)(cdiv(nb * z, 768) * 768);
      if (fill > best + 0.03) { best = fill; Z = z; }
    }
  } else if (d.kws && nb < 320 && !d.stats) {
    Z = (int)((512 + nb - 1) / nb);
    if (Z > 4) Z = 4;
    if (Z > nchunk / 2) Z = nchunk / 2;
  }
  if (Z > 1 && (long long)Z * nout > d.kws_elems) Z = (int)(d.kws_elems / nout);
  if (Z < 1) Z = 1;
  if (d.out16 && d.epi == ACCFLOW_EPI_GRU_ZR && ((d.Cout >> 1) & 7)) Z = 1;
  if (d.e0_fmt) Z = 1;   // (the split-K reduce kernel reads an fp32 residual)
  if (d.split_c0) Z = 1; // (two convolutions with their own reduction lengths and activations: the reduce kernel knows one)
  if (d.e0_fmt && (d.Cout % (L::TCW * 32))) return 1;   // every wave's rows all present, or all absent
  dim3 grid((unsigned)((long long)d.B * tiles), cdiv(d.Cout, BC), Z);
  // the tap-specialised loop when every source is 3x3 / step 1 (ACCFLOW_S16M_KT9=0: the generic loop, A/B)
  static const bool kt9_on = [] { const char* e = getenv("ACCFLOW_S16M_KT9"); return !e || atoi(e) != 0; }();
  bool kt9 = kt9_on && !d.split_c0;
  for (int s = 0; s < d.nsrc && kt9; ++s) kt9 = d.src[s].KH == 3 && d.src[s].KW == 3 && d.src[s].step == 1;
  int rc;
  if (LAY == 0) rc = accflow_s16m_launch_0(d, grid, st, kt9);
  else if (LAY == 1) rc = accflow_s16m_launch_1(d, grid, st, kt9);
  else if (LAY == 2) rc = accflow_s16m_launch_2(d, grid, st, kt9);
  else rc = accflow_s16m_launch_3(d, grid, st, kt9);
  if (rc) return rc;
  if (Z > 1) return conv_ksplit_reduce_launch(d, Z, st);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

int s16m_launch_lay(const accflow_conv_desc& d, int lay, hipStream_t st) {
  switch (lay) {
    case 0: return s16m_launch<0>(d, st);
    case 1: return s16m_launch<1>(d, st);
    case 2: return s16m_launch<2>(d, st);
    default: return s16m_launch<3>(d, st);
  }
}

}  // namespace

extern "C" long long accflow_conv_multi_pack_elems(int Cout, int nsrc, const int* C, const int* KH, const int* KW) {
  if (nsrc < 1 || nsrc > ACCFLOW_CONV_MAX_SRC || !C || !KH || !KW) return 0;
  return 3LL * multi_steps(nsrc, C, KH, KW) * 2 * accflow_conv_coutpad(Cout) * 8;
}

extern "C" int accflow_conv_pack_multi16(const float* const* w, const float* scale, int Cout, int nsrc, const int* C,
                                         const int* KH, const int* KW, void* wpatch16, float* wscale16, void* stream) {
  if (!w || !C || !KH || !KW || !wpatch16 || !wscale16 || Cout <= 0 || nsrc < 1 || nsrc > ACCFLOW_CONV_MAX_SRC) return 1;
  multi_pack_args a;
  a.nsrc = nsrc;
  for (int s = 0; s < ACCFLOW_CONV_MAX_SRC; ++s) { a.w[s] = nullptr; a.rowlen[s] = 0; }
  for (int s = 0; s < nsrc; ++s) {
    if (!w[s] || C[s] <= 0 || KH[s] <= 0 || KW[s] <= 0) return 1;
    a.w[s] = w[s];
    a.rowlen[s] = C[s] * KH[s] * KW[s];
  }
  const int CoutPad = accflow_conv_coutpad(Cout);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(conv_row_scale_multi_kernel, dim3(CoutPad), dim3(256), 0, st, a, scale, Cout, wscale16);
  const long long per_term = multi_steps(nsrc, C, KH, KW) * 2 * CoutPad * 8;
  long long step0 = 0;
  for (int s = 0; s < nsrc; ++s) {
    const int T = KH[s] * KW[s];
    const long long n = (long long)((C[s] + 15) / 16) * T * 2 * CoutPad * 8;
    hipLaunchKernelGGL(conv_pack_multi16_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, w[s], scale, Cout, C[s], T, CoutPad,
                       per_term, step0, reinterpret_cast<unsigned short*>(wpatch16), wscale16);
    step0 += (long long)((C[s] + 15) / 16) * T;
  }
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

// the in0 / in1 S16 form as sources (same pack: accflow_conv_pack_patch16's step order is (16-channel group, tap) over the
// concatenated channels, and a chunk never straddles the two sources - C0 % 16 == 0, % 32 for 1x1)
void accflow_s16m_from_legacy(accflow_conv_desc& e) {
  e.nsrc = e.in1 ? 2 : 1;
  for (int s = 0; s < e.nsrc; ++s) {
    accflow_conv_src& S = e.src[s];
    S.ptr = s ? (const void*)e.in1 : (const void*)e.in0;
    S.bs = s ? e.in1_bs : e.in0_bs;
    S.C = s ? e.C1 : e.C0;
    S.Hs = e.H; S.Ws = e.W;
    S.step = 1; S.oy = 0; S.ox = 0;
    S.KH = e.KH; S.KW = e.KW; S.padH = e.padH; S.padW = e.padW;
    S.reserved = 0;
  }
}

bool accflow_conv_s16m_eligible(const accflow_conv_desc& d) {
  if (d.nsrc < 1 || d.nsrc > ACCFLOW_CONV_MAX_SRC || !d.wpatch16 || !d.wscale16 || d.mode != ACCFLOW_CONV_F16X3) return false;
  if (d.wsplit_bs || d.offset || d.in_norm || d.Cout <= 4) return false;
  // an S16 residual operand exists in the lean residual epilogue only (every wave's 32 * TCW rows present)
  if (d.e0_fmt && (d.epi != ACCFLOW_EPI_RES_RELU || d.act != ACCFLOW_ACT_RELU || d.cb || (d.Cout % 32) || !d.e0)) return false;
  for (int s = 0; s < d.nsrc; ++s) {
    const accflow_conv_src& S = d.src[s];
    if (!S.ptr || S.C <= 0 || S.Hs <= 0 || S.Ws <= 0 || (S.step != 1 && S.step != 2) || S.KH <= 0 || S.KW <= 0) return false;
    if (S.oy < 0 || S.ox < 0 || S.oy >= S.step || S.ox >= S.step) return false;
    if (S.padH < 0 || S.padW < 0 || S.padH >= S.KH || S.padW >= S.KW) return false;
    // 32-bit buffer offsets: each source must span < 4 GiB
    if ((((long long)(d.B - 1)) * S.bs + accflow_s16_item_words(S.C, S.Hs, S.Ws)) * 4 >= (1LL << 32)) return false;
    // a 1x1 source of the 4-octet chunk form ends on a 32-channel boundary unless it is the last (a chunk never straddles)
  }
  return s16m_fits<4>(d);
}

// lay < 0: automatic
int accflow_launch_conv_s16m(const accflow_conv_desc& d, int lay, hipStream_t st) {
  if (!accflow_conv_s16m_eligible(d)) return 1;
  static const int env_lay = [] { const char* e = getenv("ACCFLOW_S16M_LAY"); return e ? atoi(e) : -1; }();
  if (lay < 0 && d.src[0].reserved > 0) lay = d.src[0].reserved - 1;   // (tests / tuning: accflow_conv_src.reserved)
  if (lay < 0) lay = env_lay;
  const bool fits8 = s16m_fits<8>(d);
  if (d.split_c0) {
    // accflow_conv_desc.split_c0: the channel block must not straddle the two convolutions - 96 -> the 96-channel layout,
    // 128 -> the 128-channel one, 64 -> a 64-channel one; both halves whole blocks (the lean epilogue, or the general one
    // with act NONE for both: the InstanceNorm encoder's raw outputs + statistics)
    const int c0 = d.split_c0, c1 = d.Cout - d.split_c0;
    if (d.epi != ACCFLOW_EPI_STORE || d.cb || (d.act != ACCFLOW_ACT_NONE && (d.act != ACCFLOW_ACT_RELU || d.stats)) || c1 <= 0)
      return 1;
    int bc;
    if (c0 == 96 && fits8) { lay = 3; bc = 96; }
    else if (c0 == 128) { lay = 0; bc = 128; }
    else if (c0 == 64) { lay = (fits8 && (long long)d.B * cdiv(d.OW, S16M_TW) * cdiv(d.OH, 8) >= 1536) ? 1 : 2; bc = 64; }
    else return 1;
    if (c1 % bc) return 1;
    return s16m_launch_lay(d, lay, st);
  }
  const long long tiles8 = (long long)d.B * cdiv(d.OW, S16M_TW) * cdiv(d.OH, 8);
  if ((lay == 1 || lay == 3) && !fits8) lay = -1;
  if (lay >= 0 && lay <= 3) return s16m_launch_lay(d, lay, st);
  // enough workgroups to fill the chip at 3 per CU
  const bool big8 = fits8 && tiles8 >= 1536 && !(d.kws && tiles8 < 320);
  if (d.Cout <= 64) return s16m_launch_lay(d, big8 ? 1 : 2, st);
  // 96-channel blocks: 96 -> 96 (encoder layer2) and 192 = 2 x 96 (convc2: 212 vs 227 us as 128 + 64, profiles/r04_s16m_bench.txt)
  if ((d.Cout <= 96 || d.Cout == 192) && fits8 && tiles8 >= 600 && !(d.kws && tiles8 < 320)) return s16m_launch_lay(d, 3, st);
  // Cout = 128 m + 64 with a pointwise epilogue: 128 m channels on the 128-channel layout, the last 64 on a 64-channel one
  // (accflow_launch_conv_direct's rule, conv2d_direct.hip)
  const bool pointwise = d.epi == ACCFLOW_EPI_STORE || d.epi == ACCFLOW_EPI_RES_RELU || d.epi == ACCFLOW_EPI_ACCUM;
  const long long tiles4 = (long long)d.B * cdiv(d.OW, S16M_TW) * cdiv(d.OH, 4);
  if (d.Cout > 128 && d.Cout % 128 > 0 && d.Cout % 128 <= 64 && pointwise && !d.stats && !accflow_tls_dry_slots && !d.cb &&
      !(d.kws && tiles4 * cdiv(d.Cout, 128) < 320)) {
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
    b.wpatch16 = reinterpret_cast<const char*>(d.wpatch16) + (long long)ch0 * 16;
    const int rc = s16m_launch_lay(a, 0, st);
    if (rc) return rc;
    return s16m_launch_lay(b, 2, st);
  }
  return s16m_launch_lay(d, 0, st);
}
