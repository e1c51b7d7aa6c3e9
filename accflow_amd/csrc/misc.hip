// Normalisation and small element-wise kernels of the path.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.0f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// nn.InstanceNorm2d defaults (extractor.py:36-39): per-(n,c) plane, biased variance, no affine.
// One workgroup per plane; two-pass mean / variance (planes are L2-resident on the re-read).
__global__ __launch_bounds__(512) void instance_norm_kernel(const float* x, const float* res,
                                                            float* out, int HW, float eps, int mode) {
  __shared__ float red[8];
  const long long base = (long long)blockIdx.x * HW;
  const float* p = x + base;
  const bool vec = (HW & 3) == 0;
  float s = 0.0f;
  if (vec) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
    for (int i = threadIdx.x; i < HW / 4; i += blockDim.x) {
      const float4 v = p4[i];
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += blockDim.x) s += p[i];
  }
  const float mean = block_sum(s, red) / (float)HW;
  float q = 0.0f;
  if (vec) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
    for (int i = threadIdx.x; i < HW / 4; i += blockDim.x) {
      const float4 v = p4[i];
      const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
      const float a = p[i] - mean;
      q += a * a;
    }
  }
  const float var = block_sum(q, red) / (float)HW;
  const float invstd = 1.0f / sqrtf(var + eps);
  const float* r = res ? res + base : nullptr;
  float* o = out + base;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) {
    float v = (p[i] - mean) * invstd;
    if (mode >= 1) v = fmaxf(v, 0.0f);
    if (mode == 2) v = fmaxf(r[i] + v, 0.0f);
    o[i] = v;
  }
}

// {mean, 1/sqrt(var + eps)} of every (b, c) plane from the partial records {sum, M2, n} the convolution epilogues wrote
// (accflow_conv_desc.stats): one 64-lane workgroup per plane; lane l combines slots l, l + 64, ... in order and the 64
// lane results are merged by a fixed butterfly - Chan's parallel-variance update in double precision, deterministic.
// Round 6: (1) `Ctot` / `c0` - the planes of channels [c0, c0 + C) of a statistics tensor over Ctot channels (the strided 3x3
// and its projection share one launch and one statistics tensor, accflow_conv_desc.split_c0).  (2) TWO-PASS merge instead of
// the chained parallel-variance update: mean = sum(sums) / sum(counts); M2 = sum(M2_i + (sum_i - n_i mean)^2 / n_i) - the
// same quantity (both exact up to double rounding), but every record's term is independent of the others, where the
// chained form ran ~45 DEPENDENT double-precision divisions per lane (15 records + 6 butterfly steps, 2 divisions each:
// 14 us per launch, 15 launches per step, whatever the memory latency - batching the loads alone changed nothing).  Fixed
// summation order (thread t: records t, t + 256, ... in order; xor butterfly per wave; the four waves' sums in wave order), hence deterministic.  `second` (blockIdx.y = 1):
// a second statistics tensor finalised by the same launch (the closing pass of a projected residual block needs two).
struct finalize_args {
  const float* stats; int slots, Ctot, c0; float* meanrstd;
};
__global__ __launch_bounds__(256) void instance_stats_finalize_kernel(const finalize_args a0, const finalize_args a1, float eps,
                                                                      int C) {
  const bool sec = blockIdx.y != 0;          // (field-wise selects: a run-time choice between two by-value structs lands in scratch)
  const float* stats = sec ? a1.stats : a0.stats;
  float* meanrstd = sec ? a1.meanrstd : a0.meanrstd;
  const int slots = sec ? a1.slots : a0.slots, Ctot = sec ? a1.Ctot : a0.Ctot, c0 = sec ? a1.c0 : a0.c0;
  const long long plane = blockIdx.x;
  const long long b = plane / C, c = plane - b * C;
  const float* p = stats + ((b * Ctot + c0 + c) * slots) * 3;
  // 256 threads per plane, 8 records per thread in registers (2 048 slots: a 240 x 512 plane has 960 or 1 920); the first
  // version of this kernel ran one wave per plane and walked the records behind the first 1 024 with a dependent load per
  // record: 30 us per launch (up to 70) on the 64-channel half-resolution planes against 5 us on the others
  constexpr int MAXR = 8;
  __shared__ double red[3][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float rs[MAXR], rm[MAXR], rn[MAXR];
#pragma unroll
  for (int u = 0; u < MAXR; ++u) {
    const int i = tid + 256 * u;
    const bool ok = i < slots;
    rs[u] = ok ? p[3 * i] : 0.0f;
    rm[u] = ok ? p[3 * i + 1] : 0.0f;
    rn[u] = ok ? p[3 * i + 2] : 0.0f;
  }
  double s = 0.0, n = 0.0;
#pragma unroll
  for (int u = 0; u < MAXR; ++u) { s += (double)rs[u]; n += (double)rn[u]; }
  for (int i = tid + 256 * MAXR; i < slots; i += 256) { s += (double)p[3 * i]; n += (double)p[3 * i + 2]; }
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); n += __shfl_xor(n, off, 64); }
  if (lane == 0) { red[0][wave] = s; red[1][wave] = n; }
  __syncthreads();
  s = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];       // (fixed order: deterministic, the same in every thread)
  n = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
  const double mean = n > 0.0 ? s / n : 0.0;
  double m2 = 0.0;
#pragma unroll
  for (int u = 0; u < MAXR; ++u) {
    const double nb = rn[u];
    if (nb > 0.0) {
      const double dd = (double)rs[u] - nb * mean;
      m2 += (double)rm[u] + dd * dd / nb;
    }
  }
  for (int i = tid + 256 * MAXR; i < slots; i += 256) {
    const double nb = p[3 * i + 2];
    if (nb > 0.0) {
      const double dd = (double)p[3 * i] - nb * mean;
      m2 += (double)p[3 * i + 1] + dd * dd / nb;
    }
  }
  for (int off = 32; off > 0; off >>= 1) m2 += __shfl_xor(m2, off, 64);
  if (lane == 0) red[2][wave] = m2;
  __syncthreads();
  if (tid == 0) {
    m2 = ((red[2][0] + red[2][1]) + red[2][2]) + red[2][3];
    const double var = n > 0.0 ? m2 / n : 0.0;  // biased variance (nn.InstanceNorm2d)
    meanrstd[2 * plane] = (float)mean;
    meanrstd[2 * plane + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}
static void launch_finalize(const float* stats, int slots, int Ctot, int c0, float* mr, int B, int C, float eps, hipStream_t st,
                            const float* stats2 = nullptr, int slots2 = 0, int Ctot2 = 0, int c02 = 0, float* mr2 = nullptr) {
  finalize_args a0{stats, slots, Ctot, c0, mr}, a1{stats2, slots2, Ctot2, c02, mr2};
  hipLaunchKernelGGL(instance_stats_finalize_kernel, dim3((unsigned)((long long)B * C), stats2 ? 2 : 1), dim3(256), 0, st, a0, a1,
                     eps, C);
}

// out = f((x - mean) * rstd): the three modes of instance_norm_kernel in ONE pass over x (float4 when HW % 4 == 0)
__global__ __launch_bounds__(256) void instance_norm_apply_kernel(const float* __restrict__ x, const float* __restrict__ meanrstd,
                                                                  const float* __restrict__ res, float* __restrict__ out,
                                                                  int HW, long long total, int mode) {
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= total) return;
  if ((HW & 3) == 0) {
    const long long plane = i4 / HW;
    const float mean = meanrstd[2 * plane], rstd = meanrstd[2 * plane + 1];
    const float4 v = *reinterpret_cast<const float4*>(x + i4);
    float o[4] = {(v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd};
    if (mode >= 1)
      for (int k = 0; k < 4; ++k) o[k] = fmaxf(o[k], 0.0f);
    if (mode == 2) {
      const float4 r = *reinterpret_cast<const float4*>(res + i4);
      o[0] = fmaxf(r.x + o[0], 0.0f); o[1] = fmaxf(r.y + o[1], 0.0f); o[2] = fmaxf(r.z + o[2], 0.0f); o[3] = fmaxf(r.w + o[3], 0.0f);
    }
    *reinterpret_cast<float4*>(out + i4) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
    for (long long i = i4; i < i4 + 4 && i < total; ++i) {
      const long long plane = i / HW;
      float v = (x[i] - meanrstd[2 * plane]) * meanrstd[2 * plane + 1];
      if (mode >= 1) v = fmaxf(v, 0.0f);
      if (mode == 2) v = fmaxf(res[i] + v, 0.0f);
      out[i] = v;
    }
  }
}

__global__ void split_tanh_relu_kernel(const float* __restrict__ cnet, float* __restrict__ net, long long net_bs,
                                       float* __restrict__ inp, long long inp_bs, int B, int hd, int cd, int HW) {
  const long long per = (long long)(hd + cd) * HW;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * per) return;
  const int b = (int)(g / per);
  const long long r = g - b * per;
  const int c = (int)(r / HW), pix = (int)(r - (long long)c * HW);
  const float v = cnet[g];
  if (c < hd) net[b * net_bs + (long long)c * HW + pix] = tanhf(v);
  else inp[b * inp_bs + (long long)(c - hd) * HW + pix] = fmaxf(v, 0.0f);
}

// The same with a gather over the source items: output item b comes from cnet item idx.v[b] - the pairs of a sequence that
// share an image1 share its context features (AccFlow: 11 pairs over 6 context frames), so the frame-major encoder output
// feeds the pair-major workspace without a pair-major copy of it.
struct split_idx { int v[64]; };
__global__ void split_tanh_relu_idx_kernel(const float* __restrict__ cnet, split_idx idx, float* __restrict__ net,
                                           long long net_bs, float* __restrict__ inp, long long inp_bs, int B, int hd, int cd,
                                           int HW) {
  const long long per = (long long)(hd + cd) * HW;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * per) return;
  const int b = (int)(g / per);
  const long long r = g - b * per;
  const int c = (int)(r / HW), pix = (int)(r - (long long)c * HW);
  const float v = cnet[(long long)idx.v[b] * per + r];
  if (c < hd) net[b * net_bs + (long long)c * HW + pix] = tanhf(v);
  else inp[b * inp_bs + (long long)(c - hd) * HW + pix] = fmaxf(v, 0.0f);
}

__global__ void coords_grid_kernel(float* __restrict__ coords, const float* __restrict__ flow_init, int B, int H8,
                                   int W8) {
  const int P = H8 * W8;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (long long)B * 2 * P) return;
  const int pix = (int)(g % P), ch = (int)((g / P) & 1);
  const int y = pix / W8, x = pix - y * W8;
  float v = ch == 0 ? (float)x : (float)y;
  if (flow_init) v += flow_init[g];
  coords[g] = v;
}

// stack (optional, (B, 16, H8, W8)): channel c*7 + ky = flow[c] shifted by ky - 3 rows (zero outside), channels 14, 15
// zero - the 7x7 convolution of the 2-channel flow (convf1, update.py:85,92) then is a 1x7 convolution of 16 channels
// with the weights re-indexed [co][c*7 + ky][kx]: same products, K = 112 on the direct matrix-core kernel instead of a
// 98-deep im2col gather
__global__ void flow_from_coords_kernel(const float* __restrict__ coords1, float* __restrict__ dst0,
                                        long long dst0_bs, float* __restrict__ dst1, long long dst1_bs,
                                        float* __restrict__ stack, int B, int H8, int W8, int is_flow) {
  const int P = H8 * W8;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (long long)B * 2 * P) return;
  const int b = (int)(g / (2 * P));
  const int r = (int)(g - (long long)b * 2 * P);
  const int ch = r / P, pix = r - ch * P;
  const int y = pix / W8, x = pix - y * W8;
  const float v = is_flow ? coords1[g] : coords1[g] - (ch == 0 ? (float)x : (float)y);
  if (dst0) dst0[b * dst0_bs + r] = v;
  if (dst1) dst1[b * dst1_bs + r] = v;
  if (stack) {
    float* s = stack + (long long)b * 16 * P + pix;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const int ys = y + ky - 3;
      const bool ok = (unsigned)ys < (unsigned)H8;
      const float c = ok ? coords1[g + (long long)(ky - 3) * W8] : 0.0f;
      const float u = (ok && !is_flow) ? c - (ch == 0 ? (float)x : (float)ys) : c;
      s[(long long)(ch * 7 + ky) * P] = u;
    }
    s[(long long)(14 + ch) * P] = 0.0f;
  }
}

// S16 form (accflow_conv_desc "S16" format): thread = (b, pixel), both flow channels - the 16-channel row-shifted stack
// as two octets x {hi, lo} 16-byte chunks, and the flow itself as one dword per term inside the motion features' last octet
__global__ __launch_bounds__(256) void flow_from_coords_s16_kernel(const float* __restrict__ coords1, float* __restrict__ dst0,
                                                                   long long dst0_bs, float* __restrict__ dst1, long long dst1_bs,
                                                                   mu32x4* __restrict__ stack, long long stack_bs,
                                                                   unsigned* __restrict__ motion, long long motion_bs, int motion_ch,
                                                                   int* guard, int B, int H8, int W8, int is_flow) {
  const int P = H8 * W8;
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= (long long)B * P) return;
  const int b = (int)(g / P), pix = (int)(g - (long long)b * P);
  const int y = pix / W8, x = pix - y * W8;
  const float* c = coords1 + (long long)b * 2 * P + pix;
  float st[16];
#pragma unroll
  for (int ch = 0; ch < 2; ++ch)
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const int ys = y + ky - 3;
      const bool ok = (unsigned)ys < (unsigned)H8;
      const float cv = ok ? c[(long long)ch * P + (long long)(ky - 3) * W8] : 0.0f;
      st[ch * 7 + ky] = (ok && !is_flow) ? cv - (ch == 0 ? (float)x : (float)ys) : cv;
    }
  st[14] = 0.0f; st[15] = 0.0f;
  const float fx = st[3], fy = st[10];
  if (dst0) { dst0[b * dst0_bs + pix] = fx; dst0[b * dst0_bs + P + pix] = fy; }
  if (dst1) { dst1[b * dst1_bs + pix] = fx; dst1[b * dst1_bs + P + pix] = fy; }
  bool bad = false;
  if (stack) {
    mu32x4* sb = stack + (b * stack_bs) / 4;     // (strides count 4-byte words; a chunk is 4 of them)
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = st[o * 8 + j];
      mu32x4 hi, lo;
      s16_split8(v, hi, lo, bad);
      sb[((long long)(o * 2 + 0)) * P + pix] = hi;
      sb[((long long)(o * 2 + 1)) * P + pix] = lo;
    }
  }
  if (motion) {
    float v[8] = {fx, fy, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    mu32x4 hi, lo;
    s16_split8(v, hi, lo, bad);
    const int oct = motion_ch >> 3, word = (motion_ch & 7) >> 1;   // the pair's dword inside the 16-byte chunk
    unsigned* mb = motion + b * motion_bs;
    mb[(((long long)(oct * 2 + 0)) * P + pix) * 4 + word] = hi[0];
    mb[(((long long)(oct * 2 + 1)) * P + pix) * 4 + word] = lo[0];
  }
  if (bad && guard) atomicOr(guard, 1);
}

// fp32 (B, C, HW planes) -> S16: thread = (b, octet, pixel); 8 strided reads (coalesced along the pixels), two 16-byte
// chunk writes.  Used where a non-convolution kernel produced a tensor that convolutions consume (tanh(cnet) -> h,
// GMA's aggregated motion features, module-boundary entries).
__global__ __launch_bounds__(256) void to_s16_kernel(const float* __restrict__ src, long long src_bs, mu32x4* __restrict__ dst,
                                                     long long dst_bs, int* guard, int B, int C, int HW) {
  const int O = (C + 7) >> 3;
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= (long long)B * O * HW) return;
  const int pix = (int)(g % HW), o = (int)((g / HW) % O), b = (int)(g / ((long long)HW * O));
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (o * 8 + j < C) ? src[b * src_bs + (long long)(o * 8 + j) * HW + pix] : 0.0f;
  mu32x4 hi, lo;
  bool bad = false;
  s16_split8(v, hi, lo, bad);
  mu32x4* d = dst + (b * dst_bs) / 4;
  d[(long long)(o * 2 + 0) * HW + pix] = hi;
  d[(long long)(o * 2 + 1) * HW + pix] = lo;
  if (bad && guard) atomicOr(guard, 1);
}

// instance_norm_apply_kernel with the result ALSO (or only: out may be NULL) written pre-split: thread = (b, octet, pixel),
// the to_s16_kernel access pattern.  The S16 copy feeds the next residual block's convolutions by LDS DMA
// (extractor.py:56-63: the block output is read by the next conv1 and, as fp32, by the next residual add).
__global__ __launch_bounds__(256) void instance_norm_apply_s16_kernel(const float* __restrict__ x, const float* __restrict__ meanrstd,
                                                                      const float* __restrict__ res, float* __restrict__ out,
                                                                      mu32x4* __restrict__ dst, long long dst_bs, int* guard,
                                                                      int B, int C, int HW, int mode) {
  const int O = (C + 7) >> 3;
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= (long long)B * O * HW) return;
  const int pix = (int)(g % HW), o = (int)((g / HW) % O), b = (int)(g / ((long long)HW * O));
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = o * 8 + j;
    v[j] = 0.0f;
    if (c < C) {
      const long long plane = (long long)b * C + c, e = plane * HW + pix;
      float t = (x[e] - meanrstd[2 * plane]) * meanrstd[2 * plane + 1];
      if (mode >= 1) t = fmaxf(t, 0.0f);
      if (mode == 2) t = fmaxf(res[e] + t, 0.0f);
      if (out) out[e] = t;
      v[j] = t;
    }
  }
  mu32x4 hi, lo;
  bool bad = false;
  s16_split8(v, hi, lo, bad);
  mu32x4* d = dst + (b * dst_bs) / 4;
  d[(long long)(o * 2 + 0) * HW + pix] = hi;
  d[(long long)(o * 2 + 1) * HW + pix] = lo;
  if (bad && guard) atomicOr(guard, 1);
}

// mode 2 with the residual read from an S16 tensor: relu((hi + lo) / 2^4 + relu(norm(x))) -> S16 (and fp32 if out != NULL)
__global__ __launch_bounds__(256) void instance_norm_apply_s16res_kernel(const float* __restrict__ x, const float* __restrict__ meanrstd,
                                                                         const mu32x4* __restrict__ res, long long res_bs,
                                                                         float* __restrict__ out, mu32x4* __restrict__ dst,
                                                                         long long dst_bs, int* guard, int B, int C, int HW) {
  const int O = (C + 7) >> 3;
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= (long long)B * O * HW) return;
  const int pix = (int)(g % HW), o = (int)((g / HW) % O), b = (int)(g / ((long long)HW * O));
  const mu32x4* rp = res + (b * res_bs) / 4;
  const mu32x4 rh = rp[(long long)(o * 2 + 0) * HW + pix], rl = rp[(long long)(o * 2 + 1) * HW + pix];
  const unsigned hw[4] = {rh.x, rh.y, rh.z, rh.w}, lw[4] = {rl.x, rl.y, rl.z, rl.w};
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = o * 8 + j;
    v[j] = 0.0f;
    if (c < C) {
      const long long plane = (long long)b * C + c, e = plane * HW + pix;
      const _Float16 hh = __builtin_bit_cast(_Float16, (unsigned short)((hw[j >> 1] >> (16 * (j & 1))) & 0xFFFFu));
      const _Float16 ll = __builtin_bit_cast(_Float16, (unsigned short)((lw[j >> 1] >> (16 * (j & 1))) & 0xFFFFu));
      const float r = ((float)hh + (float)ll) * (1.0f / (float)(1 << ACCFLOW_F16_ASHIFT));
      float t = fmaxf((x[e] - meanrstd[2 * plane]) * meanrstd[2 * plane + 1], 0.0f);
      t = fmaxf(r + t, 0.0f);
      if (out) out[e] = t;
      v[j] = t;
    }
  }
  mu32x4 hi, lo;
  bool bad = false;
  s16_split8(v, hi, lo, bad);
  mu32x4* d = dst + (b * dst_bs) / 4;
  d[(long long)(o * 2 + 0) * HW + pix] = hi;
  d[(long long)(o * 2 + 1) * HW + pix] = lo;
  if (bad && guard) atomicOr(guard, 1);
}

// The closing pass of a residual block WITH a projection in the InstanceNorm encoder (extractor.py:51-53,59-63):
// relu(norm3(proj) + relu(norm2(x))) -> S16, where `proj` is the raw output of the 1x1 stride-2 projection - a channel slice
// (batch stride res_bs) of the tensor the block's strided 3x3 launch wrote (accflow_conv_desc.split_c0) - normalised HERE
// with its own {mean, rstd} (mr3): the separate in-place norm pass over the projection is gone.
__global__ __launch_bounds__(256) void instance_norm_apply_s16proj_kernel(const float* __restrict__ x, const float* __restrict__ mr2,
                                                                          const float* __restrict__ res, long long res_bs,
                                                                          const float* __restrict__ mr3, mu32x4* __restrict__ dst,
                                                                          long long dst_bs, int* guard, int B, int C, int HW) {
  const int O = (C + 7) >> 3;
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= (long long)B * O * HW) return;
  const int pix = (int)(g % HW), o = (int)((g / HW) % O), b = (int)(g / ((long long)HW * O));
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = o * 8 + j;
    v[j] = 0.0f;
    if (c < C) {
      const long long plane = (long long)b * C + c;
      const float t = fmaxf((x[plane * HW + pix] - mr2[2 * plane]) * mr2[2 * plane + 1], 0.0f);
      const float r = (res[b * res_bs + (long long)c * HW + pix] - mr3[2 * plane]) * mr3[2 * plane + 1];
      v[j] = fmaxf(r + t, 0.0f);
    }
  }
  mu32x4 hi, lo;
  bool bad = false;
  s16_split8(v, hi, lo, bad);
  mu32x4* d = dst + (b * dst_bs) / 4;
  d[(long long)(o * 2 + 0) * HW + pix] = hi;
  d[(long long)(o * 2 + 1) * HW + pix] = lo;
  if (bad && guard) atomicOr(guard, 1);
}

__global__ void blend_kernel(const float* __restrict__ f1, const float* __restrict__ f2, const float* __restrict__ m,
                             float* __restrict__ out, int B, int C, int HW) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (long long)B * C * HW) return;
  const int b = (int)(g / ((long long)C * HW));
  const int pix = (int)(g % HW);
  const float mm = m[(long long)b * HW + pix];
  out[g] = f1[g] * mm + (1.0f - mm) * f2[g];
}

// blend with the result PRE-SPLIT only (round 6: the fused features of a fusion step feed the flow decoder's convolutions and
// nothing else - AccFlow_.py:199-200 - so the fp32 tensor and the to_s16 pass behind it are gone): thread = (b, octet, pixel)
__global__ __launch_bounds__(256) void blend_s16_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                        const float* __restrict__ m, mu32x4* __restrict__ dst, long long dst_bs,
                                                        int* guard, int B, int C, int HW) {
  const int O = (C + 7) >> 3;
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= (long long)B * O * HW) return;
  const int pix = (int)(g % HW), o = (int)((g / HW) % O), b = (int)(g / ((long long)HW * O));
  const float mm = m[(long long)b * HW + pix];
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = o * 8 + j;
    const long long e = ((long long)b * C + c) * HW + pix;
    v[j] = c < C ? f1[e] * mm + (1.0f - mm) * f2[e] : 0.0f;       // (blend_kernel's expression)
  }
  mu32x4 hi, lo;
  bool bad = false;
  s16_split8(v, hi, lo, bad);
  mu32x4* d = dst + (b * dst_bs) / 4;
  d[(long long)(o * 2 + 0) * HW + pix] = hi;
  d[(long long)(o * 2 + 1) * HW + pix] = lo;
  if (bad && guard) atomicOr(guard, 1);
}

__global__ void copy_kernel(const float* __restrict__ src, long long src_bs, float* __restrict__ dst, long long dst_bs,
                            int B, long long per) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * per) return;
  const int b = (int)(g / per);
  const long long r = g - b * per;
  dst[b * dst_bs + r] = src[b * src_bs + r];
}

__global__ void act_kernel(float* __restrict__ x, long long x_bs, int B, long long per, int act) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * per) return;
  const int b = (int)(g / per);
  const long long r = g - b * per;
  x[b * x_bs + r] = apply_act(x[b * x_bs + r], act);
}

}  // namespace

extern "C" int accflow_activation_f32(float* x, long long x_bs, int B, int C, int HW, int act, void* stream) {
  if (!x || B <= 0 || C <= 0 || HW <= 0) return 1;
  const long long per = (long long)C * HW;
  hipLaunchKernelGGL(act_kernel, dim3(cdiv(B * per, 256)), dim3(256), 0, as_stream(stream), x, x_bs, B, per, act);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_instance_norm_f32(const float* x, const float* res, float* out, int B, int C, int HW, float eps,
                                         int mode, void* stream) {
  if (!x || !out || B <= 0 || C <= 0 || HW <= 0 || mode < 0 || mode > 2 || (mode == 2 && !res)) return 1;
  hipLaunchKernelGGL(instance_norm_kernel, dim3((unsigned)((long long)B * C)), dim3(512), 0, as_stream(stream), x, res,
                     out, HW, eps, mode);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_instance_stats_finalize_f32(const float* stats, int slots, float* meanrstd, int B, int C, float eps,
                                                   void* stream) {
  if (!stats || !meanrstd || slots <= 0 || B <= 0 || C <= 0) return 1;
  launch_finalize(stats, slots, C, 0, meanrstd, B, C, eps, as_stream(stream));
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_instance_stats_finalize_sub_f32(const float* stats, int slots, int Ctot, int c0, float* meanrstd, int B,
                                                       int C, float eps, void* stream) {
  if (!stats || !meanrstd || slots <= 0 || B <= 0 || C <= 0 || c0 < 0 || c0 + C > Ctot) return 1;
  launch_finalize(stats, slots, Ctot, c0, meanrstd, B, C, eps, as_stream(stream));
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_instance_norm_apply_s16proj_f32(const float* x, const float* stats, int slots, const float* res,
                                                       long long res_bs, const float* res_stats, int res_slots, int res_ctot,
                                                       int res_c0, float* meanrstd, void* out16, long long out16_bs, int* guard,
                                                       int B, int C, int HW, float eps, void* stream) {
  if (!x || !stats || !res || !res_stats || !meanrstd || !out16 || slots <= 0 || res_slots <= 0 || B <= 0 || C <= 0 || HW <= 0 ||
      res_c0 < 0 || res_c0 + C > res_ctot)
    return 1;
  hipStream_t st = as_stream(stream);
  float* mr3 = meanrstd + 2LL * B * C;     // (meanrstd: 4 * B * C floats)
  launch_finalize(stats, slots, C, 0, meanrstd, B, C, eps, st, res_stats, res_slots, res_ctot, res_c0, mr3);   // (one launch)
  const long long n = (long long)B * ((C + 7) / 8) * HW;
  hipLaunchKernelGGL(instance_norm_apply_s16proj_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, x, meanrstd, res, res_bs, mr3,
                     reinterpret_cast<mu32x4*>(out16), out16_bs, guard, B, C, HW);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_instance_norm_apply_f32(const float* x, const float* stats, int slots, float* meanrstd,
                                               const float* res, float* out, int B, int C, int HW, float eps, int mode,
                                               void* stream) {
  if (!x || !stats || !meanrstd || !out || slots <= 0 || B <= 0 || C <= 0 || HW <= 0 || mode < 0 || mode > 2 ||
      (mode == 2 && !res))
    return 1;
  const long long total = (long long)B * C * HW;
  launch_finalize(stats, slots, C, 0, meanrstd, B, C, eps, as_stream(stream));
  hipLaunchKernelGGL(instance_norm_apply_kernel, dim3(cdiv(cdiv(total, 4), 256)), dim3(256), 0, as_stream(stream), x,
                     meanrstd, res, out, HW, total, mode);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_instance_norm_apply_s16_f32(const float* x, const float* stats, int slots, float* meanrstd,
                                                   const float* res, float* out, void* out16, long long out16_bs, int* guard,
                                                   int B, int C, int HW, float eps, int mode, void* stream) {
  if (!x || !stats || !meanrstd || !out16 || slots <= 0 || B <= 0 || C <= 0 || HW <= 0 || mode < 0 || mode > 2 ||
      (mode == 2 && !res))
    return 1;
  launch_finalize(stats, slots, C, 0, meanrstd, B, C, eps, as_stream(stream));
  const long long n = (long long)B * ((C + 7) / 8) * HW;
  hipLaunchKernelGGL(instance_norm_apply_s16_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), x, meanrstd, res,
                     out, reinterpret_cast<mu32x4*>(out16), out16_bs, guard, B, C, HW, mode);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_instance_norm_apply_s16res_f32(const float* x, const float* stats, int slots, float* meanrstd,
                                                      const void* res16, long long res16_bs, float* out, void* out16,
                                                      long long out16_bs, int* guard, int B, int C, int HW, float eps,
                                                      void* stream) {
  if (!x || !stats || !meanrstd || !res16 || !out16 || slots <= 0 || B <= 0 || C <= 0 || HW <= 0) return 1;
  launch_finalize(stats, slots, C, 0, meanrstd, B, C, eps, as_stream(stream));
  const long long n = (long long)B * ((C + 7) / 8) * HW;
  hipLaunchKernelGGL(instance_norm_apply_s16res_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), x, meanrstd,
                     reinterpret_cast<const mu32x4*>(res16), res16_bs, out, reinterpret_cast<mu32x4*>(out16), out16_bs, guard,
                     B, C, HW);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_split_tanh_relu_f32(const float* cnet, float* net, long long net_bs, float* inp,
                                           long long inp_bs, int B, int hd, int cd, int HW, void* stream) {
  if (!cnet || !net || !inp || B <= 0 || hd <= 0 || cd <= 0 || HW <= 0) return 1;
  const long long n = (long long)B * (hd + cd) * HW;
  hipLaunchKernelGGL(split_tanh_relu_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), cnet, net, net_bs,
                     inp, inp_bs, B, hd, cd, HW);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_split_tanh_relu_idx_f32(const float* cnet, int n_items, const int* idx, float* net, long long net_bs,
                                               float* inp, long long inp_bs, int B, int hd, int cd, int HW, void* stream) {
  if (!cnet || !idx || !net || !inp || n_items <= 0 || B <= 0 || hd <= 0 || cd <= 0 || HW <= 0) return 1;
  for (int b = 0; b < B; ++b)
    if (idx[b] < 0 || idx[b] >= n_items) return 1;   // (host array, validated before any launch)
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int nb = B - b0 < 64 ? B - b0 : 64;
    split_idx a;
    for (int k = 0; k < 64; ++k) a.v[k] = k < nb ? idx[b0 + k] : 0;
    const long long n = (long long)nb * (hd + cd) * HW;
    hipLaunchKernelGGL(split_tanh_relu_idx_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), cnet, a,
                       net + b0 * net_bs, net_bs, inp + b0 * inp_bs, inp_bs, nb, hd, cd, HW);
  }
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_coords_grid_f32(float* coords, const float* flow_init, int B, int H8, int W8, void* stream) {
  if (!coords || B <= 0 || H8 <= 0 || W8 <= 0) return 1;
  const long long n = (long long)B * 2 * H8 * W8;
  hipLaunchKernelGGL(coords_grid_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), coords, flow_init, B, H8,
                     W8);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_flow_from_coords_f32(const float* coords1, float* dst0, long long dst0_bs, float* dst1,
                                            long long dst1_bs, float* stack16, int is_flow, int B, int H8, int W8,
                                            void* stream) {
  if (!coords1 || (!dst0 && !dst1 && !stack16) || B <= 0 || H8 <= 0 || W8 <= 0) return 1;
  const long long n = (long long)B * 2 * H8 * W8;
  hipLaunchKernelGGL(flow_from_coords_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), coords1, dst0,
                     dst0_bs, dst1, dst1_bs, stack16, B, H8, W8, is_flow);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_flow_from_coords_s16(const float* coords1, float* dst0, long long dst0_bs, float* dst1,
                                            long long dst1_bs, void* stack16, long long stack16_bs, void* motion16,
                                            long long motion16_bs, int motion_ch, int* guard, int is_flow, int B, int H8,
                                            int W8, void* stream) {
  if (!coords1 || (!dst0 && !dst1 && !stack16 && !motion16) || B <= 0 || H8 <= 0 || W8 <= 0 || motion_ch < 0 || (motion_ch & 1))
    return 1;
  const long long n = (long long)B * H8 * W8;
  hipLaunchKernelGGL(flow_from_coords_s16_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), coords1, dst0, dst0_bs,
                     dst1, dst1_bs, reinterpret_cast<mu32x4*>(stack16), stack16_bs, reinterpret_cast<unsigned*>(motion16),
                     motion16_bs, motion_ch, guard, B, H8, W8, is_flow);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_to_s16_f32(const float* src, long long src_bs, void* dst16, long long dst16_bs, int* guard, int B,
                                  int C, int HW, void* stream) {
  if (!src || !dst16 || B <= 0 || C <= 0 || HW <= 0) return 1;
  const long long n = (long long)B * ((C + 7) / 8) * HW;
  hipLaunchKernelGGL(to_s16_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), src, src_bs,
                     reinterpret_cast<mu32x4*>(dst16), dst16_bs, guard, B, C, HW);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_blend_f32(const float* f1, const float* f2, const float* m, float* out, int B, int C, int HW,
                                 void* stream) {
  if (!f1 || !f2 || !m || !out || B <= 0 || C <= 0 || HW <= 0) return 1;
  const long long n = (long long)B * C * HW;
  hipLaunchKernelGGL(blend_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), f1, f2, m, out, B, C, HW);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_blend_s16(const float* f1, const float* f2, const float* m, void* out16, long long out16_bs, int* guard,
                                 int B, int C, int HW, void* stream) {
  if (!f1 || !f2 || !m || !out16 || B <= 0 || C <= 0 || HW <= 0) return 1;
  const long long n = (long long)B * ((C + 7) / 8) * HW;
  hipLaunchKernelGGL(blend_s16_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), f1, f2, m,
                     reinterpret_cast<mu32x4*>(out16), out16_bs, guard, B, C, HW);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_copy_f32(const float* src, long long src_bs, float* dst, long long dst_bs, int B, int C, int HW,
                                void* stream) {
  if (!src || !dst || B <= 0 || C <= 0 || HW <= 0) return 1;
  const long long per = (long long)C * HW;
  hipLaunchKernelGGL(copy_kernel, dim3(cdiv(B * per, 256)), dim3(256), 0, as_stream(stream), src, src_bs, dst, dst_bs,
                     B, per);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

// Second half of a small-Cout convolution computed as a 1x1 matrix-core conv followed by a shifted sum:
// z[b][tap*Cout + co][p] = sum_c w[co][c][tap] * x[b][c][p] (all taps at once, one pass over x on the MFMA), then
// out[b][co][y][x] = epi(act(bias[co] + sum_tap z[b][tap*Cout+co][y + ky - padH][x + kx - padW])) with zero padding.
__global__ __launch_bounds__(256) void tap_sum_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                      const float* __restrict__ e0, long long e0_bs, float* __restrict__ out,
                                                      long long out_bs, int B, int Cout, int H, int W, int KH, int KW,
                                                      int padH, int padW, int act, int epi, int nparts, long long part_stride,
                                                      long long z_bs) {
  const long long n = (long long)B * Cout * H * W;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % W), y = (int)((i / W) % H);
  const int co = (int)((i / ((long long)W * H)) % Cout), b = (int)(i / ((long long)W * H * Cout));
  const int HW = H * W;
  const float* zb = z + (long long)b * z_bs;
  float v = 0.0f;
  for (int ky = 0; ky < KH; ++ky) {
    const int yy = y + ky - padH;
    if ((unsigned)yy >= (unsigned)H) continue;
    for (int kx = 0; kx < KW; ++kx) {
      const int xx = x + kx - padW;
      if ((unsigned)xx >= (unsigned)W) continue;
      const float* zp = zb + (long long)((ky * KW + kx) * Cout + co) * HW + yy * W + xx;
      float t = zp[0];
      for (int p = 1; p < nparts; ++p) t += zp[p * part_stride];   // (ACCFLOW_EPI_TAPGEMM: one part per 128-channel block)
      v += t;
    }
  }
  if (bias) v += bias[co];
  v = apply_act(v, act);
  const long long o = (long long)co * HW + y * W + x;
  if (epi == ACCFLOW_EPI_ACCUM) v += e0[b * e0_bs + o];
  else if (epi == ACCFLOW_EPI_RES_RELU) v = fmaxf(e0[b * e0_bs + o] + v, 0.0f);
  out[b * out_bs + o] = v;
}

extern "C" int accflow_tap_sum_f32(const float* z, const float* bias, const float* e0, long long e0_bs, float* out,
                                   long long out_bs, int B, int Cout, int H, int W, int KH, int KW, int padH, int padW,
                                   int act, int epi, void* stream) {
  if (!z || !out || B <= 0 || Cout <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0) return 1;
  if (epi != ACCFLOW_EPI_STORE && epi != ACCFLOW_EPI_ACCUM && epi != ACCFLOW_EPI_RES_RELU) return 1;
  if (epi != ACCFLOW_EPI_STORE && !e0) return 1;
  const long long n = (long long)B * Cout * H * W;
  hipLaunchKernelGGL(tap_sum_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), z, bias, e0, e0_bs, out, out_bs, B,
                     Cout, H, W, KH, KW, padH, padW, act, epi, 1, 0LL, (long long)KH * KW * Cout * H * W);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_tap_sum_parts_f32(const float* z, int nparts, long long part_stride, long long z_bs, const float* bias,
                                         const float* e0, long long e0_bs, float* out, long long out_bs, int B, int Cout, int H,
                                         int W, int KH, int KW, int padH, int padW, int act, int epi, void* stream) {
  if (!z || !out || B <= 0 || Cout <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || nparts < 1 || nparts > 8) return 1;
  if (z_bs < (long long)KH * KW * Cout * H * W || (nparts > 1 && part_stride <= 0)) return 1;
  if (epi != ACCFLOW_EPI_STORE && epi != ACCFLOW_EPI_ACCUM && epi != ACCFLOW_EPI_RES_RELU) return 1;
  if (epi != ACCFLOW_EPI_STORE && !e0) return 1;
  const long long n = (long long)B * Cout * H * W;
  hipLaunchKernelGGL(tap_sum_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), z, bias, e0, e0_bs, out, out_bs, B,
                     Cout, H, W, KH, KW, padH, padW, act, epi, nparts, part_stride, z_bs);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_abi_version(void) { return ACCFLOW_ABI_VERSION; }
extern "C" int accflow_conv_desc_bytes(void) { return (int)sizeof(accflow_conv_desc); }
extern "C" int accflow_conv_src_bytes(void) { return (int)sizeof(accflow_conv_src); }
