// Gather-type ops of the path: convex upsampling, backward warp, occlusion / error maps, downflow8.
#include "common.h"

namespace {

// RAFT.upsample_flow (raft/raft.py:81-92): out[n,c,8h+a,8w+b] = sum_k softmax_k(mask[n,k*64+a*8+b,h,w])
// * 8*flow_zp[n,c,h+k/3-1,w+k%3-1].  lane = coarse pixel: every mask load is 64 consecutive floats of
// one mask channel; every lane owns 8 consecutive output columns -> two 16-B stores per (c, row).
__global__ __launch_bounds__(256) void convex_upsample_kernel(const float* __restrict__ flow, long long flow_bs,
                                                              const float* __restrict__ mask, long long mask_bs,
                                                              float* __restrict__ out, int B, int H8, int W8) {
  const int P = H8 * W8;
  const long long gp = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gp >= (long long)B * P) return;
  const int b = (int)(gp / P), pix = (int)(gp - (long long)b * P);
  const int h = pix / W8, w = pix - h * W8;
  const float* fl = flow + b * flow_bs;
  float f0[9], f1[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = h + k / 3 - 1, xx = w + k % 3 - 1;
    const bool ok = (unsigned)yy < (unsigned)H8 && (unsigned)xx < (unsigned)W8;
    f0[k] = ok ? 8.0f * fl[yy * W8 + xx] : 0.0f;
    f1[k] = ok ? 8.0f * fl[P + yy * W8 + xx] : 0.0f;
  }
  const float* mk = mask + b * mask_bs + pix;
  const int W = 8 * W8;
  float* o0 = out + ((long long)(b * 2 + 0) * (8 * H8) + 8 * h) * W + 8 * w;
  float* o1 = out + ((long long)(b * 2 + 1) * (8 * H8) + 8 * h) * W + 8 * w;
#pragma unroll 1
  for (int a = 0; a < 8; ++a) {
    float r0[8], r1[8];
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) {
      float m[9];
      float mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        m[k] = mk[(long long)(k * 64 + a * 8 + bb) * P];
        mx = fmaxf(mx, m[k]);
      }
      float s = 0.0f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        m[k] = expf(m[k] - mx);
        s += m[k];
      }
      float u0 = 0.0f, u1 = 0.0f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float wk = m[k] / s;
        u0 += wk * f0[k];
        u1 += wk * f1[k];
      }
      r0[bb] = u0;
      r1[bb] = u1;
    }
    float4* q0 = reinterpret_cast<float4*>(o0 + (long long)a * W);
    float4* q1 = reinterpret_cast<float4*>(o1 + (long long)a * W);
    q0[0] = make_float4(r0[0], r0[1], r0[2], r0[3]);
    q0[1] = make_float4(r0[4], r0[5], r0[6], r0[7]);
    q1[0] = make_float4(r1[0], r1[1], r1[2], r1[3]);
    q1[1] = make_float4(r1[4], r1[5], r1[6], r1[7]);
  }
}

struct WarpTaps {
  int o00, o01, o10, o11;
  float w00, w01, w10, w11;
};

// taps of F.grid_sample(bilinear, zeros, align_corners=True) at pixel coords (sx, sy); weights of
// out-of-plane corners are zeroed and their offsets clamped to 0 so loads stay in bounds.
__device__ __forceinline__ WarpTaps make_taps(float sx, float sy, int H, int W) {
  sx = fminf(fmaxf(sx, -1.0e6f), 1.0e6f);
  sy = fminf(fmaxf(sy, -1.0e6f), 1.0e6f);
  const float fx0 = floorf(sx), fy0 = floorf(sy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float ax = sx - fx0, ay = sy - fy0;
  const bool xin0 = (unsigned)x0 < (unsigned)W, xin1 = (unsigned)(x0 + 1) < (unsigned)W;
  const bool yin0 = (unsigned)y0 < (unsigned)H, yin1 = (unsigned)(y0 + 1) < (unsigned)H;
  WarpTaps t;
  t.o00 = (xin0 && yin0) ? y0 * W + x0 : 0;
  t.o01 = (xin1 && yin0) ? y0 * W + x0 + 1 : 0;
  t.o10 = (xin0 && yin1) ? (y0 + 1) * W + x0 : 0;
  t.o11 = (xin1 && yin1) ? (y0 + 1) * W + x0 + 1 : 0;
  t.w00 = (xin0 && yin0) ? (1.0f - ax) * (1.0f - ay) : 0.0f;
  t.w01 = (xin1 && yin0) ? ax * (1.0f - ay) : 0.0f;
  t.w10 = (xin0 && yin1) ? (1.0f - ax) * ay : 0.0f;
  t.w11 = (xin1 && yin1) ? ax * ay : 0.0f;
  return t;
}
__device__ __forceinline__ float tap_sample(const float* __restrict__ plane, const WarpTaps& t) {
  return plane[t.o00] * t.w00 + plane[t.o01] * t.w01 + plane[t.o10] * t.w10 + plane[t.o11] * t.w11;
}

constexpr int WARP_CCHUNK = 16;

// backwarp (networks/utils.py:96-124): thread = pixel, blockIdx.y = chunk of channels.
// ADD: out = flow + warp(img, flow) (C = 2: the composition of two flow fields, accflow_compose_flow_f32).
template <bool ADD>
__global__ __launch_bounds__(256) void backwarp_kernel(const float* __restrict__ img, long long img_bs,
                                                       const float* __restrict__ flow, long long flow_bs,
                                                       float* __restrict__ out, long long out_bs, int B, int C, int H,
                                                       int W) {
  const int HW = H * W;
  const long long gp = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gp >= (long long)B * HW) return;
  const int b = (int)(gp / HW), pix = (int)(gp - (long long)b * HW);
  const int y = pix / W, x = pix - y * W;
  const float u = flow[b * flow_bs + pix], v0 = flow[b * flow_bs + HW + pix];
  const WarpTaps t = make_taps((float)x + u, (float)y + v0, H, W);
  const int c0 = blockIdx.y * WARP_CCHUNK, c1 = min(C, c0 + WARP_CCHUNK);
  for (int c = c0; c < c1; ++c) {
    float v = tap_sample(img + b * img_bs + (long long)c * HW, t);
    if constexpr (ADD) v += c == 0 ? u : v0;
    out[b * out_bs + (long long)c * HW + pix] = v;
  }
}

// getOcc (AccFlow_.py:127-135)
template <bool BINARY>
__global__ __launch_bounds__(256) void get_occ_kernel(const float* __restrict__ flow, long long flow_bs,
                                                      const float* __restrict__ i1, long long i1_bs,
                                                      const float* __restrict__ i2, long long i2_bs,
                                                      float* __restrict__ out, long long out_bs, int B, int C, int H,
                                                      int W) {
  const int HW = H * W;
  const long long gp = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gp >= (long long)B * HW) return;
  const int b = (int)(gp / HW), pix = (int)(gp - (long long)b * HW);
  const int y = pix / W, x = pix - y * W;
  const float u = flow[b * flow_bs + pix], v = flow[b * flow_bs + HW + pix];
  const WarpTaps t = make_taps((float)x + u, (float)y + v, H, W);
  if constexpr (BINARY) {
    // The sum over the channels is thresholded, so its association must not depend on which kernel a batch size
    // selects: four quarter sums added as (q0 + q1) + (q2 + q3), exactly what get_occ_binary4_kernel computes with one
    // quarter per wave (a sequence evaluated alone and inside a large batch gets the same occlusion bits).
    const int cper = (C + 3) / 4;
    float q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float s = 0.0f;
      for (int c = k * cper; c < min(C, (k + 1) * cper); ++c) {
        const float wv = tap_sample(i2 + b * i2_bs + (long long)c * HW, t);
        s += fabsf(i1[b * i1_bs + (long long)c * HW + pix] - wv);
      }
      q[k] = s;
    }
    const float tot = (q[0] + q[1]) + (q[2] + q[3]);
    out[b * out_bs + pix] = (tot / (float)C <= 1.0f) ? 1.0f : 0.0f;
  } else {
    const int c0 = blockIdx.y * WARP_CCHUNK, c1 = min(C, c0 + WARP_CCHUNK);
    for (int c = c0; c < c1; ++c) {
      const float wv = tap_sample(i2 + b * i2_bs + (long long)c * HW, t);
      out[b * out_bs + (long long)c * HW + pix] = fabsf(i1[b * i1_bs + (long long)c * HW + pix] - wv);
    }
  }
}

// getOcc with the map PRE-SPLIT only (round 6; the fusion chain's consumers of both maps are convolutions: AccPlus's
// cat[df, f, o] / cat[f_, df, o] members and Blending's 1x1, AccFlow_.py:98,105,119): thread = (pixel, octet of channels);
// BINARY: one octet whose channel 0 holds the bit (the same thresholded sum, association and all, as get_occ_kernel<true>).
template <bool BINARY>
__global__ __launch_bounds__(256) void get_occ_s16_kernel(const float* __restrict__ flow, long long flow_bs,
                                                          const float* __restrict__ i1, long long i1_bs,
                                                          const float* __restrict__ i2, long long i2_bs,
                                                          mu32x4* __restrict__ dst, long long dst_bs, int* guard, int B, int C,
                                                          int H, int W) {
  const int HW = H * W;
  const long long gp = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gp >= (long long)B * HW) return;
  const int b = (int)(gp / HW), pix = (int)(gp - (long long)b * HW);
  const int y = pix / W, x = pix - y * W;
  const float u = flow[b * flow_bs + pix], v = flow[b * flow_bs + HW + pix];
  const WarpTaps t = make_taps((float)x + u, (float)y + v, H, W);
  float val[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  const int o = BINARY ? 0 : blockIdx.y;
  if constexpr (BINARY) {
    const int cper = (C + 3) / 4;
    float q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float s = 0.0f;
      for (int c = k * cper; c < min(C, (k + 1) * cper); ++c) {
        const float wv = tap_sample(i2 + b * i2_bs + (long long)c * HW, t);
        s += fabsf(i1[b * i1_bs + (long long)c * HW + pix] - wv);
      }
      q[k] = s;
    }
    const float tot = (q[0] + q[1]) + (q[2] + q[3]);
    val[0] = (tot / (float)C <= 1.0f) ? 1.0f : 0.0f;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = o * 8 + j;
      if (c < C) val[j] = fabsf(i1[b * i1_bs + (long long)c * HW + pix] - tap_sample(i2 + b * i2_bs + (long long)c * HW, t));
    }
  }
  mu32x4 hi, lo;
  bool bad = false;
  s16_split8(val, hi, lo, bad);
  mu32x4* d = dst + (b * dst_bs) / 4;
  d[(long long)(o * 2 + 0) * HW + pix] = hi;
  d[(long long)(o * 2 + 1) * HW + pix] = lo;
  if (bad && guard) atomicOr(guard, 1);
}

// downflow8 (AccFlow_.py:138-142): F.interpolate(size=(H/8, W/8), bilinear, align_corners=True) / 8
__global__ __launch_bounds__(256) void downflow8_kernel(const float* __restrict__ flow, float* __restrict__ out,
                                                        int BC, int H, int W) {
  const int H8 = H / 8, W8 = W / 8, P = H8 * W8;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (long long)BC * P) return;
  const int bc = (int)(g / P), pix = (int)(g - (long long)bc * P);
  const int y = pix / W8, x = pix - y * W8;
  const float sy_scale = H8 > 1 ? (float)(H - 1) / (float)(H8 - 1) : 0.0f;
  const float sx_scale = W8 > 1 ? (float)(W - 1) / (float)(W8 - 1) : 0.0f;
  const float sy = sy_scale * (float)y, sx = sx_scale * (float)x;
  const int y0 = (int)sy, x0 = (int)sx;
  const int yp = y0 < H - 1 ? 1 : 0, xp = x0 < W - 1 ? 1 : 0;
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  const float hy = 1.0f - ly, hx = 1.0f - lx;
  const float* p = flow + (long long)bc * H * W + (long long)y0 * W + x0;
  const float v = hy * (hx * p[0] + lx * p[xp]) + ly * (hx * p[(long long)yp * W] + lx * p[(long long)yp * W + xp]);
  out[g] = v / 8.0f;
}

}  // namespace

extern "C" int accflow_convex_upsample_f32(const float* flow, long long flow_bs, const float* mask, long long mask_bs,
                                           float* out, int B, int H8, int W8, void* stream) {
  if (!flow || !mask || !out || B <= 0 || H8 <= 0 || W8 <= 0) return 1;
  const long long np = (long long)B * H8 * W8;
  hipLaunchKernelGGL(convex_upsample_kernel, dim3(cdiv(np, 256)), dim3(256), 0, as_stream(stream), flow, flow_bs,
                     mask, mask_bs, out, B, H8, W8);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

// getOcc, binary form, for small pixel counts (the fusion chain runs it at batch 1: 7 680 pixels - one thread per pixel
// looping over 128 channels x 4 taps was latency-bound, 45 us): lane = pixel (coalesced), the 4 waves of a workgroup
// take a quarter of the channels each, partial sums meet in LDS and are added in a fixed order
// (S16OUT, round 6: `out` is an S16 tensor of one channel - octet 0, channel 0 = the bit - and out_bs counts 4-byte words)
template <bool S16OUT>
__global__ __launch_bounds__(256) void get_occ_binary4_kernel(const float* __restrict__ flow, long long flow_bs,
                                                              const float* __restrict__ i1, long long i1_bs,
                                                              const float* __restrict__ i2, long long i2_bs,
                                                              float* __restrict__ out, long long out_bs, int B, int C, int H,
                                                              int W) {
  __shared__ float part[4][64];
  const int HW = H * W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long gp = (long long)blockIdx.x * 64 + lane;
  const bool live = gp < (long long)B * HW;
  const int b = live ? (int)(gp / HW) : 0, pix = live ? (int)(gp - (long long)b * HW) : 0;
  const int y = pix / W, x = pix - y * W;
  const float u = flow[b * flow_bs + pix], v = flow[b * flow_bs + HW + pix];
  const WarpTaps t = make_taps((float)x + u, (float)y + v, H, W);
  const int cper = (C + 3) / 4, cbeg = wave * cper, cend = min(C, cbeg + cper);
  float s = 0.0f;
  for (int c = cbeg; c < cend; ++c) {
    const float wv = tap_sample(i2 + b * i2_bs + (long long)c * HW, t);
    s += fabsf(i1[b * i1_bs + (long long)c * HW + pix] - wv);
  }
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && live) {
    const float tot = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    const float bit = (tot / (float)C <= 1.0f) ? 1.0f : 0.0f;
    if constexpr (S16OUT) {
      const float val[8] = {bit, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      mu32x4 hi, lo;
      bool bad = false;
      s16_split8(val, hi, lo, bad);        // (0 and 1 always fit)
      mu32x4* d = reinterpret_cast<mu32x4*>(out) + (b * out_bs) / 4;
      d[pix] = hi;
      d[(long long)HW + pix] = lo;
    } else {
      out[b * out_bs + pix] = bit;
    }
  }
}

extern "C" int accflow_backwarp_f32(const float* img, long long img_bs, const float* flow, long long flow_bs,
                                    float* out, long long out_bs, int B, int C, int H, int W, void* stream) {
  if (!img || !flow || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return 1;
  const long long np = (long long)B * H * W;
  hipLaunchKernelGGL(backwarp_kernel<false>, dim3(cdiv(np, 256), cdiv(C, WARP_CCHUNK)), dim3(256), 0, as_stream(stream),
                     img, img_bs, flow, flow_bs, out, out_bs, B, C, H, W);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_compose_flow_f32(const float* step, long long step_bs, const float* acc, long long acc_bs,
                                        float* out, long long out_bs, int B, int H, int W, void* stream) {
  if (!step || !acc || !out || B <= 0 || H <= 0 || W <= 0) return 1;
  const long long np = (long long)B * H * W;
  hipLaunchKernelGGL(backwarp_kernel<true>, dim3(cdiv(np, 256), 1), dim3(256), 0, as_stream(stream), acc, acc_bs, step,
                     step_bs, out, out_bs, B, 2, H, W);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_get_occ_f32(const float* flow, long long flow_bs, const float* i1, long long i1_bs,
                                   const float* i2, long long i2_bs, float* out, long long out_bs, int B, int C, int H,
                                   int W, int binary, void* stream) {
  if (!flow || !i1 || !i2 || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return 1;
  const long long np = (long long)B * H * W;
  if (binary && np < (1 << 17))
    hipLaunchKernelGGL((get_occ_binary4_kernel<false>), dim3(cdiv(np, 64)), dim3(256), 0, as_stream(stream), flow, flow_bs, i1,
                       i1_bs, i2, i2_bs, out, out_bs, B, C, H, W);
  else if (binary)
    hipLaunchKernelGGL((get_occ_kernel<true>), dim3(cdiv(np, 256)), dim3(256), 0, as_stream(stream), flow, flow_bs,
                       i1, i1_bs, i2, i2_bs, out, out_bs, B, C, H, W);
  else
    hipLaunchKernelGGL((get_occ_kernel<false>), dim3(cdiv(np, 256), cdiv(C, WARP_CCHUNK)), dim3(256), 0,
                       as_stream(stream), flow, flow_bs, i1, i1_bs, i2, i2_bs, out, out_bs, B, C, H, W);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_get_occ_s16(const float* flow, long long flow_bs, const float* i1, long long i1_bs, const float* i2,
                                   long long i2_bs, void* out16, long long out16_bs, int* guard, int B, int C, int H, int W,
                                   int binary, void* stream) {
  if (!flow || !i1 || !i2 || !out16 || B <= 0 || C <= 0 || H <= 0 || W <= 0) return 1;
  const long long np = (long long)B * H * W;
  mu32x4* d = reinterpret_cast<mu32x4*>(out16);
  if (binary && np < (1 << 17))
    hipLaunchKernelGGL((get_occ_binary4_kernel<true>), dim3(cdiv(np, 64)), dim3(256), 0, as_stream(stream), flow, flow_bs, i1,
                       i1_bs, i2, i2_bs, reinterpret_cast<float*>(out16), out16_bs, B, C, H, W);
  else if (binary)
    hipLaunchKernelGGL((get_occ_s16_kernel<true>), dim3(cdiv(np, 256)), dim3(256), 0, as_stream(stream), flow, flow_bs, i1, i1_bs,
                       i2, i2_bs, d, out16_bs, guard, B, C, H, W);
  else
    hipLaunchKernelGGL((get_occ_s16_kernel<false>), dim3(cdiv(np, 256), cdiv(C, 8)), dim3(256), 0, as_stream(stream), flow,
                       flow_bs, i1, i1_bs, i2, i2_bs, d, out16_bs, guard, B, C, H, W);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_downflow8_f32(const float* flow, float* out, int B, int C, int H, int W, void* stream) {
  if (!flow || !out || B <= 0 || C <= 0 || H < 8 || W < 8 || (H % 8) || (W % 8)) return 1;
  const long long n = (long long)B * C * (H / 8) * (W / 8);
  hipLaunchKernelGGL(downflow8_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), flow, out, B * C, H, W);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

// Deformable convolution, first half: the deformed im2col "columns" of torchvision.ops.deform_conv2d (modulated,
// one offset group, stride 1; AccFlow_.py:104): cols[b][tap*C + c][y][x] = m_tap * bilinear(x[b][c], y + ky - padH +
// dy_tap, x + kx - padW + dx_tap), dy first; the whole sample is 0 when h <= -1 || h >= H || w <= -1 || w >= W and
// corners outside contribute 0.  The second half is a plain 1x1 convolution with weights w[o][c][tap] -> [o][tap*C+c]
// on the matrix cores (the fused fp32 kernel sampled every (channel, tap) once per 64-channel output block).
namespace {
__global__ __launch_bounds__(256) void deform_columns_kernel(const float* __restrict__ x, long long x_bs,
                                                             const float* __restrict__ off, long long off_bs,
                                                             const float* __restrict__ msk, long long msk_bs,
                                                             float* __restrict__ cols, int B, int C, int H, int W, int KH,
                                                             int KW, int padH, int padW) {
  const int HW = H * W, T = KH * KW;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * T * HW) return;
  const int p = (int)(i % HW), tap = (int)((i / HW) % T), b = (int)(i / ((long long)HW * T));
  const int y = p / W, xx = p - y * W, ky = tap / KW, kx = tap - ky * KW;
  const float dy = off[b * off_bs + (long long)(2 * tap) * HW + p], dx = off[b * off_bs + (long long)(2 * tap + 1) * HW + p];
  const float m = msk[b * msk_bs + (long long)tap * HW + p];
  const float h = (float)(y - padH + ky) + dy, w = (float)(xx - padW + kx) + dx;
  const bool inside = h > -1.0f && h < (float)H && w > -1.0f && w < (float)W;
  const float fh = floorf(h), fw = floorf(w);
  const int hl = (int)fh, wl = (int)fw, hh = hl + 1, wh = wl + 1;
  const float lh = h - fh, lw = w - fw, uh = 1.0f - lh, uw = 1.0f - lw;
  const bool o1 = inside && hl >= 0 && wl >= 0, o2 = inside && hl >= 0 && wh <= W - 1;
  const bool o3 = inside && hh <= H - 1 && wl >= 0, o4 = inside && hh <= H - 1 && wh <= W - 1;
  const int i1 = o1 ? hl * W + wl : 0, i2 = o2 ? hl * W + wh : 0, i3 = o3 ? hh * W + wl : 0, i4 = o4 ? hh * W + wh : 0;
  const float* src = x + b * x_bs;
  float* dst = cols + ((long long)b * T * C + (long long)tap * C) * HW + p;
  // a chunk of the channels per workgroup row (B = 1 in the fusion chain: 69 k (tap, pixel) threads looping over all 128
  // channels were latency-bound, 50 us; the sampling geometry is recomputed per chunk, which is cheap)
  const int cper = (C + gridDim.y - 1) / gridDim.y, cbeg = blockIdx.y * cper, cend = min(C, cbeg + cper);
  for (int c = cbeg; c < cend; ++c) {
    const float* plane = src + (long long)c * HW;
    const float v1 = o1 ? plane[i1] : 0.0f, v2 = o2 ? plane[i2] : 0.0f, v3 = o3 ? plane[i3] : 0.0f, v4 = o4 ? plane[i4] : 0.0f;
    dst[(long long)c * HW] = inside ? m * (uh * uw * v1 + uh * lw * v2 + lh * uw * v3 + lh * lw * v4) : 0.0f;
  }
}

// The same columns PRE-SPLIT (round 6): the 1x1 convolution behind them stages S16 chunks by LDS DMA, and nothing else reads
// them.  Thread = (pixel, tap), blockIdx.y = chunk of octets; a thread writes whole 16-byte chunks (8 channels of its tap,
// column channel tap*C + c: C % 8 == 0).  LOGIT: `msk` holds the modulation's LOGITS and the sigmoid (AccFlow_.py:103) is
// applied here - the in-place activation launch over the ZeroConv output's last 9 channels is gone.
template <bool LOGIT>
__global__ __launch_bounds__(256) void deform_columns_s16_kernel(const float* __restrict__ x, long long x_bs,
                                                                 const float* __restrict__ off, long long off_bs,
                                                                 const float* __restrict__ msk, long long msk_bs,
                                                                 mu32x4* __restrict__ cols, long long cols_bs, int* guard, int B,
                                                                 int C, int H, int W, int KH, int KW, int padH, int padW) {
  const int HW = H * W, T = KH * KW;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * T * HW) return;
  const int p = (int)(i % HW), tap = (int)((i / HW) % T), b = (int)(i / ((long long)HW * T));
  const int y = p / W, xx = p - y * W, ky = tap / KW, kx = tap - ky * KW;
  const float dy = off[b * off_bs + (long long)(2 * tap) * HW + p], dx = off[b * off_bs + (long long)(2 * tap + 1) * HW + p];
  float m = msk[b * msk_bs + (long long)tap * HW + p];
  if constexpr (LOGIT) m = sigmoidf_(m);
  const float h = (float)(y - padH + ky) + dy, w = (float)(xx - padW + kx) + dx;
  const bool inside = h > -1.0f && h < (float)H && w > -1.0f && w < (float)W;
  const float fh = floorf(h), fw = floorf(w);
  const int hl = (int)fh, wl = (int)fw, hh = hl + 1, wh = wl + 1;
  const float lh = h - fh, lw = w - fw, uh = 1.0f - lh, uw = 1.0f - lw;
  const bool o1 = inside && hl >= 0 && wl >= 0, o2 = inside && hl >= 0 && wh <= W - 1;
  const bool o3 = inside && hh <= H - 1 && wl >= 0, o4 = inside && hh <= H - 1 && wh <= W - 1;
  const int i1 = o1 ? hl * W + wl : 0, i2 = o2 ? hl * W + wh : 0, i3 = o3 ? hh * W + wl : 0, i4 = o4 ? hh * W + wh : 0;
  const float* src = x + b * x_bs;
  const int OC = C >> 3;                                   // octets per tap
  const int oper = (OC + gridDim.y - 1) / gridDim.y, obeg = blockIdx.y * oper, oend = min(OC, obeg + oper);
  mu32x4* d = cols + (b * cols_bs) / 4;
  bool bad = false;
  for (int o = obeg; o < oend; ++o) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* plane = src + (long long)(o * 8 + j) * HW;
      const float v1 = o1 ? plane[i1] : 0.0f, v2 = o2 ? plane[i2] : 0.0f, v3 = o3 ? plane[i3] : 0.0f, v4 = o4 ? plane[i4] : 0.0f;
      v[j] = inside ? m * (uh * uw * v1 + uh * lw * v2 + lh * uw * v3 + lh * lw * v4) : 0.0f;   // (deform_columns_kernel's expression)
    }
    mu32x4 hi, lo;
    s16_split8(v, hi, lo, bad);
    const long long oct = (long long)tap * OC + o;
    d[(oct * 2 + 0) * HW + p] = hi;
    d[(oct * 2 + 1) * HW + p] = lo;
  }
  if (bad && guard) atomicOr(guard, 1);
}
}  // namespace

extern "C" int accflow_deform_columns_s16(const float* x, long long x_bs, const float* offset, long long offset_bs,
                                          const float* dmask, long long dmask_bs, int mask_is_logit, void* cols16,
                                          long long cols16_bs, int* guard, int B, int C, int H, int W, int KH, int KW, int padH,
                                          int padW, void* stream) {
  if (!x || !offset || !dmask || !cols16 || B <= 0 || C <= 0 || (C & 7) || H <= 0 || W <= 0 || KH <= 0 || KW <= 0) return 1;
  const long long n = (long long)B * KH * KW * H * W;
  const int chunks = n < (1 << 18) ? (C >= 64 ? 8 : 1) : (n < (1 << 20) ? 2 : 1);
  mu32x4* d = reinterpret_cast<mu32x4*>(cols16);
  if (mask_is_logit)
    hipLaunchKernelGGL((deform_columns_s16_kernel<true>), dim3(cdiv(n, 256), chunks), dim3(256), 0, as_stream(stream), x, x_bs,
                       offset, offset_bs, dmask, dmask_bs, d, cols16_bs, guard, B, C, H, W, KH, KW, padH, padW);
  else
    hipLaunchKernelGGL((deform_columns_s16_kernel<false>), dim3(cdiv(n, 256), chunks), dim3(256), 0, as_stream(stream), x, x_bs,
                       offset, offset_bs, dmask, dmask_bs, d, cols16_bs, guard, B, C, H, W, KH, KW, padH, padW);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}

extern "C" int accflow_deform_columns_f32(const float* x, long long x_bs, const float* offset, long long offset_bs,
                                          const float* dmask, long long dmask_bs, float* cols, int B, int C, int H, int W,
                                          int KH, int KW, int padH, int padW, void* stream) {
  if (!x || !offset || !dmask || !cols || B <= 0 || C <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0) return 1;
  const long long n = (long long)B * KH * KW * H * W;
  const int chunks = n < (1 << 18) ? (C >= 64 ? 8 : 1) : (n < (1 << 20) ? 2 : 1);
  hipLaunchKernelGGL(deform_columns_kernel, dim3(cdiv(n, 256), chunks), dim3(256), 0, as_stream(stream), x, x_bs, offset, offset_bs,
                     dmask, dmask_bs, cols, B, C, H, W, KH, KW, padH, padW);
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
