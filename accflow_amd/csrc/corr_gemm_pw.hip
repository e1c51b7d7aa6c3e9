// Levels 0 and 1 of the displaced correlation pyramid (raft/corr.py:8-22,47-55) for a batch of pairs in ONE launch of
// PERSISTENT workgroups whose displaced store runs underneath the next tile's matrix work.
//
// Why.  corr_disp_ring_kernel (conv2d_direct.hip; one launch per pair, one 128 x 128 tile per workgroup: K loop, then the
// LDS shear + stores) takes 141 us per pair at 60 x 128, and ablation builds show its phases ADDING UP
// (profiles/r05_corr_gemm_experiments.txt): the store phase 68 us - 313 MB at the 4.6 TB/s a store-only kernel reaches -,
// the operand DMA 39, the MFMAs 18, skeleton ~30; more workgroups per CU, a deeper operand ring, phase-shifted starts,
// aligned store pieces or cache-policy bits change nothing.  The store of a tile has to overlap with matrix work inside
// ONE workgroup.  So: one workgroup per CU walks over ~155 tiles, 8 waves in two roles (as corr_lookup_conv.hip):
//   * waves 0-3 MULTIPLY: the ring kernel's K loop as one continuous stream of 16-deep steps ACROSS tiles - operand tiles
//     DMA'd into a 3-slot LDS ring three steps ahead, the fragments of step g+1 read from LDS while the 12 MFMAs of step
//     g run (two register sets), one workgroup barrier per step; at the end of a tile the accumulators go to a full-tile
//     LDS staging buffer (2 target rows x 64 columns x 128 query pixels, fp32) and the next tile starts at once;
//   * waves 4-7 STORE: during the 16 steps of the next tile they shear the staged tile out of LDS (corr_disp_store2's
//     diagonals: 64 consecutive query pixels of one displacement row per store instruction) and form level 1 from both
//     staged rows, 5 store instructions per wave and step.  Their vector-memory counter holds only stores, the
//     multipliers' only operand DMA.
// Same products in the same order, the same ((a + b) + c) + d pooling: bit-identical to corr_disp_ring_kernel.
#include "conv_common.h"
#include <utility>

namespace {

template <int... I, class F>
__device__ __forceinline__ void cpw_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void cpw_static_for(F&& f) { cpw_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

constexpr int CPW_SLOT_CHUNKS = 2 * 2 * 2 * 128;          // [A|B][term][octet][128] 16-byte chunks of one 16-deep step
constexpr int CPW_SLOTS = 3;
constexpr int CPW_RING_BYTES = CPW_SLOTS * CPW_SLOT_CHUNKS * 16;
constexpr int CPW_STAGE_FLOATS = 2 * 64 * DISP_PITCH;     // [target row h][column c][query pixel p]
constexpr int CPW_LDS_BYTES = CPW_RING_BYTES + CPW_STAGE_FLOATS * 4;
constexpr int CPW_NSTEP = 16;                           // 16-deep steps per tile: C = 256 (other channel counts: the ring kernel)
constexpr int CPW_MAXPAIRS = 16;

struct cpw_params {
  const char* packs; long long pack_bytes;     // per-frame operand packs (accflow_corr_pack_f32)
  float* lvl0; float* lvl1;
  long long pair0, pair1;                      // floats per pair of level 0 / 1
  int npairs, H8, W8, Kpad, CoutPad;
  float osc;
  int idx1[CPW_MAXPAIRS], idx2[CPW_MAXPAIRS];  // query / target frame of each pair
};

__device__ __forceinline__ int cpw_karg(int byte_off) {   // a dword of the kernel arguments by a run-time offset (scalar load)
  typedef const __attribute__((address_space(4))) int* kint_ptr;
  return *(kint_ptr)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + byte_off);
}

struct cpw_tile { int pair, pb, yo, xc; };

// tools only (accflow_debug_cpw_prof): per workgroup {work, wait, barriers} cycle sums of wave 0 (multiplying) and wave 4
// (storing): work = from leaving a barrier to arriving at the next one, wait = inside the barrier
unsigned long long* g_cpw_prof = nullptr;
#define CPW_BARRIER()                                                        \
  do {                                                                       \
    if (PROF) { const unsigned long long t_ = __builtin_readcyclecounter(); pw_work += t_ - pw_last; pw_last = t_; } \
    __builtin_amdgcn_s_barrier();                                            \
    if (PROF) { const unsigned long long t_ = __builtin_readcyclecounter(); pw_wait += t_ - pw_last; pw_last = t_; ++pw_n; } \
  } while (0)

template <bool PROF>
__global__ __launch_bounds__(1024, 4) void corr_disp_pw_kernel(const cpw_params a, unsigned long long* prof) {
  unsigned long long pw_work = 0, pw_wait = 0, pw_n = 0, pw_last = PROF ? __builtin_readcyclecounter() : 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* ring = reinterpret_cast<u32x4*>(smem);
  float* T = reinterpret_cast<float*>(smem + CPW_RING_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H8 = a.H8, W8 = a.W8, P = H8 * W8;
  const int npb = (P + 127) >> 7;
  const int ncx = (W8 + 63) >> 6, nqt = ((H8 + 1) >> 1) * ncx;
  // XCD-aware tile order (as the ring kernel): workgroups w and w + 8 share an XCD and its L2; XCD x owns the query blocks
  // x, x + 8, ...; inside the class the query block runs fastest, then the target tile, then the pair
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, G8 = gridDim.x >> 3;
  const int percol = xcd < npb ? (npb - xcd + 7) >> 3 : 0;
  const int ntot = a.npairs * nqt * percol;
  auto tile_of = [&](int t) {
    cpw_tile ti;
    const int pbi = t % percol, rest = t / percol;
    const int qt = rest % nqt;
    ti.pair = rest / nqt;
    ti.pb = xcd + 8 * pbi;
    ti.yo = qt / ncx;
    ti.xc = qt - ti.yo * ncx;
    return ti;
  };
  const int nmine = j < ntot ? (ntot - j + G8 - 1) / G8 : 0;   // tiles of this workgroup: t = j + r * G8
  if (nmine == 0) return;
  constexpr int NSTEP = CPW_NSTEP;                             // (host-checked: Kpad = 16 * NSTEP)
  const unsigned oct_bytes = (unsigned)a.CoutPad * 16u, term_bytes = (unsigned)(a.Kpad / 8) * oct_bytes;

  if (wave < 8) {
    // ======================= multiplying waves: 64 query pixels x 32 target pixels each =======================
    const int wc = wave >> 2, wq = wave & 3;
    const int l31 = lane & 31, kh = lane >> 5;
    // operand DMA: 16 pieces per step (corr_disp_ring_kernel's), piece id = wave * 2 + i: A = ids 0-7 (waves 0-3), B = ids 8-15
    struct dma_addr { const char* base; unsigned pvoff[2]; };
    auto dma_setup = [&](int r) {                // addresses of this workgroup's tile r; beyond the last tile: masked (zeros)
      dma_addr d;
      const bool ok = r < nmine;
      const cpw_tile ti = tile_of(ok ? j + r * G8 : j);
      const int fr = cpw_karg((wave < 4 ? __builtin_offsetof(cpw_params, idx1) : __builtin_offsetof(cpw_params, idx2)) + ti.pair * 4);
      d.base = a.packs + (long long)fr * a.pack_bytes;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int half = (wave * 2 + i) & 1;
        if (wave < 4) {
          d.pvoff[i] = ok ? (unsigned)(ti.pb * 128 + half * 64 + lane) * 16u : 0xFFFFFFFFu;   // (rows beyond P are zero in the pack)
        } else {
          const int y2 = 2 * ti.yo + half, x2 = ti.xc * 64 + lane;
          d.pvoff[i] = (ok && y2 < H8 && x2 < W8) ? (unsigned)(y2 * W8 + x2) * 16u : 0xFFFFFFFFu;
        }
      }
      return d;
    };
    auto issue = [&](const dma_addr& d, int step, int slot) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(d.base), 0, (int)(2 * term_bytes), 0x00020000);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int id = wave * 2 + i;
        const int isb = id >> 3, row = (id >> 1) & 3, half = id & 1;
        const int t = row >> 1, o = row & 1;
        const unsigned soff = (unsigned)t * term_bytes + (unsigned)(2 * step + o) * oct_bytes;
        const int dst = slot * CPW_SLOT_CHUNKS + isb * 512 + row * 128 + half * 64;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)&ring[dst], 16, (int)d.pvoff[i], (int)soff, 0, 0);
      }
    };
    auto read_frags = [&](int slot, bf16x8 (&A)[2][2], bf16x8 (&Bf)[2]) {
      const u32x4* sl = ring + slot * CPW_SLOT_CHUNKS;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) A[t][i] = __builtin_bit_cast(bf16x8, sl[(t * 2 + kh) * 128 + wc * 64 + i * 32 + l31]);
        Bf[t] = __builtin_bit_cast(bf16x8, sl[512 + (t * 2 + kh) * 128 + wq * 32 + l31]);
      }
    };
    f32x16 acc[2];
    auto zero_acc = [&]() {
#pragma unroll
      for (int tc = 0; tc < 2; ++tc)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tc][r] = 0.0f;
    };
    auto mfma3 = [&](const bf16x8 (&A)[2][2], const bf16x8 (&Bf)[2]) {
      constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};   // (w_lo x_hi), (w_hi x_lo), (w_hi x_hi): the ring kernel's order
#pragma unroll
      for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
          acc[tc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[PA[pr]][tc]), __builtin_bit_cast(f16x8, Bf[PB[pr]]),
                                                          acc[tc], 0, 0, 0);
    };
    zero_acc();
    // The step stream: step s of tile r lives in ring slot (r + s) % 3 (16 % 3 = 1); the DMA runs 3 steps ahead of the MFMAs
    // - ACROSS tiles: steps 13-15 of a tile request steps 0-2 of the next one -, the fragment reads 1 step ahead.  The 16
    // steps of a tile are straight-line code (no branch between a step's fragment reads and its MFMAs: at a join the
    // compiler waits for the reads - the first form of this loop paid an LDS round trip per step that way); requests beyond
    // the last tile are masked (they deliver zeros nobody multiplies).
    dma_addr dc = dma_setup(0), dn = dma_setup(1);
    issue(dc, 0, 0);
    issue(dc, 1, 1);
    issue(dc, 2, 2);
    bf16x8 A0[2][2], B0[2], A1[2][2], B1[2];
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // step 0's two pieces are the oldest of the six in flight
    CPW_BARRIER();                // (every multiplying wave's pieces of step 0; the storing waves join)
    read_frags(0, A0, B0);
    int rb = 0;                   // r % 3
    for (int r = 0; r < nmine; ++r) {
      cpw_static_for<NSTEP>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this step's fragments are in registers (slot reuse below)
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     // my pieces of step s + 1 (those of s + 2 may be in flight)
        CPW_BARRIER();
        int s0 = rb + (s % 3), s1 = rb + ((s + 1) % 3);
        s0 -= s0 >= 3 ? 3 : 0;
        s1 -= s1 >= 3 ? 3 : 0;
        if constexpr (s + 3 < NSTEP) issue(dc, s + 3, s0);   // step s + 3 into the slot step s was read from
        else issue(dn, s + 3 - NSTEP, s0);
        if constexpr ((s & 1) == 0) { read_frags(s1, A1, B1); mfma3(A0, B0); }
        else { read_frags(s1, A0, B0); mfma3(A1, B1); }
      });
      // ---- hand-over: the staged tile r - 1 has been drained; stage tile r ----
      CPW_BARRIER();
#pragma unroll
      for (int tc = 0; tc < 2; ++tc)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const f32x4 v = {acc[tc][4 * r4], acc[tc][4 * r4 + 1], acc[tc][4 * r4 + 2], acc[tc][4 * r4 + 3]};
          *reinterpret_cast<f32x4*>(&T[(wq * 32 + l31) * DISP_PITCH + wc * 64 + tc * 32 + 8 * r4 + 4 * (lane >> 5)]) = v;
        }
      zero_acc();
      dc = dn;
      dn = dma_setup(r + 2);
      rb = rb == 2 ? 0 : rb + 1;
      CPW_BARRIER();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (masked requests past the last tile)
    for (int s = 0; s < NSTEP + 2; ++s) CPW_BARRIER();       // the storing waves drain the last tile
    if (PROF && prof && tid == 0) { prof[blockIdx.x * 8 + 0] = pw_work; prof[blockIdx.x * 8 + 1] = pw_wait; prof[blockIdx.x * 8 + 2] = pw_n; prof[blockIdx.x * 8 + 3] = nmine; }
    return;
  }

  // ======================= storing waves: lane = query pixel =======================
  // Wave sw owns the query half-block ph = sw & 1 (pl = ph * 64 + lane) and a quarter g4 = sw >> 1 of its work.  Level 0:
  // for target row h and u = 0..63 the lane reads column c = (lane + u) & 63 of ITS pixel - the lanes of a store are 64
  // consecutive query pixels whose displacement (dy, dx) agrees wherever x1 and c advance together: corr_disp_store2's
  // diagonals with the roles of lane and column swapped, so that the pixel's (y1, x1) are per-lane constants of the tile
  // (no table in LDS, a dozen VALU instructions per store).  LDS reads: bank (4 c + pl) mod 32 = (5 lane + 4 u) mod 32,
  // conflict-free.  Level 1: corr_disp_store2's cells, k = (g4 * 8 + i + lane / 2) & 31, both staged rows at once.
  const int sw = wave - 8;
  const int ph = sw & 1, g4 = sw >> 1;
  const int H1 = H8 >> 1, W1 = W8 >> 1;
  const int pl = ph * 64 + lane;
  CPW_BARRIER();                  // (the multiplying waves' barrier ahead of step 0)
  cpw_tile prev = tile_of(j);
  for (int r = 0; r <= nmine; ++r) {
    const bool drain = r > 0;                    // tile r - 1 sits in the staging buffer
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
        a.lvl0 + (long long)prev.pair * a.pair0 + (long long)prev.pb * P * 128, 0, (int)((unsigned)P * 512u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rl1 = __builtin_amdgcn_make_buffer_rsrc(
        a.lvl1 + (long long)prev.pair * a.pair1 + (long long)prev.pb * H1 * W1 * 128, 0, (int)((unsigned)(H1 * W1) * 512u), 0x00020000);
    const int yo = prev.yo, xc = prev.xc;
    // per-lane constants of the drained tile
    const int p = prev.pb * 128 + pl;
    const bool pok = p < P;
    const int y1 = p / W8, x1 = p - y1 * W8;
    int rowb[2];                                 // (dy mod H8) * W8 of target row h, or -1: row outside the image / pixel padding
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int dy = 2 * yo + h - y1;
      dy += (dy >> 31) & H8;
      rowb[h] = (pok && 2 * yo + h < H8) ? dy * W8 : -1;
    }
    const int xbase = xc * 64 - x1;              // dx = c + xbase (mod W8)
    int dy1 = yo - (y1 >> 1);
    dy1 += (dy1 >> 31) & H1;
    const int row1 = (pok && yo < H1) ? dy1 * W1 : -1;
    const int x1h = x1 >> 1;
    cpw_static_for<NSTEP>([&](auto s_) {
      constexpr int s = decltype(s_)::value;
      CPW_BARRIER();
      if (drain) {
        // level 0: units e = g4 * 32 + 2 s, + 1 of this half-block's 128 (h = e >> 6, u = e & 63)
        float v[2];
        int cc[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int e = g4 * 32 + 2 * s + q;
          const int h = e >> 6, u = e & 63;
          cc[q] = (lane + u) & 63;
          v[q] = T[(h * 64 + cc[q]) * DISP_PITCH + pl];
        }
        float l1[4];
        int k1 = 0;
        constexpr bool do1 = (s & 1) == 0;         // level 1: unit i = s / 2 of this wave's 8
        if constexpr (do1) {
          k1 = (g4 * 8 + (s >> 1) + (lane >> 1)) & 31;
          l1[0] = T[(2 * k1) * DISP_PITCH + pl];
          l1[1] = T[(2 * k1 + 1) * DISP_PITCH + pl];
          l1[2] = T[(64 + 2 * k1) * DISP_PITCH + pl];
          l1[3] = T[(64 + 2 * k1 + 1) * DISP_PITCH + pl];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int e = g4 * 32 + 2 * s + q;
          const int h = e >> 6;
          const int x2 = xc * 64 + cc[q];
          int dx = cc[q] + xbase;
          dx += (dx >> 31) & W8;
          const int rbq = h ? rowb[1] : rowb[0];
          const unsigned off = ((unsigned)(rbq + dx) * 128u + (unsigned)pl) * 4u;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[q] * a.osc), rout, (int)((rbq >= 0 && x2 < W8) ? off : 0xFFFFFFFFu), 0, 0);
        }
        if constexpr (do1) {
          const float vv = ((((l1[0] + l1[1]) + l1[2]) + l1[3]) * a.osc) * 0.25f;   // corr_disp_store2's / the pool kernel's order
          const int xo = xc * 32 + k1;
          int dx = xo - x1h;
          dx += (dx >> 31) & W1;
          const unsigned off = ((unsigned)(row1 + dx) * 128u + (unsigned)pl) * 4u;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vv), rl1, (int)((row1 >= 0 && xo < W1) ? off : 0xFFFFFFFFu), 0, 0);
        }
      }
    });
    CPW_BARRIER();                // hand-over: this wave has read everything it needs of the staged tile
    if (r < nmine) prev = tile_of(j + r * G8);
    CPW_BARRIER();                // tile r is staged
  }
  if (PROF && prof && tid == 512) { prof[blockIdx.x * 8 + 4] = pw_work; prof[blockIdx.x * 8 + 5] = pw_wait; prof[blockIdx.x * 8 + 6] = pw_n; }
}
#undef CPW_BARRIER

}  // namespace

// tools only (tools/cpw_prof.py): device buffer of 8 x workgroups uint64, NULL = off.  Process-wide, not thread-safe.
extern "C" int accflow_debug_cpw_prof(unsigned long long* buf) { g_cpw_prof = buf; return 0; }

// levels 0 and 1 of `B` pairs (frames idx1[b] -> idx2[b] of the per-frame packs); returns 0 or a hipError_t
int accflow_launch_corr_disp_pw(const void* packs, long long pack_bytes, const int* idx1, const int* idx2, float* lvl0,
                                float* lvl1, long long pair0, long long pair1, int B, int H8, int W8, int Kpad, int CoutPad,
                                float osc, hipStream_t st) {
  static const int ok = [] {
    const int r = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(corr_disp_pw_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, CPW_LDS_BYTES);
    return r ? r : (int)hipFuncSetAttribute(reinterpret_cast<const void*>(corr_disp_pw_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            CPW_LDS_BYTES);
  }();
  if (ok != 0) return ok;
  static const int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 8 ? n / 8 * 8 : 8;
  }();
  for (int b0 = 0; b0 < B; b0 += CPW_MAXPAIRS) {
    cpw_params a = {};
    a.packs = reinterpret_cast<const char*>(packs); a.pack_bytes = pack_bytes;
    a.lvl0 = lvl0 + (long long)b0 * pair0; a.lvl1 = lvl1 + (long long)b0 * pair1;
    a.pair0 = pair0; a.pair1 = pair1;
    a.npairs = B - b0 < CPW_MAXPAIRS ? B - b0 : CPW_MAXPAIRS;
    a.H8 = H8; a.W8 = W8; a.Kpad = Kpad; a.CoutPad = CoutPad; a.osc = osc;
    for (int i = 0; i < a.npairs; ++i) { a.idx1[i] = idx1[b0 + i]; a.idx2[i] = idx2[b0 + i]; }
    if (g_cpw_prof) hipLaunchKernelGGL(corr_disp_pw_kernel<true>, dim3(ncu), dim3(1024), CPW_LDS_BYTES, st, a, g_cpw_prof);
    else hipLaunchKernelGGL(corr_disp_pw_kernel<false>, dim3(ncu), dim3(1024), CPW_LDS_BYTES, st, a, (unsigned long long*)nullptr);
  }
  ACCFLOW_RETURN_LAUNCH_STATUS();
}
