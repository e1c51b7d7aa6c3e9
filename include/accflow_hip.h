/*
 * accflow_hip.h -- C-ABI of libaccflow_hip.so, the gfx950 (MI355X) kernel library behind the
 * AccFlow inference hot path (RAFT / GMA pair estimator + AccFlow backward accumulation).
 *
 * The reference (mulns/AccFlow) has no FFI of its own: every op below replaces a PyTorch /
 * torchvision library call made from the reference's Python (file:line cited per entry, relative to
 * the reference root).  A maintainer binds these entry points with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data unless the name says otherwise; tensors are
 *     NCHW, inner (H, W) plane contiguous; where a `*_bs` argument exists it is the batch stride in
 *     ELEMENTS, so a channel slice [c0:c1) of a larger (B, C, H, W) buffer can be passed without a
 *     copy (pointer = base + c0*H*W, bs = C*H*W);
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); kernels are only enqueued on it,
 *     never synchronised; the library allocates no device memory, never calls a synchronising HIP API and
 *     keeps NO mutable global state (the only statics are environment-derived constants read once): every
 *     piece of state - workspaces, the fp16 range-guard flag - is a caller-owned buffer passed per call, so
 *     concurrent calls from several host threads and for several devices are safe as long as the buffers
 *     of two in-flight calls do not overlap (nn.DataParallel drives the modules with one thread per GPU,
 *     test_cvo.py:18,26);
 *   - return value: 0 on success, otherwise a hipError_t cast to int (launch-configuration errors
 *     are reported as hipErrorInvalidValue = 1).  Nothing throws or aborts.
 */
#ifndef ACCFLOW_HIP_H
#define ACCFLOW_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define ACCFLOW_ABI_VERSION 20

/* activation applied to (acc + bias) */
enum { ACCFLOW_ACT_NONE = 0, ACCFLOW_ACT_RELU = 1, ACCFLOW_ACT_SIGMOID = 2, ACCFLOW_ACT_TANH = 3 };

/* conv epilogues; v = act(acc + bias[ch]) */
enum {
  ACCFLOW_EPI_STORE = 0,    /* out = v                                                             */
  ACCFLOW_EPI_RES_RELU = 1, /* out = relu(e0 + v)            residual block end, extractor.py:62   */
  ACCFLOW_EPI_GRU_ZR = 2,   /* ch <  Cout/2: out[ch] = v (z);  ch >= Cout/2: out2[ch-Cout/2] =     */
                            /*   v * e0[ch-Cout/2]  (r*h)        update.py:47-49 / 54-56           */
  ACCFLOW_EPI_GRU_Q = 3,    /* out = (1-e1)*e0 + e1*v   (h' = (1-z)h + z q)   update.py:50-51      */
  ACCFLOW_EPI_ACCUM = 4,    /* out = e0 + v             (coords1 += delta)    raft.py:136          */
  ACCFLOW_EPI_TAPGEMM = 5   /* v never leaves the workgroup: it is the input of a SECOND, small-Cout     */
                            /* convolution whose per-tap 1x1 products are written instead (desc.tg_*;    */
                            /* FlowHead: conv2(relu(conv1(net))), update.py:12-13)                       */
};

/* arithmetic of the MFMA convolution kernel (accflow_conv_desc.mode) */
enum {
  ACCFLOW_CONV_F32 = 0,    /* fp32-input MFMA, bitwise an fp32 fmaf chain                                  */
  ACCFLOW_CONV_BF16X3 = 2, /* operands split into 2 bf16 terms, 3 bf16 MFMAs, fp32 accumulate (~2^-16)     */
  ACCFLOW_CONV_BF16X6 = 3, /* operands split into 3 bf16 terms, 6 bf16 MFMAs, fp32 accumulate (~2^-23)     */
  /* operands split into 2 fp16 terms (hi + lo = x * 2^s, round to nearest each), 3 fp16 MFMAs, fp32 accumulate, the
   * power-of-two scales undone exactly in the epilogue.  Error model per operand: hi + lo carries 22 significant bits
   * while lo is a NORMAL fp16 number, i.e. for |x * 2^s| >= 2^-3; below that lo is an fp16 subnormal and the operand
   * keeps an ABSOLUTE resolution of 2^-25 / 2^s.  The scales make that floor irrelevant:
   *   - weights: each output row is scaled at pack time so that its largest magnitude lies in [2^10, 2^11)
   *     (accflow_conv_pack_patch16), so every weight within 2^-13 of its row's maximum has 22 bits and smaller ones an
   *     absolute error of 2^-35 of the row maximum - whatever the checkpoint's weight magnitudes;
   *   - activations: scaled by 2^ACCFLOW_F16_ASHIFT in the kernel: 22 bits for |x| >= 2^-7, absolute resolution 2^-29
   *     (1.9e-9) below, and |x| >= 65520 / 2^ACCFLOW_F16_ASHIFT = 4095 overflows - the kernels then OR 1 into *guard
   *     and the caller recomputes in BF16X6.
   * The product drops the lo*lo term (2^-22 relative).  Used by the direct kernel when wpatch16 is given and by the
   * displaced correlation GEMM; every other kernel runs BF16X6 arithmetic in this mode. */
  ACCFLOW_CONV_F16X3 = 4
};
#define ACCFLOW_F16_ASHIFT 4

/* One source of the multi-source S16 form of accflow_conv_desc (nsrc > 0): an S16 tensor (see accflow_conv_desc.in_fmt
 * for the format) of C channels and Hs x Ws pixels, read through a pixel step and with a sub-kernel of its own.  For
 * output pixel (y, x) and tap (ty, tx) in KH x KW the source contributes the input pixel
 *     (step * (y + ty - padH) + oy,  step * (x + tx - padW) + ox)      - zero when outside Hs x Ws -
 * of each of its channels.  step = 1, (oy, ox) = (0, 0) and the conv's own KH / KW / pad is a plain member of a channel
 * concatenation (torch.cat([...], 1) feeding a conv: AccFlow_.py:98-107 with 3 and 4 members); step = 2 with the four
 * origins (py, px) and the taps of that parity expresses a STRIDE-2 convolution (extractor.py:9,52) as stride-1 work:
 * a 3x3, stride 2, pad 1 conv = sources {(0,0): 1x1}, {(0,1): 1x2, padW 1}, {(1,0): 2x1, padH 1}, {(1,1): 2x2, pads 1}
 * with weights w[:, :, 1 + ...]: tap ty of a class-1 axis is original tap 2*ty, of a class-0 axis original tap 1. */
typedef struct accflow_conv_src {
  const void* ptr;               /* S16 tensor: (B, ceil(C/8), 2 terms, Hs, Ws) 16-byte chunks                         */
  long long bs;                  /* batch stride in 4-byte words                                                      */
  int C, Hs, Ws;
  int step, oy, ox;
  int KH, KW, padH, padW;
  int reserved;                  /* 0; src[0].reserved = 1 + wave layout forces that layout (tests / tuning)            */
} accflow_conv_src;
#define ACCFLOW_CONV_MAX_SRC 4

/* One direct (implicit-GEMM) 2-D convolution, cross-correlation as nn.Conv2d, groups=1, dilation=1.
 * The input is the channel concatenation of up to two tensors (in1 may be NULL with C1 = 0), which
 * is how torch.cat([...], dim=1) feeding a conv (update.py:46,49,93,129; AccFlow_.py:98-107) is
 * expressed without materialising the cat.  Weights are pre-packed by accflow_conv_pack_f32. */
typedef struct accflow_conv_desc {
  const float* in0;  const float* in1;
  long long in0_bs, in1_bs;
  int C0, C1;                    /* Cin = C0 + C1                                                  */
  int B, H, W;                   /* input size                                                     */
  int OH, OW;                    /* output size = floor((H + 2*pad - K)/stride) + 1                */
  int KH, KW, stride, padH, padW;
  int Cout;
  const float* wpack;            /* [Kpad][CoutPad], from accflow_conv_pack_f32                    */
  const int*   ktab;             /* [Kpad] int4 {c_in_src, ky, kx, src}                            */
  int Kpad, CoutPad;
  const float* bias;             /* [Cout] or NULL                                                 */
  float* out;  long long out_bs;
  int act, epi;
  const float* e0; long long e0_bs;
  const float* e1; long long e1_bs;
  float* out2; long long out2_bs;
  /* deformable mode (torchvision.ops.deform_conv2d, modulated, 1 offset group; AccFlow_.py:104):
   * offset = (B, 2*KH*KW, OH, OW) with channel 2t = dy, 2t+1 = dx of tap t; dmask = (B, KH*KW, ..) */
  const float* offset; long long offset_bs;
  const float* dmask;  long long dmask_bs;
  /* split-bf16 path: wsplit = [3][Kpad/8][CoutPad][8] bf16 from accflow_conv_pack_bf16s (NULL: fp32 only) */
  const void* wsplit;
  int mode;                      /* ACCFLOW_CONV_*                                                 */
  /* patch kernel (stride-1 "same" convs, split-bf16 modes): weights in (16-channel chunk, tap) step order from
   * accflow_conv_pack_patch (NULL: im2col kernels only) */
  const void* wpatch;
  long long wsplit_bs;           /* != 0: one split weight matrix per batch item, this many BYTES apart (GMA)  */
  /* optional split-K workspace (>= B*Cout*OH*OW floats per part): small grids (batch-1 fusion chain) run the direct
   * kernel in 2-4 K-parts into it and a second kernel sums the parts in a fixed order and applies the epilogue */
  float* kws; long long kws_elems;
  /* ACCFLOW_CONV_F16X3: the fp16 pack from accflow_conv_pack_patch16 and a device int the kernels OR with 1 when an
   * activation does not fit fp16's range (NULL: no report) */
  const void* wpatch16; int* guard;
  /* per-output-channel multiplier applied to the accumulator before the bias (NULL = 1): the inverse of the fp16
   * pack's row scale and of the activation scale, written by accflow_conv_pack_patch16 ([CoutPad] floats) */
  const float* wscale16;
  /* fp16 form of wsplit for the im2col kernel (strided / 7x7-stem / < 16-input-channel convs in ACCFLOW_CONV_F16X3):
   * [2 (+1 unused)][Kpad/8][CoutPad][8] fp16 from accflow_conv_pack_split16, same row scales (wscale16); NULL: those
   * convs run BF16X6 arithmetic */
  const void* wsplit16;
  /* InstanceNorm statistics of the OUTPUT, gathered in the epilogue (ACCFLOW_EPI_STORE + ACCFLOW_ACT_NONE only): every
   * wave writes, for each of its output channels, {sum, sum of squared deviations from its own mean, count} over the
   * pixels it holds to stats[((b * Cout + ch) * stat_slots + slot) * 3 ..]; stat_slots must equal
   * accflow_conv_stat_slots(desc) (0 there = this call cannot produce them).  accflow_instance_norm_apply_f32 combines
   * the partials in a fixed order (deterministic) - extractor.py:36-39,56-63 without re-reading the plane for mean and
   * variance.  NULL: nothing is written. */
  float* stats; int stat_slots;
  /* optional pre-activation addend, (B, Cout, OH, OW) fp32: v = act(acc + bias[ch] + pre[b, ch, y, x]) (GRU epilogues:
   * the contribution of the iteration-invariant context features `inp` to the gate convolutions, update.py:46-50, is
   * convolved once per pair and added here in each of the 12 iterations); NULL: none */
  const float* pre; long long pre_bs;
  /* normalise-on-load: in0 is the raw output of a convolution whose InstanceNorm statistics are final - in_norm =
   * {mean, 1/sqrt(var + eps)} per (batch item, channel) of in0, (B, C0, 2) floats as written by
   * accflow_instance_stats_finalize_f32 - and the kernel reads relu((x - mean) * rstd) instead of x (zero padding
   * unchanged): extractor.py:56-57 `relu(norm1(conv1(x)))` feeding conv2 without a pass of its own.  Only where
   * accflow_conv_in_norm_supported(desc) != 0 (direct kernel, single source of <= 256 channels); NULL: plain input */
  const float* in_norm;
  float acc_scale;               /* internal (correlation GEMM): uniform accumulator multiplier, 0 = none            */
  /* "S16" activations (ACCFLOW_CONV_F16X3 only): a tensor (B, C, H, W) kept PRE-SPLIT in HBM as the fp16 hi / lo terms
   * the matrix-core kernels multiply - storage (B, O = ceil(C/8), 2 terms, H, W) of 16-byte chunks, chunk (b, o, t, y, x)
   * = the 8 halfs {term t of x[b, 8o + j, y, x] * 2^ACCFLOW_F16_ASHIFT, j = 0..7}, hi = fp16(v), lo = fp16(v - hi), channels
   * >= C zero; accflow_s16_item_words(C, H, W) 4-byte words per batch item (every *_bs stride of an S16 tensor counts
   * 4-byte words, like the fp32 strides).  Same bytes as fp32, same values as the split the kernels otherwise perform on
   * the fly (results are bit-identical), but the consumer's loop needs no gather, no conversion and no LDS store: the
   * direct kernel stages the chunks with `buffer_load_dwordx4 ... lds` DMA (out-of-image lanes are out of the
   * descriptor's range and arrive as zeros = the convolution's padding).  The PRODUCER checks the fp16 range (guard).
   *   in_fmt  bit 0: in0 is an S16 tensor of C0 channels, bit 1: in1 of C1 channels (direct-kernel shapes only:
   *           stride 1, "same", Cin >= 16, C0 % 16 == 0 (% 32 for 1x1) when in1 is given; anything else returns 1)
   *   out16   S16 copy of the epilogue's result (STORE / RES_RELU / ACCUM / GRU_Q: of `out`, Cout channels; GRU_ZR: of
   *           `out2` = r*h, Cout/2 channels); with it `out` (GRU_ZR: `out2`) may be NULL.  Channel PAIRS (2k, 2k+1) with
   *           2k >= Cout are not written (an odd last channel's partner is written as zero): the remaining halfs of a
   *           partial last octet belong to the caller (RAFT's motion features: 126 conv channels + the 2 flow channels,
   *           written by accflow_flow_from_coords_s16) and must otherwise have been zeroed by it. */
  int in_fmt;
  void* out16; long long out16_bs;
  /* channel-block scatter (direct kernel, STORE / ACCUM epilogues; 0 = off): output channel ch lives in block ch / cb at
   * channel ch % cb, consecutive blocks out_cbs / e0_cbs / out16_cbs 4-byte words apart in out / e0 / out16 - one GEMM
   * whose output rows are SEVERAL batch items' channels (GMA aggregation of the pairs that share an attention matrix,
   * stacked along the rows) reads its residual from and writes straight into those items' slices.  cb % 32 == 0. */
  int cb;
  long long out_cbs, e0_cbs, out16_cbs;
  /* multi-source S16 form (ACCFLOW_CONV_F16X3): nsrc in 1..4 sources src[0..nsrc) replace in0 / in1 / C0 / C1 / KH / KW /
   * stride / pad (ignored); H = OH and W = OW = the output size; wpatch16 / wscale16 = the pack written by
   * accflow_conv_pack_multi16 for the same (C, KH, KW) list: reduction order source, 16-channel group, tap.  Every
   * epilogue, out16, cb, kws (split-K) and stats work as for the two-source form.  Limits: (8 + KH - 1) * (32 + KW - 1)
   * <= 384 per source (3x3, 1x5, 5x1, 1x7 all fit).  0: the in0 / in1 form. */
  int nsrc;
  accflow_conv_src src[ACCFLOW_CONV_MAX_SRC];
  /* != 0: e0 is an S16 tensor (e0_bs in 4-byte words): the residual operand of ACCFLOW_EPI_RES_RELU read as (hi + lo) /
   * 2^ACCFLOW_F16_ASHIFT - the block input the encoders keep pre-split only (extractor.py:60-63).  Multi-source kernel,
   * ACCFLOW_EPI_RES_RELU + ACCFLOW_ACT_RELU, Cout % 32 == 0 (% 96 in the 96-channel layout), no channel-block scatter;
   * anything else returns 1. */
  int e0_fmt;
  /* ACCFLOW_EPI_TAPGEMM (direct kernel, f16x3, S16 sources, Cout % 128 == 0, act = ACCFLOW_ACT_RELU, no split-K; anything
   * else returns 1): x = relu(conv + bias) is split into fp16 hi / lo in registers - exactly the values an out16 store
   * would hold - and multiplied by the tap matrix of a following KHxKW convolution with tg_rows = KH*KW*Cout2 <= 18 rows (ACCFLOW_TAPGEMM_MAXROWS)
   * (row = tap * Cout2 + channel): tg_w16 / tg_scale = that convolution's 1x1 "all taps at once" pack
   * (accflow_conv_pack_patch16 of the (tg_rows, Cout, 1, 1) matrix, tg_coutpad rows per octet, Cout / 16 steps) with the
   * input channels of every 32-channel block in ACCUMULATOR order - position 16 s + 8 h + e of a block holds channel
   * 8 (2 s + e / 4) + 4 h + e % 4 - so that an MFMA accumulator tile is the next product's B operand without a shuffle.
   * Each 128-channel workgroup writes ITS partial sums, scaled by tg_scale, to
   * tg_out[(channel block) * tg_out_ps + b * tg_out_bs + row * OH*OW + pixel]; accflow_tap_sum_parts_f32 adds the
   * Cout / 128 parts while it sums the taps.  `out` / `out16` are not written. */
  const void* tg_w16; const float* tg_scale;
  float* tg_out; long long tg_out_bs, tg_out_ps;
  int tg_rows, tg_coutpad;
  /* multi-source form only (0 = off): TWO convolutions of the same tensor in one launch.  Output channels [0, split_c0) are
   * the convolution over all nsrc sources with activation `act`; output channels [split_c0, Cout) are a second convolution
   * that reads SOURCE 0 ALONE (the workgroups of those channel blocks stop behind source 0's steps; the pack's rows >=
   * split_c0 of the other sources are never read) and takes ACCFLOW_ACT_NONE.  This is a residual block's first, strided
   * 3x3 convolution together with its 1x1 stride-2 projection of the block input (extractor.py:9,52-53 `downsample`):
   * as parity-class sources (above) the 3x3's class (0, 0) source - its centre tap - reads exactly the pixels (2Y, 2X) the
   * projection reads, so the projection costs one extra 1x1 product per channel block instead of a launch of its own.
   * split_c0 in {64, 96, 128} and Cout - split_c0 a multiple of the chosen channel block (96 -> the 96-channel layout);
   * ACCFLOW_EPI_STORE, act NONE or RELU, no split-K, no channel-block scatter; anything else returns 1. */
  int split_c0;
  /* fp32 tensors of THIS call in the pixel-major layout (B, C/8, H*W, 8) - element (b, c, p) at ((b*C/8 + c/8)*H*W + p)*8 + c%8,
   * batch strides unchanged: the 4 consecutive channels an MFMA accumulator lane holds are 16 contiguous bytes, one load or
   * store instead of four.  bit 0: out; bit 1: pre; bit 2: e1.  Two users (anything else returns 1), both S16-source
   * direct-kernel convolutions on whole 128-channel blocks without split-K: (1) ACCFLOW_EPI_STORE with p32 = 1 - the GRU's
   * context convolutions write their addend this way; (2) the GRU epilogues of the refinement loop with e0_fmt = 1 (the state
   * h read from its pre-split tensor, (hi + lo) / 2^ACCFLOW_F16_ASHIFT): ACCFLOW_EPI_GRU_ZR with p32 = 3 (z out, pre) and
   * out16 = r*h; ACCFLOW_EPI_GRU_Q with p32 = 6 (pre, e1 = z), out = NULL, out16 = the new state - no fp32 state exists
   * inside the loop (1x5 / 5x1 kernels: the tap-specialised instantiations carry this epilogue). */
  int p32;
} accflow_conv_desc;

/* 4-byte words per batch item of an S16 tensor of C channels */
long long accflow_s16_item_words(int C, int H, int W);
/* fp32 (B, C, H*W planes; src_bs = batch stride) -> S16 (dst16_bs in 4-byte words); guard as in accflow_conv_desc */
int accflow_to_s16_f32(const float* src, long long src_bs, void* dst16, long long dst16_bs, int* guard, int B, int C,
                       int HW, void* stream);

/* sizes of the packed buffers for a conv with K = Cin*KH*KW reduction terms */
int accflow_conv_kpad(int Cin, int KH, int KW);
int accflow_conv_coutpad(int Cout);

/* w: (Cout, Cin, KH, KW) as nn.Conv2d.weight.  scale: optional per-output-channel multiplier folded
 * into the packed weights (BatchNorm-eval fold, extractor.py:150-157; ZeroConv2d exp(3*scale),
 * modules.py:94-96; the 0.25 mask factor, update.py:135).  C0 = channels taken from in0 (the rest
 * from in1).  tap_major != 0 orders k as (tap, c) (used by the deformable mode), else (c, tap). */
int accflow_conv_pack_f32(const float* w, const float* scale, int Cout, int Cin, int KH, int KW,
                          int C0, int tap_major, float* wpack, int* ktab, void* stream);

/* same weights split into three bf16 terms for the ACCFLOW_CONV_BF16X3 / X6 modes; wsplit holds
 * 3 * Kpad * CoutPad uint16. */
int accflow_conv_pack_bf16s(const float* w, const float* scale, int Cout, int Cin, int KH, int KW,
                            void* wsplit, void* stream);

/* weights for the LDS-patch kernel: accflow_conv_patch_elems(...) uint16 = [3 terms][ceil(Cin/16)*KH*KW steps]
 * [2 octets][CoutPad][8] bf16 */
long long accflow_conv_patch_elems(int Cout, int Cin, int KH, int KW);
int accflow_conv_pack_patch(const float* w, const float* scale, int Cout, int Cin, int KH, int KW,
                            void* wpatch, void* stream);
/* the same pack as two fp16 terms (same size and layout, third term unused) of w[ch] * scale[ch] * 2^k[ch], k[ch]
 * chosen per output row so that the row's largest magnitude lies in [2^10, 2^11) (k = 0 for an all-zero or non-finite
 * row); wscale16[CoutPad] receives 2^-(k[ch] + ACCFLOW_F16_ASHIFT), the multiplier accflow_conv_desc.wscale16 expects.
 * Finite weights always fit. */
int accflow_conv_pack_patch16(const float* w, const float* scale, int Cout, int Cin, int KH, int KW,
                              void* wpatch16, float* wscale16, void* stream);

/* wsplit's layout as two fp16 terms of the row-scaled weights (see accflow_conv_pack_patch16; identical scales);
 * writes wsplit16 (3 * Kpad * CoutPad uint16, third term unused) and wscale16[CoutPad]. */
int accflow_conv_pack_split16(const float* w, const float* scale, int Cout, int Cin, int KH, int KW,
                              void* wsplit16, float* wscale16, void* stream);

/* Every pack of one convolution in ONE launch: wpack + ktab (accflow_conv_pack_f32) and, for each non-NULL destination, wsplit
 * (accflow_conv_pack_bf16s), wpatch (accflow_conv_pack_patch), wpatch16 + wscale16 (accflow_conv_pack_patch16), wsplit16
 * (accflow_conv_pack_split16) - bit-identical to the single-purpose entry points.  transpose_flip != 0: `w` is the
 * (Cin, Cout, KH, KW) weight of a forward convolution and the packs are those of its input-gradient convolution
 * W[o][c][ky][kx] = w[c][o][KH-1-ky][KW-1-kx] (train.py's backward: no transposed copy).  tap_major packs have no matrix-core
 * forms (returns 1 if one is requested). */
int accflow_conv_pack_all_f32(const float* w, const float* scale, int Cout, int Cin, int KH, int KW, int C0, int tap_major,
                              int transpose_flip, float* wpack, int* ktab, void* wsplit, void* wpatch, void* wpatch16,
                              float* wscale16, void* wsplit16, void* stream);

/* Weight pack of a multi-source convolution (accflow_conv_desc.nsrc): w[s] = (Cout, C[s], KH[s], KW[s]) fp32, the weights
 * that multiply source s (for a plain concatenation: the channel slice of the conv's weight; for the parity classes of a
 * strided conv: its tap subset).  Layout as accflow_conv_pack_patch16 - [2 terms (+1 unused)][steps][2 octets][CoutPad][8]
 * fp16 of w * scale[ch] * 2^k[ch] - with steps = sum_s ceil(C[s] / 16) * KH[s] * KW[s] in the order (source, 16-channel
 * group, tap), k[ch] from the row maximum over ALL sources; wscale16[CoutPad] as there.
 * accflow_conv_multi_pack_elems = uint16 elements of the pack. */
long long accflow_conv_multi_pack_elems(int Cout, int nsrc, const int* C, const int* KH, const int* KW);
int accflow_conv_pack_multi16(const float* const* w, const float* scale, int Cout, int nsrc, const int* C, const int* KH,
                              const int* KW, void* wpatch16, float* wscale16, void* stream);

int accflow_conv2d_f32(const accflow_conv_desc* desc, void* stream);
/* number of statistic slots per (batch item, output channel) the kernel chosen for this descriptor writes (see
 * accflow_conv_desc.stats); 0 if that kernel does not gather statistics */
int accflow_conv_stat_slots(const accflow_conv_desc* desc);
/* != 0 if the kernel chosen for this descriptor can apply accflow_conv_desc.in_norm */
int accflow_conv_in_norm_supported(const accflow_conv_desc* desc);

/* CorrBlock.corr + the 3 avg_pool2d levels (raft/corr.py:8-22, 47-55; gma/corr.py identical).
 * fmap1/fmap2: (B, C, H8, W8).  lvl[l]: (B*H8*W8, Hl, Wl) with Hl = H8 >> l (floor), fp32.
 * lvl0[b,i,j] = <fmap1[b,:,i], fmap2[b,:,j]> / sqrt(C). */
int accflow_corr_volume_f32(const float* fmap1, const float* fmap2, float* lvl0, float* lvl1,
                            float* lvl2, float* lvl3, int B, int C, int H8, int W8, void* stream);

/* Same volume with level 0 computed on the split-bf16 matrix cores (mode = ACCFLOW_CONV_BF16X3 / X6; with
 * ACCFLOW_CONV_F32 identical to the call above).  ws: accflow_corr_volume_ws_bytes(C, H8, W8) bytes of device
 * workspace (the split of one pair's fmap1), reused pair after pair. */
long long accflow_corr_volume_ws_bytes(int C, int H8, int W8);
int accflow_corr_volume_split_f32(const float* fmap1, const float* fmap2, float* lvl0, float* lvl1,
                                  float* lvl2, float* lvl3, void* ws, int mode, int B, int C, int H8,
                                  int W8, void* stream);

/* CorrBlock.__call__ (raft/corr.py:24-45 + bilinear_sampler raft/utils/utils.py:66-80), radius 4,
 * 4 levels: out[b, l*81 + i*9 + j, y, x] = bilinear_zeros(lvl_l[b,y,x], cx/2^l + i-4, cy/2^l + j-4)
 * with (cx, cy) = coords[b, 0:2, y, x].  out: (B, 324, H8, W8) with batch stride out_bs. */
int accflow_corr_lookup_f32(const float* lvl0, const float* lvl1, const float* lvl2,
                            const float* lvl3, const float* coords, float* out, long long out_bs,
                            int B, int H8, int W8, void* stream);

/* Displacement-indexed variants (the layout the estimators use on the hot path; same lookup results).  Level l is
 * E_l[b][p/128][dy][dx][p%128] with p = y1*W8 + x1 the query pixel (blocks of 128, the last one zero-padded),
 * dy = (y' - (y1 >> l)) mod Hl and dx = (x' - (x1 >> l)) mod Wl for target cell (y', x'): a permutation of the
 * reference's corr_pyramid[l] (raft/corr.py:8-22), accflow_corr_disp_level_elems(H8, W8, l) floats per pair.  Query
 * pixels that look at the same displacement - neighbours under a smooth flow - read consecutive addresses.  Requires a
 * split conv mode (level 0 is written by the matrix-core kernel's displaced epilogue); ws as for
 * accflow_corr_volume_split_f32. */
int accflow_corr_disp_supported(int H8, int W8);
long long accflow_corr_disp_level_elems(int H8, int W8, int level);
/* guard: ACCFLOW_CONV_F16X3 only - device int ORed with 1 when a feature value does not fit the fp16 split's range
 * (both feature maps are packed as fp16 hi + lo of x * 2^ACCFLOW_F16_ASHIFT); NULL = no report */
int accflow_corr_volume_disp_f32(const float* fmap1, const float* fmap2, float* lvl0, float* lvl1,
                                 float* lvl2, float* lvl3, void* ws, int mode, int* guard, int B, int C,
                                 int H8, int W8, void* stream);
/* Per-frame form of accflow_corr_volume_disp_f32 (AccFlow evaluates 11 pairs over 7 frames, A13): the feature maps
 * fmaps (F, C, H8, W8) are split ONCE per frame into packs (F x accflow_corr_pack_bytes(C, H8, W8) bytes; C % 16 == 0,
 * W8 even), then pair b correlates queries = frame idx1[b] with targets = frame idx2[b].  idx1 / idx2 are HOST arrays
 * of B ints, read during the call (not retained); every index must lie in [0, F), F = the number of frames `packs`
 * holds - an index outside returns 1 before anything is launched.  Same values as the per-pair call for C = 256. */
long long accflow_corr_pack_bytes(int C, int H8, int W8);
int accflow_corr_pack_f32(const float* fmaps, void* packs, int mode, int* guard, int F, int C, int H8, int W8,
                          void* stream);
int accflow_corr_volume_disp_packed_f32(const void* packs, int F, const int* idx1, const int* idx2, float* lvl0,
                                        float* lvl1, float* lvl2, float* lvl3, int mode, int* guard, int B, int C, int H8,
                                        int W8, void* stream);
int accflow_corr_disp_pool_f32(const float* lvl0, float* lvl1, float* lvl2, float* lvl3, int B, int H8,
                               int W8, void* stream);
int accflow_corr_lookup_disp_f32(const float* lvl0, const float* lvl1, const float* lvl2,
                                 const float* lvl3, const float* coords, float* out, long long out_bs,
                                 int B, int H8, int W8, void* stream);

/* The same lookup writing its result PRE-SPLIT (accflow_conv_desc "S16" format) for the motion encoder's 1x1 convolution
 * (convc1, raft/update.py:83,90): out16 = an S16 tensor of 4 x 88 channels, channel l*88 + j*9 + i = the reference's
 * channel l*81 + i*9 + j (tap i along x, j along y of level l), channels l*88 + 81 .. l*88 + 87 zero - every octet of 8
 * channels belongs to one level, so each lane writes whole 16-byte chunks (the weights of the consuming convolution are
 * permuted accordingly at pack time).  guard: ORed with 1 when a value does not fit the scaled fp16 range (may be NULL). */
int accflow_corr_lookup_disp_s16(const float* lvl0, const float* lvl1, const float* lvl2, const float* lvl3,
                                 const float* coords, void* out16, long long out16_bs, int* guard, int B, int H8,
                                 int W8, void* stream);

/* CorrBlock.__call__ FUSED with BasicMotionEncoder's first convolution (raft/corr.py:24-45 -> raft/update.py:89-90,
 * `cor = relu(convc1(corr))`; gma/update.py identical): the 4 x 81 blended taps never reach HBM - they are the reduction
 * axis of the 1x1 convolution and go from the lanes that computed them through LDS into the matrix cores; only
 * act(convc1(lookup)) is written, pre-split (out16: S16 tensor of Cout channels) and / or fp32 (out, batch stride out_bs;
 * either may be NULL).  Displaced pyramid only.  Cout must be 256 (returns 1 otherwise); act = ACCFLOW_ACT_NONE / _RELU.
 * wpatch16 / wscale16: accflow_conv_pack_patch16 of the (Cout, 336, 1, 1) weight re-indexed to the kernel's reduction
 * order (accflow_corr_lookup_convc1_kpad() = 336 entries): with tap n = j*9 + i (i along x, j along y) of level l being
 * the reference's input channel l*81 + i*9 + j,
 *     k = 32*(n / 8) + 8*l + n % 8   for n < 80,      k = 320 + l   for n = 80,      k = 324 .. 335: zero weights.
 * The taps are blended and split exactly as by accflow_corr_lookup_disp_s16; the products are summed in this order
 * (fp32 accumulation), i.e. results equal lookup -> convolution up to fp32 association.  guard as above. */
int accflow_corr_lookup_convc1_kpad(void);
int accflow_corr_lookup_convc1_s16(const float* lvl0, const float* lvl1, const float* lvl2, const float* lvl3,
                                   const float* coords, const void* wpatch16, const float* wscale16, const float* bias,
                                   void* out16, long long out16_bs, float* out, long long out_bs, int act, int* guard,
                                   int B, int H8, int W8, int Cout, void* stream);

/* RAFT.upsample_flow (raft/raft.py:81-92; gma/gma.py:57-68; AccFlow_.py:27-38):
 * flow (B,2,H8,W8), mask (B,576,H8,W8) -> out (B,2,8*H8,8*W8). */
int accflow_convex_upsample_f32(const float* flow, long long flow_bs, const float* mask,
                                long long mask_bs, float* out, int B, int H8, int W8, void* stream);

/* backwarp (networks/utils.py:96-124): out[n,c,y,x] = bilinear_zeros(img[n,c], x+u, y+v). */
int accflow_backwarp_f32(const float* img, long long img_bs, const float* flow, long long flow_bs,
                         float* out, long long out_bs, int B, int C, int H, int W, void* stream);

/* Composition of two flow fields at the same resolution (warm-start seed, raft.py:123-124 `flow_init`; README.md:11
 * "warmstart"): out[n,:,y,x] = step[n,:,y,x] + bilinear_zeros(acc[n,:], x + step_u, y + step_v), i.e. the flow
 * a -> c given step = a -> b and acc = b -> c.  All (B,2,H,W). */
int accflow_compose_flow_f32(const float* step, long long step_bs, const float* acc, long long acc_bs,
                             float* out, long long out_bs, int B, int H, int W, void* stream);

/* getOcc (AccFlow_.py:127-135).  binary != 0: out (B,1,H,W) = mean_c|i1 - warp(i2,flow)| <= 1 ? 1:0;
 * binary == 0: out (B,C,H,W) = |i1 - warp(i2, flow)|. */
int accflow_get_occ_f32(const float* flow, long long flow_bs, const float* i1, long long i1_bs,
                        const float* i2, long long i2_bs, float* out, long long out_bs, int B, int C,
                        int H, int W, int binary, void* stream);

/* The same maps PRE-SPLIT only (out16: S16 tensor of 1 channel - binary - or C channels; out16_bs in 4-byte words; guard as
 * in accflow_conv_desc): in the fusion chain both maps are read by convolutions alone (AccFlow_.py:98,105,119).  Same values
 * (the binary map's thresholded channel sum keeps its association). */
int accflow_get_occ_s16(const float* flow, long long flow_bs, const float* i1, long long i1_bs, const float* i2,
                        long long i2_bs, void* out16, long long out16_bs, int* guard, int B, int C, int H, int W, int binary,
                        void* stream);

/* downflow8 (AccFlow_.py:138-142): bilinear align_corners resize to (H/8, W/8), then / 8. */
int accflow_downflow8_f32(const float* flow, float* out, int B, int C, int H, int W, void* stream);

/* InstanceNorm2d (no affine, eps, biased var; extractor.py:36-39) over each (b,c) plane of `x`,
 * fused with what follows it in ResidualBlock.forward (extractor.py:56-63):
 *   mode 0: out = norm(x)                    (downsample branch)
 *   mode 1: out = relu(norm(x))
 *   mode 2: out = relu(res + relu(norm(x)))  (block end)                                         */
int accflow_instance_norm_f32(const float* x, const float* res, float* out, int B, int C, int HW,
                              float eps, int mode, void* stream);

/* The same three modes from the statistics a convolution gathered in its epilogue (accflow_conv_desc.stats with `slots`
 * slots per plane): one pass over x instead of three.  meanrstd: workspace of 2*B*C floats (receives mean and
 * 1/sqrt(var + eps) per plane, combined from the partials in a fixed order with the parallel-variance formula in
 * double precision). */
/* only the first half of the call below: meanrstd (B, C, 2) from the statistic partials */
int accflow_instance_stats_finalize_f32(const float* stats, int slots, float* meanrstd, int B, int C, float eps,
                                        void* stream);
int accflow_instance_norm_apply_f32(const float* x, const float* stats, int slots, float* meanrstd, const float* res,
                                    float* out, int B, int C, int HW, float eps, int mode, void* stream);
/* accflow_instance_stats_finalize_f32 for the channels [c0, c0 + C) of a statistics tensor over Ctot channels (the two
 * convolutions of an accflow_conv_desc.split_c0 launch share one): meanrstd (B, C, 2) dense */
int accflow_instance_stats_finalize_sub_f32(const float* stats, int slots, int Ctot, int c0, float* meanrstd, int B, int C,
                                            float eps, void* stream);

/* The same pass with the result ALSO written pre-split (accflow_conv_desc "S16" format: out16, out16_bs in 4-byte words) for
 * the convolutions that read it; `out` (fp32) may then be NULL.  guard as in accflow_conv_desc. */
int accflow_instance_norm_apply_s16_f32(const float* x, const float* stats, int slots, float* meanrstd, const float* res,
                                        float* out, void* out16, long long out16_bs, int* guard, int B, int C, int HW,
                                        float eps, int mode, void* stream);
/* ... and with the residual operand of mode 2 read from an S16 tensor (res16, res16_bs words) as (hi + lo) / 2^4 */
int accflow_instance_norm_apply_s16res_f32(const float* x, const float* stats, int slots, float* meanrstd, const void* res16,
                                           long long res16_bs, float* out, void* out16, long long out16_bs, int* guard, int B,
                                           int C, int HW, float eps, void* stream);

/* The closing pass of a residual block WITH a projection (extractor.py:51-53,59-63): out16 = relu(norm3(res) + relu(norm2(x)))
 * where res is the RAW output of the block's 1x1 stride-2 projection - a channel slice (batch stride res_bs floats) of the
 * tensor its accflow_conv_desc.split_c0 launch wrote, with the statistics channels [res_c0, res_c0 + C) of res_stats (over
 * res_ctot channels, res_slots slots) - normalised here instead of in a pass of its own.  meanrstd: workspace of 4*B*C floats. */
int accflow_instance_norm_apply_s16proj_f32(const float* x, const float* stats, int slots, const float* res, long long res_bs,
                                            const float* res_stats, int res_slots, int res_ctot, int res_c0, float* meanrstd,
                                            void* out16, long long out16_bs, int* guard, int B, int C, int HW, float eps,
                                            void* stream);

/* net = tanh(cnet[:, :hd]), inp = relu(cnet[:, hd:]) (raft.py:116-119) written to two slices. */
int accflow_split_tanh_relu_f32(const float* cnet, float* net, long long net_bs, float* inp,
                                long long inp_bs, int B, int hd, int cd, int HW, void* stream);
/* The same with a gather: output item b is computed from item idx[b] of cnet (n_items items of (hd + cd) x HW floats,
 * contiguous) - pairs that share an image1 share its context features.  idx: HOST array of B ints in [0, n_items),
 * validated before any launch. */
int accflow_split_tanh_relu_idx_f32(const float* cnet, int n_items, const int* idx, float* net, long long net_bs, float* inp,
                                    long long inp_bs, int B, int hd, int cd, int HW, void* stream);

/* coords_grid (raft/utils/utils.py:83-87) [+ flow_init]: coords[b,0]=x, coords[b,1]=y. */
int accflow_coords_grid_f32(float* coords, const float* flow_init, int B, int H8, int W8,
                            void* stream);

/* flow = coords1 - coords0 (raft.py:131) written to up to two destinations (the 2-channel input of
 * convf1 and the tail slice of the GRU input, update.py:97) and, optionally, as the row-shifted stack
 * stack16 (B, 16, H8, W8): channel c*7 + ky = flow[c] shifted by ky - 3 rows with zero fill, channels 14, 15 zero -
 * the 7x7 convolution of the 2-channel flow (convf1, update.py:85,92) equals a 1x7 convolution of this stack with the
 * weights re-indexed [co][c*7 + ky][0][kx] (same products, same sum).  is_flow != 0: `coords1` already holds the flow
 * (nothing is subtracted). */
int accflow_flow_from_coords_f32(const float* coords1, float* dst0, long long dst0_bs, float* dst1,
                                 long long dst1_bs, float* stack16, int is_flow, int B, int H8, int W8, void* stream);
/* S16 form: the fp32 flow into dst0 / dst1 (either may be NULL), the row-shifted stack as an S16 tensor of 16 channels
 * (stack16, stack16_bs) and the two flow channels as channels motion_ch, motion_ch + 1 (motion_ch even) of the S16 tensor
 * motion16 - the halfs the convolution that fills the rest of that octet leaves alone (accflow_conv_desc.out16). */
int accflow_flow_from_coords_s16(const float* coords1, float* dst0, long long dst0_bs, float* dst1, long long dst1_bs,
                                 void* stack16, long long stack16_bs, void* motion16, long long motion16_bs, int motion_ch,
                                 int* guard, int is_flow, int B, int H8, int W8, void* stream);

/* Blending (AccFlow_.py:122-124): out = f1*m + (1-m)*f2, m (B,1,H,W) already sigmoid-ed. */
int accflow_blend_f32(const float* f1, const float* f2, const float* m, float* out, int B, int C,
                      int HW, void* stream);

/* ... with the result pre-split only (out16: S16 tensor of C channels): the fused features feed the flow decoder's
 * convolutions and nothing else (AccFlow_.py:199-200) */
int accflow_blend_s16(const float* f1, const float* f2, const float* m, void* out16, long long out16_bs, int* guard, int B, int C,
                      int HW, void* stream);

/* First half of a deformable convolution as two passes (torchvision.ops.deform_conv2d, modulated, one offset group,
 * stride 1; AccFlow_.py:104): cols = (B, KH*KW*C, H, W) with channel tap*C + c = m_tap * bilinear(x[b,c], y+ky-padH+dy_tap,
 * x+kx-padW+dx_tap).  offset = (B, 2*KH*KW, H, W) with channel 2t = dy, 2t+1 = dx; dmask = (B, KH*KW, H, W).  The second
 * half is accflow_conv2d_f32 with a 1x1 pack of the weights reordered to [o][tap*C + c]. */
int accflow_deform_columns_f32(const float* x, long long x_bs, const float* offset, long long offset_bs,
                               const float* dmask, long long dmask_bs, float* cols, int B, int C, int H, int W,
                               int KH, int KW, int padH, int padW, void* stream);

/* The same columns PRE-SPLIT (cols16: S16 tensor of KH*KW*C channels, C % 8 == 0; cols16_bs in 4-byte words): the 1x1
 * convolution behind them stages them by LDS DMA.  mask_is_logit != 0: dmask holds the modulation's logits and the sigmoid
 * (AccFlow_.py:103) is applied while sampling. */
int accflow_deform_columns_s16(const float* x, long long x_bs, const float* offset, long long offset_bs, const float* dmask,
                               long long dmask_bs, int mask_is_logit, void* cols16, long long cols16_bs, int* guard, int B, int C,
                               int H, int W, int KH, int KW, int padH, int padW, void* stream);

/* Small-Cout "same" convolutions (flow heads update.py:10, blending mask AccFlow_.py:19,118) as a 1x1 matrix-core conv
 * over all taps at once, z = (B, KH*KW*Cout, H, W) with channel tap*Cout + co from weights w[co][c][tap], followed by
 * this shifted sum: out[b,co,y,x] = epi(act(bias[co] + sum_tap z[b, tap*Cout+co, y+ky-padH, x+kx-padW])), zero outside.
 * epi: ACCFLOW_EPI_STORE, _ACCUM (+ e0) or _RES_RELU. */
int accflow_tap_sum_f32(const float* z, const float* bias, const float* e0, long long e0_bs, float* out,
                        long long out_bs, int B, int Cout, int H, int W, int KH, int KW, int padH, int padW,
                        int act, int epi, void* stream);
/* The same sum over z given as `nparts` partial tensors `part_stride` floats apart, each (B, KH*KW*Cout, H, W) with batch
 * stride z_bs floats (what ACCFLOW_EPI_TAPGEMM writes: one part per 128-channel block of the producing convolution);
 * the parts are added in index order, then the taps in the order of accflow_tap_sum_f32. */
int accflow_tap_sum_parts_f32(const float* z, int nparts, long long part_stride, long long z_bs, const float* bias,
                              const float* e0, long long e0_bs, float* out, long long out_bs, int B, int Cout, int H, int W,
                              int KH, int KW, int padH, int padW, int act, int epi, void* stream);

/* in-place activation (ACCFLOW_ACT_*) of a (B, C, HW) channel slice; used for sigmoid(m) on the mask
 * channels of the ZeroConv2d output (AccFlow_.py:102-103). */
int accflow_activation_f32(float* x, long long x_bs, int B, int C, int HW, int act, void* stream);

/* strided copy of a (B, C, HW) tensor (used to place tensors into concat slices). */
int accflow_copy_f32(const float* src, long long src_bs, float* dst, long long dst_bs, int B, int C,
                     int HW, void* stream);

/* GMA (gma/modules.py:54-76, 102-115), heads = 1.
 * attention: qk (B, 2*D, P) from to_qk; attn (B, P, P) = softmax_j(scale * <q_i, k_j>).
 * aggregate: out (B, D, P) = fmap + gamma[0] * (attn @ v^T)^T with v (B, D, P). */
int accflow_gma_attention_f32(const float* qk, float* attn, int B, int D, int P, float scale,
                              void* stream);
int accflow_gma_aggregate_f32(const float* attn, const float* v, const float* fmap,
                              const float* gamma, float* out, long long out_bs, int B, int D, int P,
                              void* stream);

/* Hot-path variants on the TRANSPOSED attention attnT[b][j][i] (j-major; a (1, P, h, w) activation tensor), with the
 * aggregation running as per-pair 1x1 convolutions on the matrix cores (mode = ACCFLOW_CONV_F16X3 - v * gamma as fp16
 * hi + lo with per-row scales, guard as for accflow_conv_desc - or BF16X3/X6).
 * ws: accflow_gma_aggregate_ws_bytes(B, D, P) bytes.  Pairs out of the same image1 share their attention: a caller
 * stacks their v / fmap / out rows (D = n * 128) and passes B = 1 - one GEMM, the attention read once.
 * accflow_gma_attention_t_f32: P = H * W; in the split modes (mode != ACCFLOW_CONV_F32, ws = accflow_gma_attention_ws_bytes
 * bytes, scale = D^-1/2) the logits come from the correlation volume's matrix-core GEMM in its fp32-equivalent bf16x6
 * form, else (or ws NULL) from the fp32 MFMA GEMM; then a two-sweep column softmax. */
long long accflow_gma_attention_ws_bytes(int D, int H, int W);
/* Hot path in ACCFLOW_CONV_F16X3 (S16 format, see accflow_conv_desc): the attention of B feature maps, softmax over j,
 * stored j-major and PRE-SPLIT as an S16 tensor of P = H*W "channels" j over the H x W pixels i per item
 * (accflow_s16_item_words(P, H, W) words apart) - built once per image1, read by every aggregation of the 12 refinement
 * iterations without any conversion.  Two passes of one register-only GEMM over the fp16 hi/lo packs of q and k (column
 * statistics, then the normalised exponentials): no logits matrix exists in memory.  D % 16 == 0; ws:
 * accflow_gma_attention_s16_ws_bytes(D, H, W) bytes; guard as in accflow_conv_desc (q / k outside the fp16 range). */
long long accflow_gma_attention_s16_ws_bytes(int D, int H, int W);
int accflow_gma_attention_s16(const float* qk, void* attn16, void* ws, int* guard, int B, int D, int H, int W, float scale,
                              void* stream);
/* out_k = fmap_k + gamma * (attn @ v_k^T) for the n items k that share ONE S16 attention (gma/modules.py:102-115; the pairs
 * (i, i-1) and (i, 0) of AccFlow's schedule have the same image1): v = (n, D, P) contiguous; item k's residual at
 * fmap + k*fmap_bs, its fp32 result at out + k*out_bs (out may be NULL), its S16 result at out16 + k*out16_bs words (may be
 * NULL); D % 32 == 0; ws: accflow_gma_aggregate_s16_ws_bytes(n, D, P) bytes. */
long long accflow_gma_aggregate_s16_ws_bytes(int n, int D, int P);
int accflow_gma_aggregate_s16(const void* attn16, const float* v, const float* fmap, long long fmap_bs, const float* gamma,
                              float* out, long long out_bs, void* out16, long long out16_bs, void* ws, int* guard, int n, int D,
                              int H, int W, void* stream);
int accflow_gma_attention_t_f32(const float* qk, float* attnT, void* ws, int mode, int B, int D, int H, int W,
                                float scale, void* stream);
long long accflow_gma_aggregate_ws_bytes(int B, int D, int P);
int accflow_gma_aggregate_t_f32(const float* attnT, const float* v, const float* fmap, const float* gamma,
                                float* out, long long out_bs, void* ws, int mode, int* guard, int B, int D, int H, int W,
                                void* stream);

/* ---- Backward of the fusion heads (the part of train_acc.py:113-312 that carries gradients - the estimator is
 * frozen, train_acc.py:164, and AccFlow_.py:172,182,195,198 detach flows / occlusion / error maps).  fp32 arithmetic.  The
 * input gradient of a stride-1 convolution is accflow_conv2d_f32 with the transposed, flipped weights. ---- */
/* dx = dy * act'(y) from the activation's OUTPUT y; (B, CHW) planes with batch strides (channel slices of wider tensors) */
int accflow_act_backward_f32(const float* dy, long long dy_bs, const float* y, long long y_bs, float* dx, long long dx_bs, int B,
                             long long CHW, int act, void* stream);
/* dst += src over (B, CHW) planes with batch strides: gradient accumulation where a tensor feeds several consumers */
int accflow_add_f32(float* dst, long long dst_bs, const float* src, long long src_bs, int B, long long CHW, void* stream);
/* dpred = scale * sign(pred - gt): the gradient of loss.py:34-36's  mean |pred - gt|  with scale = 1 / n */
int accflow_l1_grad_f32(const float* pred, const float* gt, float* dpred, long long n, float scale, void* stream);
/* zero insertion dst (B, C, Hd, Wd) [y s, x s] = src (B, C, OH, OW) [y, x], zeros elsewhere: lays the output gradient of a
 * stride-s convolution out so that its input gradient is a stride-1 convolution with the transposed, flipped weights */
int accflow_dilate_f32(const float* src, long long src_bs, float* dst, int B, int C, int OH, int OW, int Hd, int Wd, int stride,
                       void* stream);
/* Blending (AccFlow_.py:122-124) out = f1 m + (1 - m) f2:  df1, df2 (B, C, HW), dm (B, 1, HW); all contiguous */
int accflow_blend_backward_f32(const float* dy, const float* f1, const float* f2, const float* m, float* df1, float* df2,
                               float* dm, int B, int C, int HW, void* stream);
/* Convex upsampling (raft.py:81-92): dup (B, 2, 8 H8, 8 W8) -> dflow (B, 2, H8, W8; zeroed here, float atomics) and dmask
 * (B, 576, H8, W8); flow / mask = the forward inputs; all contiguous */
int accflow_convex_upsample_backward_f32(const float* dup, const float* flow, const float* mask, float* dflow, float* dmask,
                                         int B, int H8, int W8, void* stream);
/* Weight / bias gradient of a convolution: dw (Cout, Cin, KH, KW) = sum_{b,y,x} dy[b,co,y,x] x[b,ci,y s+ky-padH,x s+kx-padW],
 * db (Cout; may be NULL) = sum dy.  x (B, Cin, H, W) / dy (B, Cout, OH, OW) with batch strides; dw / db are zeroed here (float
 * atomics over the pixel-axis parts: the summation order is not fixed) */
int accflow_conv_wgrad_f32(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw, float* db, int B, int Cin,
                           int Cout, int H, int W, int KH, int KW, int stride, int padH, int padW, void* stream);
/* Modulated deformable convolution (torchvision.ops.deform_conv2d, AccFlow_.py:104), backward from the gradient of its deformed
 * columns dcols (B, KH*KW*C, H, W) = W^T dY: dx (B, C, H, W; zeroed here, float atomics), doffset (B, 2*KH*KW, H, W; channel 2t =
 * d/d(dy_t), 2t+1 = d/d(dx_t)), ddmask (B, KH*KW, H, W).  x / offset / dmask = the forward inputs. */
int accflow_deform_conv_backward_f32(const float* x, long long x_bs, const float* offset, long long offset_bs, const float* dmask,
                                     long long dmask_bs, const float* dcols, float* dx, long long dx_bs, float* doffset,
                                     float* ddmask, int B, int C, int H, int W, int KH, int KW, int padH, int padW, void* stream);

int accflow_abi_version(void);
/* sizeof(accflow_conv_desc) / sizeof(accflow_conv_src) as this library was compiled: a binding compares them with its own mirror
 * of the structures at load time (accflow_amd/_lib.py does), so that a field added on one side only fails loudly. */
int accflow_conv_desc_bytes(void);
int accflow_conv_src_bytes(void);

#ifdef __cplusplus
}
#endif
#endif /* ACCFLOW_HIP_H */
