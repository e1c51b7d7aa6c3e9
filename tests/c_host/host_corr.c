/* A plain C host of libaccflow_hip.so: no Python, no torch, no device code of its own - what a maintainer who binds the
 * C-ABI from another language gets.  It runs the CorrBlock path (raft/corr.py:8-55) both ways the header offers:
 *   reference layout  accflow_corr_volume_f32 + accflow_corr_lookup_f32
 *   hot-path layout   accflow_corr_volume_disp_f32 (split-bf16 matrix cores) + accflow_corr_lookup_disp_f32
 * usage: host_corr IN.bin OUT.bin
 *   IN.bin : int32 B, C, H8, W8; float32 fmap1[B*C*P], fmap2[B*C*P], coords[B*2*P]          (P = H8*W8)
 *   OUT.bin: float32 out_ref[B*324*P], out_disp[B*324*P]
 * Built and run by tests/test_c_host.py (the expected values come from the oracle there). */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "accflow_hip.h"

#define CHECK(x) do { int rc_ = (int)(x); if (rc_ != 0) { fprintf(stderr, "%s failed: %d (line %d)\n", #x, rc_, __LINE__); return 2; } } while (0)

static float* dev_alloc(long long n) {
  void* p = NULL;
  if (hipMalloc(&p, (size_t)n * sizeof(float)) != hipSuccess) return NULL;
  return (float*)p;
}

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s IN.bin OUT.bin\n", argv[0]); return 1; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  int hdr[4];
  if (fread(hdr, sizeof(int), 4, f) != 4) return 1;
  const int B = hdr[0], C = hdr[1], H8 = hdr[2], W8 = hdr[3], P = H8 * W8;
  const long long nf = (long long)B * C * P, nc = (long long)B * 2 * P, no = (long long)B * 324 * P;
  float* h_in = (float*)malloc((size_t)(2 * nf + nc) * sizeof(float));
  if (fread(h_in, sizeof(float), (size_t)(2 * nf + nc), f) != (size_t)(2 * nf + nc)) return 1;
  fclose(f);
  if (accflow_abi_version() != ACCFLOW_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 3; }

  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  float *f1 = dev_alloc(nf), *f2 = dev_alloc(nf), *co = dev_alloc(nc), *o1 = dev_alloc(no), *o2 = dev_alloc(no);
  if (!f1 || !f2 || !co || !o1 || !o2) return 2;
  CHECK(hipMemcpyAsync(f1, h_in, (size_t)nf * 4, hipMemcpyHostToDevice, st));
  CHECK(hipMemcpyAsync(f2, h_in + nf, (size_t)nf * 4, hipMemcpyHostToDevice, st));
  CHECK(hipMemcpyAsync(co, h_in + 2 * nf, (size_t)nc * 4, hipMemcpyHostToDevice, st));

  /* reference layout: lvl[l] = (B*P, H8 >> l, W8 >> l) */
  float* lv[4];
  for (int l = 0; l < 4; ++l) {
    lv[l] = dev_alloc((long long)B * P * (H8 >> l) * (W8 >> l));
    if (!lv[l]) return 2;
  }
  CHECK(accflow_corr_volume_f32(f1, f2, lv[0], lv[1], lv[2], lv[3], B, C, H8, W8, st));
  CHECK(accflow_corr_lookup_f32(lv[0], lv[1], lv[2], lv[3], co, o1, 324LL * P, B, H8, W8, st));

  /* hot-path layout (when offered for this size): displaced pyramid, level 0 from the matrix-core GEMM */
  int have_disp = accflow_corr_disp_supported(H8, W8);
  if (have_disp) {
    float* dl[4];
    for (int l = 0; l < 4; ++l) {
      dl[l] = dev_alloc((long long)B * accflow_corr_disp_level_elems(H8, W8, l));
      if (!dl[l]) return 2;
    }
    void* ws = NULL;
    CHECK(hipMalloc(&ws, (size_t)accflow_corr_volume_ws_bytes(C, H8, W8)));
    CHECK(accflow_corr_volume_disp_f32(f1, f2, dl[0], dl[1], dl[2], dl[3], ws, ACCFLOW_CONV_BF16X6, NULL, B, C, H8, W8, st));
    CHECK(accflow_corr_lookup_disp_f32(dl[0], dl[1], dl[2], dl[3], co, o2, 324LL * P, B, H8, W8, st));
  } else {
    CHECK(hipMemcpyAsync(o2, o1, (size_t)no * 4, hipMemcpyDeviceToDevice, st));
  }
  float* h_out = (float*)malloc((size_t)(2 * no) * sizeof(float));
  CHECK(hipMemcpyAsync(h_out, o1, (size_t)no * 4, hipMemcpyDeviceToHost, st));
  CHECK(hipMemcpyAsync(h_out + no, o2, (size_t)no * 4, hipMemcpyDeviceToHost, st));
  CHECK(hipStreamSynchronize(st));
  f = fopen(argv[2], "wb");
  if (!f) { perror(argv[2]); return 1; }
  fwrite(h_out, sizeof(float), (size_t)(2 * no), f);
  fclose(f);
  printf("host_corr: B=%d C=%d %dx%d, displaced layout %s\n", B, C, H8, W8, have_disp ? "yes" : "no");
  return 0;
}
