// Probe (run on the GPU box): what does `buffer_load_dwordx4 ... offen lds` write into LDS for lanes whose offset lies
// outside the buffer descriptor's range?  The S16 patch loader relies on zeros (zero padding of the convolution).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* src, unsigned* out, int nbytes) {
  __shared__ u32x4 L[128];
  const int lane = threadIdx.x;
  L[lane] = u32x4{0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu};
  L[64 + lane] = u32x4{0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu};
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, nbytes, 0x00020000);
  // lanes 0..39 in range, 40..47 explicitly masked (0xFFFFFFFF), 48..63 just beyond the range
  unsigned voff = lane < 40 ? lane * 16 : (lane < 48 ? 0xFFFFFFFFu : (unsigned)nbytes + (lane - 48) * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)&L[0], 16, voff, 0, 0, 0);
  // second piece with a scalar offset, into the upper half
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)&L[64], 16, voff, 64, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = 0; i < 2; ++i) {
    u32x4 v = L[i * 64 + lane];
    for (int j = 0; j < 4; ++j) out[(i * 64 + lane) * 4 + j] = v[j];
  }
}
int main() {
  const int n = 40 * 16 + 64;   // bytes in range: lanes 0..39 of piece 0 (+64 for the soffset piece)
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
  unsigned *d, *o;
  hipMalloc(&d, 4096); hipMalloc(&o, 128 * 16);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
  std::vector<unsigned> r(512);
  hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
  int bad_in = 0, nonzero_oob = 0, stale = 0;
  for (int p = 0; p < 2; ++p)
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const unsigned v = r[(p * 64 + l) * 4 + j];
        const long off = (l < 40 ? l * 16 : -1);
        const bool inr = off >= 0 && off + p * 64 + 16 <= n;
        if (inr) { if (v != 0x1000 + (off + p * 64) / 4 + j) ++bad_in; }
        else { if (v == 0xDEADBEEFu) ++stale; else if (v != 0) ++nonzero_oob; }
      }
  printf("lds-dma probe: in-range mismatches %d, out-of-range dwords left stale %d, out-of-range dwords non-zero %d\n", bad_in, stale, nonzero_oob);
  printf("piece0 lane 39..49 dword0:"); for (int l = 39; l < 50; ++l) printf(" %x", r[l * 4]); printf("\n");
  return (bad_in || stale || nonzero_oob) ? 1 : 0;
}
