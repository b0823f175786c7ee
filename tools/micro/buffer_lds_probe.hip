// Probe of `buffer_load_dwordx4 ... lds` on gfx950: (1) out-of-range lanes (voffset = ~0) deliver zeros into LDS, (2) the scalar offset is
// added to the address but takes no part in the range check, (3) the LDS destination (M0) may lie above 64 KiB.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/buffer_lds_probe.hip -o /tmp/buffer_lds_probe && /tmp/buffer_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void probe(const float* in, float* out, int nbytes, int soff, int lds_off) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* dst = reinterpret_cast<float*>(smem + lds_off);
  for (int i = threadIdx.x; i < 256; i += 64) dst[i] = -1.f;  // stale marker
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, (short)0, nbytes, 0x00020000);
  unsigned vo = threadIdx.x * 16;
  if (threadIdx.x % 3 == 1) vo = 0xffffffffu;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(smem + lds_off), 16, vo, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = dst[i];
}
int main() {
  const int n = 4096;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, n * 4);
  hipMalloc(&o, 256 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int bad = 0;
  for (int lds_off : {0, 100 * 1024, 150 * 1024}) {
    for (int soff : {0, 2048}) {
      // range: the buffer holds 2048 + 512 bytes -> with soff = 2048 lanes >= 32 exceed it only through the scalar offset
      probe<<<1, 64, 160 * 1024>>>(d, o, 2048 + 512, soff, lds_off);
      std::vector<float> r(256);
      hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
      int wrong = 0, beyond_nonzero = 0;
      for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 4; ++k) {
          const float got = r[l * 4 + k];
          const bool oob_lane = l % 3 == 1;
          const long byte = (long)l * 16 + soff + k * 4;
          float want = oob_lane ? 0.f : (float)(byte / 4);
          if ((long)l * 16 >= 2048 + 512) want = 0.f;
          if (!oob_lane && byte >= 2048 + 512) {  // only the scalar offset carries it past the end: report what the hardware does
            if (got != 0.f) ++beyond_nonzero;
            continue;
          }
          if (got != want) ++wrong;
        }
      printf("lds_off %6d soff %4d: wrong %d, lanes past the end through soffset that still read data: %d words\n", lds_off, soff, wrong, beyond_nonzero);
      bad += wrong;
    }
  }
  printf(bad ? "FAILED\n" : "OK\n");
  return bad ? 1 : 0;
}
