// Launch-to-launch cost of dependent kernels on the legacy default stream against created streams (gfx950, ROCm 7): why the engine's
// launch stream must be the default one.  hipcc --offload-arch=gfx950 -O2 -o /tmp/slb stream_launch_bench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void tiny(float* p) { p[blockIdx.x * blockDim.x + threadIdx.x] += 1.f; }
__global__ void biglds(float* p) {
  extern __shared__ float s[];
  s[threadIdx.x] = p[blockIdx.x * blockDim.x + threadIdx.x];
  __syncthreads();
  p[blockIdx.x * blockDim.x + threadIdx.x] = s[(threadIdx.x + 1) & 255] + 1.f;
}

__global__ void stream_copy(const float4* a, float4* b, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) b[i] = a[i];
}
// one 512 MiB copy (read + write = 1 GiB of traffic) per launch: does a stream get the whole chip?
static double run_copy(hipStream_t st, const float4* a, float4* b, long long n) {
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(stream_copy, dim3(256 * 8), dim3(256), 0, st, a, b, n);
  (void)hipStreamSynchronize(st);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(stream_copy, dim3(256 * 8), dim3(256), 0, st, a, b, n);
  (void)hipStreamSynchronize(st);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20;
}

static double run(hipStream_t st, bool big, float* d, int n) {
  for (int i = 0; i < 50; ++i) {
    if (big) hipLaunchKernelGGL(biglds, dim3(256), dim3(256), 128 * 1024, st, d);
    else hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, st, d);
  }
  (void)hipStreamSynchronize(st);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) {
    if (big) hipLaunchKernelGGL(biglds, dim3(256), dim3(256), 128 * 1024, st, d);
    else hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, st, d);
  }
  (void)hipStreamSynchronize(st);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 256 * 4);
  (void)hipMemset(d, 0, 256 * 256 * 4);
  (void)hipFuncSetAttribute((const void*)biglds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStream_t nb, bl, ph, pl;
  (void)hipStreamCreateWithFlags(&nb, hipStreamNonBlocking);
  (void)hipStreamCreateWithFlags(&bl, hipStreamDefault);
  (void)hipStreamCreateWithPriority(&ph, hipStreamNonBlocking, hi);
  (void)hipStreamCreateWithPriority(&pl, hipStreamNonBlocking, lo);
  struct { const char* name; hipStream_t s; } cases[] = {{"default (null)", nullptr}, {"created non-blocking", nb}, {"created blocking", bl},
                                                         {"created high priority", ph}, {"created low priority", pl}, {"default (null) again", nullptr}};
  const long long n = (512LL << 20) / 16;
  float4 *a, *b;
  (void)hipMalloc(&a, n * 16);
  (void)hipMalloc(&b, n * 16);
  (void)hipMemset(a, 0, n * 16);
  for (auto& c : cases) {
    const double t = run(c.s, false, d, 2000), tb = run(c.s, true, d, 2000), tc = run_copy(c.s, a, b, n);
    printf("%-24s  tiny kernel %6.2f us/launch   128 KiB-LDS kernel %6.2f us/launch   512 MiB copy %7.1f us = %5.2f TB/s\n", c.name, t, tb, tc,
           2.0 * n * 16 / tc * 1e-6);
  }
  return 0;
}
