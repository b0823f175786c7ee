// Does cross-stream event traffic make a created launch stream slow?  A main stream S runs a chain of dependent kernels over an
// L2-resident buffer; every `every`-th kernel an event is recorded on S, a side stream T waits for it and runs a kernel of its own,
// and S waits for T's event `lag` kernels later (the engine's weight-gradient / lane pattern).  S = legacy default stream or a created
// one; event flags = engine's (DisableTiming | DisableSystemFence), DisableTiming only, or default.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/seb stream_event_bench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void touch(float4* p, long long n, float a) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float4 v = p[i];
    v.x = v.x * a + 1.f;
    p[i] = v;
  }
}

static double run(hipStream_t S, hipStream_t T, unsigned flags, bool cross, float4* a, float4* b, long long n, int every) {
  const int K = 600;
  std::vector<hipEvent_t> ev(2 * (K / every + 2));
  for (auto& e : ev) (void)hipEventCreateWithFlags(&e, flags);
  auto body = [&]() {
    int k = 0;
    for (int i = 0; i < K; ++i) {
      hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, S, a, n, 1.0001f);
      if (cross && i % every == 0) {
        (void)hipEventRecord(ev[k], S);
        (void)hipStreamWaitEvent(T, ev[k], 0);
        hipLaunchKernelGGL(touch, dim3(256), dim3(256), 0, T, b, n / 4, 1.0001f);
        (void)hipEventRecord(ev[k + 1], T);
        k += 2;
      }
    }
    if (cross) (void)hipStreamWaitEvent(S, ev[k - 1], 0);
  };
  body();
  (void)hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < 5; ++r) body();
  (void)hipDeviceSynchronize();
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (5.0 * K);
  for (auto& e : ev) (void)hipEventDestroy(e);
  return us;
}

int main() {
  const long long n = (8LL << 20) / 16;  // 8 MiB: stays in L2 / MALL between the kernels of the chain
  float4 *a, *b;
  (void)hipMalloc(&a, n * 16);
  (void)hipMalloc(&b, n * 16);
  (void)hipMemset(a, 0, n * 16);
  (void)hipMemset(b, 0, n * 16);
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStream_t created, side;
  (void)hipStreamCreateWithFlags(&created, hipStreamNonBlocking);
  (void)hipStreamCreateWithPriority(&side, hipStreamNonBlocking, lo);
  struct { const char* name; unsigned f; } flags[] = {{"DisableTiming|DisableSystemFence", hipEventDisableTiming | hipEventDisableSystemFence},
                                                      {"DisableTiming", hipEventDisableTiming}, {"default", hipEventDefault}};
  printf("us per main-stream kernel (8 MiB read+write each), 600-kernel chain\n");
  for (int main_created = 0; main_created < 2; ++main_created) {
    hipStream_t S = main_created ? created : nullptr;
    printf("main stream %-8s  no side traffic: %6.2f\n", main_created ? "created" : "default", run(S, side, flags[0].f, false, a, b, n, 5));
    for (auto& f : flags)
      for (int every : {5, 1})
        printf("main stream %-8s  side kernel every %d, events %-34s: %6.2f\n", main_created ? "created" : "default", every, f.name,
               run(S, side, f.f, true, a, b, n, every));
  }
  return 0;
}
