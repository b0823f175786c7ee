// How many hardware queues can a process keep at work before a dependent kernel chain slows down?  (the engine's
// step went from 6.5 to 16 ms with a fifth stream at work.)  Stream 0 (the legacy default stream) runs a chain of 400 dependent kernels
// over an L2-resident buffer; n-1 further created streams (priorities cycling low / high / normal) each run a slower trickle of small
// kernels, loosely coupled to the chain by events like the engine's side streams.  Reported: us per chain kernel for n = 1 .. 8 streams.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/qcb queue_count_bench.hip
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

int main() {
  const long long n = (8LL << 20) / 16;
  const int MAXS = 8, K = 400;
  float4* buf[MAXS];
  for (int i = 0; i < MAXS; ++i) {
    (void)hipMalloc(&buf[i], n * 16);
    (void)hipMemset(buf[i], 0, n * 16);
  }
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStream_t st[MAXS];
  st[0] = nullptr;
  for (int i = 1; i < MAXS; ++i) {
    const int pr = (i % 3 == 1) ? lo : (i % 3 == 2) ? hi : 0;
    (void)hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, pr);
  }
  std::vector<hipEvent_t> ev(K);
  for (auto& e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence);
  printf("streams at work   us per kernel of the main chain (8 MiB read+write each)\n");
  for (int ns = 1; ns <= MAXS; ++ns) {
    auto body = [&]() {
      for (int i = 0; i < K; ++i) {
        hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, st[0], buf[0], n, 1.0001f);
        if (ns > 1 && i % 8 == 0) {  // every 8th kernel: one side stream (round robin) picks up work that depends on the chain so far
          const int s = 1 + (i / 8) % (ns - 1);
          (void)hipEventRecord(ev[i], st[0]);
          (void)hipStreamWaitEvent(st[s], ev[i], 0);
          hipLaunchKernelGGL(touch, dim3(128), dim3(256), 0, st[s], buf[s], n / 2, 1.0001f);
          hipLaunchKernelGGL(touch, dim3(128), dim3(256), 0, st[s], buf[s], n / 2, 1.0001f);
        }
      }
    };
    body();
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 5; ++r) body();
    (void)hipDeviceSynchronize();
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (5.0 * K);
    printf("%8d          %7.2f\n", ns, us);
  }
  return 0;
}
