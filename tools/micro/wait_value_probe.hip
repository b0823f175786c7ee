// Probe: can a kernel on stream A release work on stream B through hipStreamWaitValue32 on signal memory (no event record on A)?
// And what does a dependent kernel boundary on A cost with (a) nothing, (b) hipEventRecord + hipStreamWaitEvent(B), (c) a flag store
// inside the kernel + hipStreamWaitValue32(B)?      hipcc --offload-arch=gfx950 -O2 -o /tmp/wvp tools/micro/wait_value_probe.hip && /tmp/wvp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void work(float* p, int n, unsigned* flag, unsigned val, unsigned* done_ctr) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = p[i % n];
  for (int k = 0; k < 6000; ++k) v = v * 1.0001f + 0.5f;
  p[i % n] = v;
  if (flag) {  // the last block to finish publishes the value
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      unsigned t = atomicAdd(done_ctr, 1u);
      if (t == gridDim.x - 1) {
        *done_ctr = 0;
        __hip_atomic_store(flag, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}
__global__ void consumer(float* p, int n, const unsigned* flag, unsigned expect, int* bad) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && flag && __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < expect) atomicAdd(bad, 1);
  p[i % n] += 1.0f;
}

int main() {
  int dev = 0, can = 0;
  CK(hipSetDevice(dev));
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t A, B;
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&A, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&B, hipStreamNonBlocking, lo));
  const int n = 1 << 20;
  float *pa, *pb;
  CK(hipMalloc(&pa, n * 4));
  CK(hipMalloc(&pb, n * 4));
  unsigned* flag = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory);
  printf("hipExtMallocWithFlags(hipMallocSignalMemory) -> %s\n", hipGetErrorString(e));
  if (e != hipSuccess) return 1;
  CK(hipMemset(flag, 0, 8));
  unsigned* ctr;
  int* bad;
  CK(hipMalloc(&ctr, 4));
  CK(hipMemset(ctr, 0, 4));
  CK(hipMalloc(&bad, 4));
  CK(hipMemset(bad, 0, 4));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  CK(hipDeviceSynchronize());
  const int ITER = 300, BLK = 256;
  double host_us = 0;
  auto run = [&](int mode) -> double {
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, A);
    auto t0 = std::chrono::steady_clock::now();
    static unsigned seq = 0;
    for (int it = 0; it < ITER; ++it) {
      ++seq;
      if (mode == 2) hipLaunchKernelGGL(work, dim3(256), dim3(BLK), 0, A, pa, n, flag, seq, ctr);
      else hipLaunchKernelGGL(work, dim3(256), dim3(BLK), 0, A, pa, n, (unsigned*)nullptr, 0u, (unsigned*)nullptr);
      if (mode == 6 || mode == 7) {    // a stream-ordered value write on A, a polling wait on B (7: and a consumer behind it)
        (void)hipStreamWriteValue32(A, flag, seq, 0);
        (void)hipStreamWaitValue32(B, flag, seq, hipStreamWaitValueGte, 0xffffffffu);
        if (mode == 7) hipLaunchKernelGGL(consumer, dim3(64), dim3(BLK), 0, B, pb, n, (const unsigned*)flag, seq, bad);
      } else if (mode == 3) {
        (void)hipEventRecord(ev, A);   // the marker alone: nobody waits for it
      } else if (mode == 4) {
        (void)hipEventRecord(ev, A);   // marker + a waiting stream with nothing to run
        (void)hipStreamWaitEvent(B, ev, 0);
      } else if (mode == 5) {          // no marker: an independent consumer on B (what the consumer's own CU-time costs A)
        hipLaunchKernelGGL(consumer, dim3(64), dim3(BLK), 0, B, pb, n, (const unsigned*)nullptr, 0u, bad);
      } else if (mode == 1) {
        (void)hipEventRecord(ev, A);
        (void)hipStreamWaitEvent(B, ev, 0);
        hipLaunchKernelGGL(consumer, dim3(64), dim3(BLK), 0, B, pb, n, (const unsigned*)nullptr, 0u, bad);
      } else if (mode == 2) {
        (void)hipStreamWaitValue32(B, flag, seq, hipStreamWaitValueGte, 0xffffffffu);
        hipLaunchKernelGGL(consumer, dim3(64), dim3(BLK), 0, B, pb, n, (const unsigned*)flag, seq, bad);
      }
      hipLaunchKernelGGL(work, dim3(256), dim3(BLK), 0, A, pa, n, (unsigned*)nullptr, 0u, (unsigned*)nullptr);
    }
    (void)hipEventRecord(e1, A);
    auto t1 = std::chrono::steady_clock::now();
    (void)hipStreamSynchronize(A);
    (void)hipStreamSynchronize(B);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    host_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / ITER;
    return ms * 1e3 / ITER;
  };
  for (int rep = 0; rep < 2; ++rep) {
    const double a = run(0); const double ha = host_us; const double b = run(1); const double hb2 = host_us; const double c = run(2); const double hc = host_us;
    printf("host enqueue per iteration: %.1f / %.1f / %.1f us\n", ha, hb2, hc);
    int hb = 0;
    CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("per iteration (2 kernels on A): plain %.2f us | + event record/wait for B %.2f us | + in-kernel flag, wait-value on B %.2f us | consumer saw a stale flag %d times\n", a, b, c, hb);
    const double d3 = run(3), d4 = run(4), d5 = run(5), d6 = run(6), d7 = run(7);
    printf("                                hipStreamWriteValue32 on A + hipStreamWaitValue32 on B %.2f us | ... + consumer %.2f us\n", d6, d7);
    printf("                                event record alone %.2f us | record + waiting (empty) stream %.2f us | independent consumer on B, no marker %.2f us\n", d3, d4, d5);
  }
  return 0;
}
