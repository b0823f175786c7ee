// Grid-barrier cost on MI355X: variants of arrival / polling, empty kernels around them.
//   hipcc --offload-arch=gfx950 -O3 -o build/grid_barrier_bench tools/micro/grid_barrier_bench.hip && build/grid_barrier_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int SLEEP, bool FENCE>
__device__ __forceinline__ void barrier_flat(unsigned* bar, unsigned nblocks) {
  if (FENCE) __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned prev = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == nblocks - 1) {
      __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&bar[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned spins = 0;
      while (__hip_atomic_load(&bar[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) {
        __builtin_amdgcn_s_sleep(SLEEP);
        if (++spins > (1u << 22)) { bar[2] = 1; break; }
      }
    }
  }
  __syncthreads();
  if (FENCE) __threadfence();
}

// two levels: groups of 32 workgroups share a counter (64-byte apart); the last of a group arrives at the root
template <int SLEEP>
__device__ __forceinline__ void barrier_tree(unsigned* bar, unsigned nblocks) {
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned grp = blockIdx.x >> 5, ngrp = (nblocks + 31) >> 5;
    const unsigned gsize = min(32u, nblocks - grp * 32);
    unsigned* gc = bar + 16 + grp * 16;
    bool release = false;
    const unsigned p = __hip_atomic_fetch_add(gc, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (p == gsize - 1) {
      __hip_atomic_store(gc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned prev = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      release = prev == ngrp - 1;
    }
    if (release) {
      __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&bar[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned spins = 0;
      while (__hip_atomic_load(&bar[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) {
        __builtin_amdgcn_s_sleep(SLEEP);
        if (++spins > (1u << 22)) { bar[2] = 1; break; }
      }
    }
  }
  __syncthreads();
  __threadfence();
}

// no cache maintenance at all: relaxed device-scope atomics only (performed at the device's coherence point); the workgroup
// barrier orders the block's earlier atomics (s_waitcnt vmcnt(0)) before thread 0's arrival
template <int SLEEP, int GROUP>
__device__ __forceinline__ void barrier_relaxed(unsigned* bar, unsigned nblocks) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool release;
    if (GROUP > 1) {
      const unsigned grp = blockIdx.x / GROUP, ngrp = (nblocks + GROUP - 1) / GROUP;
      const unsigned gsize = min((unsigned)GROUP, nblocks - grp * GROUP);
      unsigned* gc = bar + 16 + grp * 16;
      release = false;
      const unsigned p = __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (p == gsize - 1) {
        __hip_atomic_store(gc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        release = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1;
      }
    } else {
      release = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1;
    }
    if (release) {
      __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned spins = 0;
      while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
        __builtin_amdgcn_s_sleep(SLEEP);
        if (++spins > (1u << 22)) { bar[2] = 1; break; }
      }
    }
  }
  __syncthreads();
}

template <int V>
__global__ __launch_bounds__(256) void k(unsigned* bar, float* out, int with_barrier) {
  float x = threadIdx.x;
  if (with_barrier) {
    if (V == 0) barrier_flat<8, true>(bar, gridDim.x);
    if (V == 1) barrier_flat<64, true>(bar, gridDim.x);
    if (V == 2) barrier_flat<8, false>(bar, gridDim.x);
    if (V == 3) barrier_tree<8>(bar, gridDim.x);
    if (V == 4) barrier_tree<32>(bar, gridDim.x);
    if (V == 5) barrier_flat<1, true>(bar, gridDim.x);
    if (V == 6) barrier_relaxed<8, 1>(bar, gridDim.x);
    if (V == 7) barrier_relaxed<2, 1>(bar, gridDim.x);
    if (V == 8) barrier_relaxed<8, 32>(bar, gridDim.x);
    if (V == 9) barrier_relaxed<32, 1>(bar, gridDim.x);
  }
  if (x < 0) out[0] = x;
}

template <int V>
int run(unsigned* bar, float* out, int nb, int wb, const char* name) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<V>, dim3(nb), dim3(256), 0, 0, bar, out, wb);
  CK(hipDeviceSynchronize());
  const int N = 200;
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k<V>, dim3(nb), dim3(256), 0, 0, bar, out, wb);
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  unsigned w[3];
  CK(hipMemcpy(w, bar, 12, hipMemcpyDeviceToHost));
  printf("%-28s blocks %5d  barrier %d : %7.2f us per launch  (flag %u)\n", name, nb, wb, ms * 1e3 / N, w[2]);
  return 0;
}

int main() {
  unsigned* bar;
  float* out;
  CK(hipMalloc(&bar, 1 << 16));
  CK(hipMemset(bar, 0, 1 << 16));
  CK(hipMalloc(&out, 256));
  for (int nb : {256, 512, 1024}) {
    run<0>(bar, out, nb, 0, "empty kernel");
    run<0>(bar, out, nb, 1, "flat sleep8 fence");
    run<1>(bar, out, nb, 1, "flat sleep64 fence");
    run<5>(bar, out, nb, 1, "flat sleep1 fence");
    run<2>(bar, out, nb, 1, "flat sleep8 nofence");
    run<3>(bar, out, nb, 1, "tree32 sleep8");
    run<4>(bar, out, nb, 1, "tree32 sleep32");
    run<6>(bar, out, nb, 1, "relaxed sleep8");
    run<7>(bar, out, nb, 1, "relaxed sleep2");
    run<9>(bar, out, nb, 1, "relaxed sleep32");
    run<8>(bar, out, nb, 1, "relaxed tree32 sleep8");
  }
  return 0;
}
