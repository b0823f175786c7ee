// Device helpers shared by the BatchNorm streaming passes (bn_act.hip) and the fp32 stem kernels (stem.hip).
#pragma once
#include "bn_act.h"

namespace cvx_bn {

__device__ __forceinline__ long long view_off(const ViewDesc& v, long long m, int hw) {
  if (v.bstride == (long long)hw * v.ld) return m * v.ld;  // images are back to back (block-uniform test): no division
  const unsigned mu = (unsigned)m;  // M < 2^32 (checked on the host): one 32-bit division per row
  const unsigned b = mu / (unsigned)hw;
  const unsigned pix = mu - b * (unsigned)hw;
  return (long long)b * v.bstride + (long long)pix * v.ld;
}

// LDS bytes fold_replicas needs for C channels: [C][2 values][CVX_FIX_WORDS] 64-bit accumulators, reused for the result
__host__ __device__ constexpr int fold_ws_bytes(int C) { return C * 2 * CVX_FIX_WORDS * 8; }

// Sums the cvx_stat_replicas(C) fixed-point slabs [R][C][2][CVX_FIX_WORDS] and leaves the two per-channel totals as doubles
// in ws: value q of channel c at ((double*)ws)[q * C + c].  All 256 threads load slab entries in parallel (one round of
// independent, coalesced 16-byte loads) and add them with INTEGER LDS atomics: exact and order-independent, and the
// block pays one memory latency instead of R dependent ones.  Every thread of the 256-thread block must call it; it ends
// with a barrier.
__device__ __forceinline__ void fold_replicas(const long long* part, int C, long long* ws) {
  const int nacc = C * 2 * CVX_FIX_WORDS;
  for (int i = threadIdx.x; i < nacc; i += 256) ws[i] = 0;
  __syncthreads();
  const int total = cvx_stat_replicas(C) * C * 2;  // (replica, channel, value) entries of two 64-bit words (coarse, fine)
  for (int e = threadIdx.x; e < total; e += 256) {
    const longlong2 q = *reinterpret_cast<const longlong2*>(part + (long long)e * CVX_FIX_WORDS);
    const int cv = e % (C * 2);  // channel * 2 + value
    atomicAdd(reinterpret_cast<unsigned long long*>(&ws[cv * 2]), (unsigned long long)q.x);
    atomicAdd(reinterpret_cast<unsigned long long*>(&ws[cv * 2 + 1]), (unsigned long long)q.y);
  }
  __syncthreads();
  // convert in place: hold this thread's results in registers across the barrier (C <= 2048: at most 16 per thread)
  double r[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int cv = threadIdx.x + 256 * k;
    r[k] = cv < C * 2 ? cvx_fix_to_double(ws[cv * 2], ws[cv * 2 + 1]) : 0.0;
  }
  __syncthreads();
  double* out = reinterpret_cast<double*>(ws);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int cv = threadIdx.x + 256 * k;
    if (cv < C * 2) out[(cv & 1) * C + (cv >> 1)] = r[k];
  }
  __syncthreads();
}

// block-level accumulation of per-thread channel sums, bit-reproducible.  Lanes of a wave that share a channel group
// (lane % CG, when CG divides 64) are folded with a fixed xor-butterfly; each wave parks its CG*8*NV sums in its own
// LDS slot; NV*C threads add the four wave slots in fixed order and issue ONE fixed-point atomic add per value.
// Channel-group counts that do not divide 64 (C = 80, 144: Detect head) take the parked-partials column walk instead.
// part: replica slabs [R][C][2] fixed-point values; NV = 1 fills value 0 only.
// rep_block: the block's ordinal among the blocks that feed `part` (picks the replica slab)
template <int NV>
__device__ __forceinline__ void block_channel_sums(float (&v)[NV][8], int C, int CG, int cg, bool active, float* sred, long long* part,
                                                   int rep_block) {
  const bool pow2 = (64 % CG) == 0;
  if (pow2) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < NV; ++q)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float x = v[q][i];  // every thread is active when CG divides 64 (RP * CG == 256)
        for (int o = CG; o < 64; o <<= 1) x += __shfl_xor(x, o);
        if (lane < CG) sred[(wave * C + cg * 8 + i) * NV + q] = x;
      }
    __syncthreads();
    for (int j = threadIdx.x; j < NV * C; j += 256) {
      const int c = j / NV, q = j - c * NV;
      const float acc = (sred[(0 * C + c) * NV + q] + sred[(1 * C + c) * NV + q]) + (sred[(2 * C + c) * NV + q] + sred[(3 * C + c) * NV + q]);
      cvx_fix_atomic_add(part, ((long long)(rep_block % cvx_stat_replicas(C)) * C + c) * 2 + q, acc);
    }
    return;
  }
  const int RP = 256 / CG;
  if (active) {
    float* dst = sred + (size_t)threadIdx.x * (NV * 8);
#pragma unroll
    for (int q = 0; q < NV; ++q)
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[q * 8 + i] = v[q][i];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < NV * C; j += 256) {
    const int c = j / NV, q = j - c * NV;
    const int g = c >> 3, i = c & 7;
    float acc = 0.f;
    for (int r = 0; r < RP; ++r) acc += sred[(size_t)(r * CG + g) * (NV * 8) + q * 8 + i];
    cvx_fix_atomic_add(part, ((long long)(rep_block % cvx_stat_replicas(C)) * C + c) * 2 + q, acc);
  }
}

// batch statistics of C channels from the folded sums (ws as left by fold_replicas) -> mean / invstd (fp32, LDS or
// registers of the caller); thread 0's block also publishes them and updates the running statistics.
struct BnMoments {
  float mean, invstd;
};
__device__ __forceinline__ BnMoments moments_of(const long long* ws, int C, int c, long long M, float eps, double* var_out) {
  const double* s = reinterpret_cast<const double*>(ws);
  const double cnt = (double)M;
  const double mu = s[c] / cnt;
  double var = s[C + c] / cnt - mu * mu;
  if (var < 0.0) var = 0.0;
  if (var_out) *var_out = var;
  return BnMoments{(float)mu, (float)(1.0 / sqrt(var + (double)eps))};
}

}  // namespace cvx_bn
