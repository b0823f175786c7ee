// BatchNorm(train/eval) + SiLU elementwise passes around the MFMA convolutions (gfx950).
//
// Forward (train):  conv epilogue adds per-channel (sum, sumsq) into 16 replica slabs (64-bit fixed-point atomics) ->
//                   bn_silu_apply: every block folds the replicas into mean/invstd (fp64, identical in all blocks),
//                   block 0 also stores them and updates the running statistics, then
//                   a = silu(gamma*(y-mean)*invstd+beta) (+res)
// Backward:         bn_bwd_reduce (sum dz, sum dz*xhat into replica slabs) -> bn_bwd_apply: every block folds them
//                   into c1/c2, block 0 accumulates dgamma/dbeta, dy = gamma*invstd*(dz - c1 - xhat*c2),
//                   residual gradient pass-through.
// All passes are HBM-bound streams: 16-byte (8 x fp16) accesses, one fixed channel group per thread so the
// per-channel coefficients live in registers.  The replica slabs are zeroed once per step by the engine.
#include "bn_act.h"
#include <cstdlib>

namespace {

constexpr int UNR = 4;  // rows in flight per thread in the streaming passes

__device__ __forceinline__ long long view_off(const ViewDesc& v, long long m, int hw) {
  if (v.bstride == (long long)hw * v.ld) return m * v.ld;  // images are back to back (block-uniform test): no division
  const unsigned mu = (unsigned)m;  // M < 2^32 (checked on the host): one 32-bit division per row
  const unsigned b = mu / (unsigned)hw;
  const unsigned pix = mu - b * (unsigned)hw;
  return (long long)b * v.bstride + (long long)pix * v.ld;
}

// sums the CVX_STAT_REPLICAS fixed-point slabs [R][C][2] into s0/s1 (LDS).  All 256 threads load slab entries in
// parallel (one round of independent, coalesced 16-byte loads) and add them with INTEGER LDS atomics: exact and
// order-independent, and the block pays one memory latency instead of R dependent ones.
__device__ __forceinline__ void fold_replicas(const long long* part, int C, double* s0, double* s1) {
  long long* i0 = reinterpret_cast<long long*>(s0);
  long long* i1 = reinterpret_cast<long long*>(s1);
  for (int c = threadIdx.x; c < C; c += 256) {
    i0[c] = 0;
    i1[c] = 0;
  }
  __syncthreads();
  const int total = CVX_STAT_REPLICAS * C;  // entries of two 64-bit values
  for (int e = threadIdx.x; e < total; e += 256) {
    const longlong2 q = *reinterpret_cast<const longlong2*>(part + (long long)e * 2);
    const int c = e % C;
    atomicAdd(reinterpret_cast<unsigned long long*>(&i0[c]), (unsigned long long)q.x);
    atomicAdd(reinterpret_cast<unsigned long long*>(&i1[c]), (unsigned long long)q.y);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const long long a = i0[c], b = i1[c];
    s0[c] = cvx_fix_to_double(a);
    s1[c] = cvx_fix_to_double(b);
  }
}

// eval: fold running stats into per-channel scale/shift for the conv epilogue
__global__ void bn_fold_kernel(int n, const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps,
                               float* scale, float* shift) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float sc = gamma[i] / sqrtf(rvar[i] + eps);
    scale[i] = sc;
    shift[i] = beta[i] - rmean[i] * sc;
  }
}

__global__ __launch_bounds__(256) void bn_fold_all_kernel(const BnFoldDesc* descs, const float* params, const float* stats, float eps) {
  const BnFoldDesc d = descs[blockIdx.x];
  for (int i = threadIdx.x; i < d.C; i += 256) {
    const float sc = params[d.gamma_off + i] / sqrtf(stats[d.rvar_off + i] + eps);
    d.scale[i] = sc;
    d.shift[i] = params[d.beta_off + i] - stats[d.rmean_off + i] * sc;
  }
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const long long* part, int C, float inv_scale, float* dbias) {
  __shared__ double s0[CVX_BN_MAX_C], s1[CVX_BN_MAX_C];
  fold_replicas(part, C, s0, s1);
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) dbias[c] += (float)(s0[c] * inv_scale);
}

// ---------------------------------------------------------------------------------------------
// streaming passes.  Thread layout: CG = C/8 channel groups; thread -> (row slot r, group cg);
// a block walks `rows_per_block` consecutive rows in steps of RP = 256 / CG.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_silu_apply_kernel(const half_t* y, long long M, int C, int hw, BnTrainArgs a, ViewDesc out,
                                                            ViewDesc res, int rows_per_block) {
  __shared__ double s0[CVX_BN_MAX_C], s1[CVX_BN_MAX_C];
  __shared__ float s_sc[CVX_BN_MAX_C], s_sh[CVX_BN_MAX_C];
  fold_replicas(a.stats, C, s0, s1);
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const double cnt = (double)M;
    double mu = s0[c] / cnt;
    double var = s1[c] / cnt - mu * mu;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)a.eps);
    const float g = a.gamma[c], bt = a.beta[c];
    s_sc[c] = (float)(g * is);
    s_sh[c] = (float)((double)bt - mu * g * is);
    if (blockIdx.x == 0) {
      a.mean[c] = (float)mu;
      a.invstd[c] = (float)is;
      const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
      a.rmean[c] = (float)((1.0 - a.momentum) * (double)a.rmean[c] + a.momentum * mu);
      a.rvar[c] = (float)((1.0 - a.momentum) * (double)a.rvar[c] + a.momentum * unbiased);
    }
  }
  __syncthreads();
  const int CG = C >> 3;
  const int RP = 256 / CG;
  const int cg = threadIdx.x % CG, r = threadIdx.x / CG;
  if (r >= RP) return;
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sc[i] = s_sc[cg * 8 + i];
    sh[i] = s_sh[cg * 8 + i];
  }
  const long long m0 = (long long)blockIdx.x * rows_per_block;
  const long long m1 = min(M, m0 + rows_per_block);
  auto one = [&](long long m, const h8& v, const h8& rr) {
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = cvx_silu((float)v[i] * sc[i] + sh[i]);
    if (res.p) {
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] += (float)rr[i];
    }
    h8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (half_t)f[i];
    *reinterpret_cast<h8*>(out.p + view_off(out, m, hw) + cg * 8) = o;
  };
  // UNR rows per trip with every load issued before the first use: the passes are latency-bound otherwise
  long long m = m0 + r;
  for (; m + (long long)(UNR - 1) * RP < m1; m += (long long)UNR * RP) {
    h8 v[UNR], rr[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      v[u] = *reinterpret_cast<const h8*>(y + (m + u * RP) * C + cg * 8);
      if (res.p) rr[u] = *reinterpret_cast<const h8*>(res.p + view_off(res, m + u * RP, hw) + cg * 8);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) one(m + u * RP, v[u], rr[u]);
  }
  for (; m < m1; m += RP) {
    h8 v = *reinterpret_cast<const h8*>(y + m * C + cg * 8), rr = {};
    if (res.p) rr = *reinterpret_cast<const h8*>(res.p + view_off(res, m, hw) + cg * 8);
    one(m, v, rr);
  }
}

struct Coef8 {
  float a[8], b[8];
};

__device__ __forceinline__ void load_coef(const BnCoef& k, int c0, Coef8& sc_sh, Coef8& mu_is) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float mu = k.mean[c0 + i], is = k.invstd[c0 + i];
    float g = k.gamma[c0 + i], bt = k.beta[c0 + i];
    sc_sh.a[i] = g * is;
    sc_sh.b[i] = bt - mu * g * is;
    mu_is.a[i] = mu;
    mu_is.b[i] = is;
  }
}

// block-level accumulation of per-thread channel sums, bit-reproducible.  Lanes of a wave that share a channel group
// (lane % CG, when CG divides 64) are folded with a fixed xor-butterfly; each wave parks its CG*8*NV sums in its own
// LDS slot; NV*C threads add the four wave slots in fixed order and issue ONE 64-bit fixed-point atomic per value.
// Channel-group counts that do not divide 64 (C = 80, 144: Detect head) take the parked-partials column walk instead.
template <int NV>
__device__ __forceinline__ void block_channel_sums(float (&v)[NV][8], int C, int CG, int cg, bool active, float* sred, long long* part) {
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
      cvx_fix_atomic_add(&part[((long long)(blockIdx.x % CVX_STAT_REPLICAS) * C + c) * 2 + q], acc);
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
    cvx_fix_atomic_add(&part[((long long)(blockIdx.x % CVX_STAT_REPLICAS) * C + c) * 2 + q], acc);
  }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const half_t* y, long long M, int C, int hw, BnCoef k, ViewDesc gout, long long* part,
                                                            int rows_per_block) {
  __shared__ float sacc[256 * 16];
  const int CG = C >> 3;
  const int RP = 256 / CG;
  const int cg = threadIdx.x % CG, r = threadIdx.x / CG;
  const bool active = r < RP;
  float acc[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[0][i] = acc[1][i] = 0.f;
  if (active) {
    Coef8 s, u;
    load_coef(k, cg * 8, s, u);
    const long long m0 = (long long)blockIdx.x * rows_per_block;
    const long long m1 = min(M, m0 + rows_per_block);
    auto one = [&](const h8& v, const h8& g) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float yy = (float)v[i];
        float dz = (float)g[i] * cvx_silu_grad(yy * s.a[i] + s.b[i]);
        acc[0][i] += dz;
        acc[1][i] += dz * ((yy - u.a[i]) * u.b[i]);
      }
    };
    long long m = m0 + r;
    for (; m + (long long)(UNR - 1) * RP < m1; m += (long long)UNR * RP) {
      h8 v[UNR], g[UNR];
#pragma unroll
      for (int q = 0; q < UNR; ++q) {
        v[q] = *reinterpret_cast<const h8*>(y + (m + q * RP) * C + cg * 8);
        g[q] = *reinterpret_cast<const h8*>(gout.p + view_off(gout, m + q * RP, hw) + cg * 8);
      }
#pragma unroll
      for (int q = 0; q < UNR; ++q) one(v[q], g[q]);
    }
    for (; m < m1; m += RP) {
      h8 v = *reinterpret_cast<const h8*>(y + m * C + cg * 8);
      h8 g = *reinterpret_cast<const h8*>(gout.p + view_off(gout, m, hw) + cg * 8);
      one(v, g);
    }
  }
  block_channel_sums<2>(acc, C, CG, cg, active, sacc, part);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const half_t* y, long long M, int C, int hw, BnCoef k, const long long* part,
                                                           float inv_scale, float* dgamma, float* dbeta, ViewDesc gout, half_t* dy, ViewDesc gres,
                                                           int res_accumulate, int rows_per_block) {
  __shared__ double s0[CVX_BN_MAX_C], s1[CVX_BN_MAX_C];
  fold_replicas(part, C, s0, s1);
  __syncthreads();
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      dgamma[c] += (float)(s1[c] * inv_scale);
      dbeta[c] += (float)(s0[c] * inv_scale);
    }
  }
  const int CG = C >> 3;
  const int RP = 256 / CG;
  const int cg = threadIdx.x % CG, r = threadIdx.x / CG;
  if (r >= RP) return;
  Coef8 s, u;
  load_coef(k, cg * 8, s, u);
  float k1[8], k2[8], gi[8];
  const double cnt = (double)M;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    k1[i] = (float)(s0[cg * 8 + i] / cnt);
    k2[i] = (float)(s1[cg * 8 + i] / cnt);
    gi[i] = k.gamma[cg * 8 + i] * u.b[i];
  }
  const long long m0 = (long long)blockIdx.x * rows_per_block;
  const long long m1 = min(M, m0 + rows_per_block);
  auto one = [&](long long m, const h8& v, h8 g, const h8& old) {
    h8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float yy = (float)v[i];
      float dz = (float)g[i] * cvx_silu_grad(yy * s.a[i] + s.b[i]);
      float xh = (yy - u.a[i]) * u.b[i];
      o[i] = (half_t)(gi[i] * (dz - k1[i] - xh * k2[i]));
    }
    *reinterpret_cast<h8*>(dy + m * C + cg * 8) = o;
    if (gres.p) {
      if (res_accumulate) {
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] = (half_t)((float)g[i] + (float)old[i]);
      }
      *reinterpret_cast<h8*>(gres.p + view_off(gres, m, hw) + cg * 8) = g;
    }
  };
  const bool rd_old = gres.p && res_accumulate;
  long long m = m0 + r;
  for (; m + (long long)(UNR - 1) * RP < m1; m += (long long)UNR * RP) {
    h8 v[UNR], g[UNR], old[UNR];
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      v[q] = *reinterpret_cast<const h8*>(y + (m + q * RP) * C + cg * 8);
      g[q] = *reinterpret_cast<const h8*>(gout.p + view_off(gout, m + q * RP, hw) + cg * 8);
      if (rd_old) old[q] = *reinterpret_cast<const h8*>(gres.p + view_off(gres, m + q * RP, hw) + cg * 8);
    }
#pragma unroll
    for (int q = 0; q < UNR; ++q) one(m + q * RP, v[q], g[q], old[q]);
  }
  for (; m < m1; m += RP) {
    h8 v = *reinterpret_cast<const h8*>(y + m * C + cg * 8);
    h8 g = *reinterpret_cast<const h8*>(gout.p + view_off(gout, m, hw) + cg * 8);
    h8 old = {};
    if (rd_old) old = *reinterpret_cast<const h8*>(gres.p + view_off(gres, m, hw) + cg * 8);
    one(m, v, g, old);
  }
}

// column sums of a [M][C] fp16 view (bias gradient of the head's 1x1 output convs)
__global__ __launch_bounds__(256) void colsum_reduce_kernel(long long M, int C, int hw, ViewDesc g, long long* part, int rows_per_block) {
  __shared__ float sacc[256 * 8];
  const int CG = C >> 3;
  const int RP = 256 / CG;
  const int cg = threadIdx.x % CG, r = threadIdx.x / CG;
  const bool active = r < RP;
  float acc[1][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[0][i] = 0.f;
  if (active) {
    const long long m0 = (long long)blockIdx.x * rows_per_block;
    const long long m1 = min(M, m0 + rows_per_block);
    for (long long m = m0 + r; m < m1; m += RP) {
      h8 v = *reinterpret_cast<const h8*>(g.p + view_off(g, m, hw) + cg * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[0][i] += (float)v[i];
    }
  }
  block_channel_sums<1>(acc, C, CG, cg, active, sacc, part);
}

}  // namespace

int cvx_stream_rows_per_block(long long M, int C, int kb_per_block) {
  const int CG = C / 8;
  const int RP = 256 / CG;
  static const int kb_env = getenv("CVX_BN_KB") ? atoi(getenv("CVX_BN_KB")) : 0;  // tuning experiments only
  if (kb_env > 0) kb_per_block = kb_env;
  long long target = ((long long)kb_per_block * 1024) / (2LL * C);
  if (target < RP) target = RP;
  long long rows = ((target + RP - 1) / RP) * RP;
  long long blocks = (M + rows - 1) / rows;
  while (blocks > 16384) {
    rows *= 2;
    blocks = (M + rows - 1) / rows;
  }
  return (int)rows;
}
static int blocks_for(long long M, int rows) { return (int)((M + rows - 1) / rows); }

static int check_c(int C, long long M = 0) {
  CVX_CHECK(C % 8 == 0 && C >= 8 && C <= CVX_BN_MAX_C, "bn_act: C must be a multiple of 8 in [8, 1024]");
  CVX_CHECK(M < (1LL << 32), "bn_act: more than 2^32 rows");
  return 0;
}

int cvx_bn_fold_all(const BnFoldDesc* descs, int n, const float* params, const float* stats, float eps, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(bn_fold_all_kernel, dim3(n), dim3(256), 0, st, descs, params, stats, eps);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_bn_fold(int n, const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps, float* scale, float* shift,
                hipStream_t st) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3(cvx_cdiv(n, 256)), dim3(256), 0, st, n, gamma, beta, rmean, rvar, eps, scale, shift);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_bn_silu_apply(const half_t* y, long long M, int C, int hw, const BnTrainArgs& a, const ViewDesc& out, const ViewDesc& res,
                      hipStream_t st) {
  CVX_TRY(check_c(C, M));
  int rows = cvx_stream_rows_per_block(M, C, 16);
  hipLaunchKernelGGL(bn_silu_apply_kernel, dim3(blocks_for(M, rows)), dim3(256), 0, st, y, M, C, hw, a, out, res, rows);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_bn_bwd_reduce(const half_t* y, long long M, int C, int hw, const BnCoef& k, const ViewDesc& gout, long long* part, hipStream_t st) {
  CVX_TRY(check_c(C, M));
  int rows = cvx_stream_rows_per_block(M, C, 32);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(blocks_for(M, rows)), dim3(256), 0, st, y, M, C, hw, k, gout, part, rows);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_bn_bwd_apply(const half_t* y, long long M, int C, int hw, const BnCoef& k, const long long* part, float inv_scale, float* dgamma,
                     float* dbeta, const ViewDesc& gout, half_t* dy, const ViewDesc& gres, int res_accumulate, hipStream_t st) {
  CVX_TRY(check_c(C, M));
  int rows = cvx_stream_rows_per_block(M, C, 32);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks_for(M, rows)), dim3(256), 0, st, y, M, C, hw, k, part, inv_scale, dgamma, dbeta, gout,
                     dy, gres, res_accumulate, rows);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_colsum(long long M, int C, int hw, const ViewDesc& g, long long* part, float inv_scale, float* dbias, hipStream_t st) {
  CVX_TRY(check_c(C, M));
  int rows = cvx_stream_rows_per_block(M, C, 32);
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3(blocks_for(M, rows)), dim3(256), 0, st, M, C, hw, g, part, rows);
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(1), dim3(256), 0, st, part, C, inv_scale, dbias);
  CVX_HIP(hipGetLastError());
  return 0;
}
