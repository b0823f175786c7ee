// BatchNorm(train/eval) + SiLU elementwise passes around the MFMA convolutions (gfx950).
//
// Forward (train):  conv epilogue -> per-block (sum, sumsq) partials -> bn_finalize (fp64 combine,
//                   running-stat update) -> bn_silu_apply: a = silu(gamma*(y-mean)*invstd+beta) (+res)
// Backward:         bn_bwd_reduce (sum dz, sum dz*xhat partials) -> bn_bwd_finalize (dgamma, dbeta,
//                   c1, c2) -> bn_bwd_apply: dy = gamma*invstd*(dz - c1 - xhat*c2), residual pass-through.
// All passes are HBM-bound streams: 16-byte (8 x fp16) accesses, one fixed channel group per thread
// so the per-channel coefficients live in registers.
#include "bn_act.h"

namespace {

__device__ __forceinline__ long long view_off(const ViewDesc& v, long long m, int hw) {
  long long b = m / hw;
  long long pix = m - b * hw;
  return b * v.bstride + pix * v.ld;
}

// ---------------------------------------------------------------------------------------------
// partial-slab combine helpers: partials laid out [P][C][2] (fp32)
// block = 256 threads = 16 channels x 16 partial lanes
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void combine_partials(float* part, int P, int C, int c, int lane16, double& s0, double& s1) {
  // the replica slabs are accumulated with atomics by the producer kernel; consume and re-zero them
  double a = 0.0, b = 0.0;
  if (c < C) {
    for (int p = lane16; p < P; p += 16) {
      float* q = part + ((long long)p * C + c) * 2;
      a += (double)q[0];
      b += (double)q[1];
      q[0] = 0.f;
      q[1] = 0.f;
    }
  }
  // lanes of one channel are 16 consecutive threads: reduce with shuffles
  for (int o = 1; o < 16; o <<= 1) {
    a += __shfl_xor(a, o);
    b += __shfl_xor(b, o);
  }
  s0 = a;
  s1 = b;
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(float* part, int P, int C, double count, float momentum, float eps,
                                                          float* mean, float* invstd, float* rmean, float* rvar) {
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int l16 = threadIdx.x & 15;
  double s, ss;
  combine_partials(part, P, C, c, l16, s, ss);
  if (c < C && l16 == 0) {
    double mu = s / count;
    double var = ss / count - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * mu);
    rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * unbiased);
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(float* part, int P, int C, double count, float inv_scale,
                                                              float* c1, float* c2, float* dgamma, float* dbeta) {
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int l16 = threadIdx.x & 15;
  double sdz, sdzx;
  combine_partials(part, P, C, c, l16, sdz, sdzx);
  if (c < C && l16 == 0) {
    c1[c] = (float)(sdz / count);
    c2[c] = (float)(sdzx / count);
    dgamma[c] += (float)(sdzx * inv_scale);
    dbeta[c] += (float)(sdz * inv_scale);
  }
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(float* part, int P, int C, float inv_scale, float* dbias) {
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int l16 = threadIdx.x & 15;
  double s, unused;
  combine_partials(part, P, C, c, l16, s, unused);
  if (c < C && l16 == 0) dbias[c] += (float)(s * inv_scale);
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

// ---------------------------------------------------------------------------------------------
// streaming passes.  Thread layout: CG = C/8 channel groups; thread -> (row slot r, group cg);
// a block walks `rows_per_block` consecutive rows in steps of RP = 256 / CG.
// ---------------------------------------------------------------------------------------------
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

__global__ __launch_bounds__(256) void bn_silu_apply_kernel(const half_t* y, long long M, int C, int hw, BnCoef k, ViewDesc out, ViewDesc res,
                                                            int rows_per_block) {
  const int CG = C >> 3;
  const int RP = 256 / CG;
  const int cg = threadIdx.x % CG, r = threadIdx.x / CG;
  if (r >= RP) return;
  Coef8 s, u;
  load_coef(k, cg * 8, s, u);
  const long long m0 = (long long)blockIdx.x * rows_per_block;
  const long long m1 = min(M, m0 + rows_per_block);
  for (long long m = m0 + r; m < m1; m += RP) {
    h8 v = *reinterpret_cast<const h8*>(y + m * C + cg * 8);
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = cvx_silu((float)v[i] * s.a[i] + s.b[i]);
    if (res.p) {
      h8 rr = *reinterpret_cast<const h8*>(res.p + view_off(res, m, hw) + cg * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] += (float)rr[i];
    }
    h8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (half_t)f[i];
    *reinterpret_cast<h8*>(out.p + view_off(out, m, hw) + cg * 8) = o;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const half_t* y, long long M, int C, int hw, BnCoef k, ViewDesc gout, float* part,
                                                            int rows_per_block) {
  __shared__ float sacc[2 * 512];
  const int CG = C >> 3;
  const int RP = 256 / CG;
  const int cg = threadIdx.x % CG, r = threadIdx.x / CG;
  for (int i = threadIdx.x; i < 2 * C; i += 256) sacc[i] = 0.f;
  __syncthreads();
  if (r < RP) {
    Coef8 s, u;
    load_coef(k, cg * 8, s, u);
    float a1[8], a2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a1[i] = a2[i] = 0.f;
    const long long m0 = (long long)blockIdx.x * rows_per_block;
    const long long m1 = min(M, m0 + rows_per_block);
    for (long long m = m0 + r; m < m1; m += RP) {
      h8 v = *reinterpret_cast<const h8*>(y + m * C + cg * 8);
      h8 g = *reinterpret_cast<const h8*>(gout.p + view_off(gout, m, hw) + cg * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float yy = (float)v[i];
        float dz = (float)g[i] * cvx_silu_grad(yy * s.a[i] + s.b[i]);
        a1[i] += dz;
        a2[i] += dz * ((yy - u.a[i]) * u.b[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      atomicAdd(&sacc[(cg * 8 + i) * 2 + 0], a1[i]);
      atomicAdd(&sacc[(cg * 8 + i) * 2 + 1], a2[i]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) atomicAdd(&part[(long long)(blockIdx.x % CVX_STAT_REPLICAS) * C * 2 + i], sacc[i]);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const half_t* y, long long M, int C, int hw, BnCoef k, const float* c1, const float* c2,
                                                           ViewDesc gout, half_t* dy, ViewDesc gres, int res_accumulate, int rows_per_block) {
  const int CG = C >> 3;
  const int RP = 256 / CG;
  const int cg = threadIdx.x % CG, r = threadIdx.x / CG;
  if (r >= RP) return;
  Coef8 s, u;
  load_coef(k, cg * 8, s, u);
  float k1[8], k2[8], gi[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    k1[i] = c1[cg * 8 + i];
    k2[i] = c2[cg * 8 + i];
    gi[i] = k.gamma[cg * 8 + i] * u.b[i];
  }
  const long long m0 = (long long)blockIdx.x * rows_per_block;
  const long long m1 = min(M, m0 + rows_per_block);
  for (long long m = m0 + r; m < m1; m += RP) {
    h8 v = *reinterpret_cast<const h8*>(y + m * C + cg * 8);
    h8 g = *reinterpret_cast<const h8*>(gout.p + view_off(gout, m, hw) + cg * 8);
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
      half_t* q = gres.p + view_off(gres, m, hw) + cg * 8;
      if (res_accumulate) {
        h8 old = *reinterpret_cast<const h8*>(q);
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] = (half_t)((float)g[i] + (float)old[i]);
      }
      *reinterpret_cast<h8*>(q) = g;
    }
  }
}

// column sums of a [M][C] fp16 view (bias gradient of the head's 1x1 output convs)
__global__ __launch_bounds__(256) void colsum_reduce_kernel(long long M, int C, int hw, ViewDesc g, float* part, int rows_per_block) {
  __shared__ float sacc[2 * 512];
  const int CG = C >> 3;
  const int RP = 256 / CG;
  const int cg = threadIdx.x % CG, r = threadIdx.x / CG;
  for (int i = threadIdx.x; i < 2 * C; i += 256) sacc[i] = 0.f;
  __syncthreads();
  if (r < RP) {
    float a1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a1[i] = 0.f;
    const long long m0 = (long long)blockIdx.x * rows_per_block;
    const long long m1 = min(M, m0 + rows_per_block);
    for (long long m = m0 + r; m < m1; m += RP) {
      h8 v = *reinterpret_cast<const h8*>(g.p + view_off(g, m, hw) + cg * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) a1[i] += (float)v[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(&sacc[(cg * 8 + i) * 2], a1[i]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) atomicAdd(&part[(long long)(blockIdx.x % CVX_STAT_REPLICAS) * C * 2 + i], sacc[i]);
}

}  // namespace

int cvx_stream_rows_per_block(long long M, int C) {
  // aim for ~64 KB of fp16 rows per block, but at least 2048 blocks' worth of parallelism is not needed
  const int CG = C / 8;
  const int RP = 256 / CG;
  long long target = (16 * 1024) / (2LL * C);
  if (target < RP) target = RP;
  long long rows = ((target + RP - 1) / RP) * RP;
  long long blocks = (M + rows - 1) / rows;
  while (blocks > 16384) {
    rows *= 2;
    blocks = (M + rows - 1) / rows;
  }
  return (int)rows;
}
int cvx_stream_blocks(long long M, int C) {
  int rows = cvx_stream_rows_per_block(M, C);
  return (int)((M + rows - 1) / rows);
}

static int check_c(int C) {
  CVX_CHECK(C % 8 == 0 && C >= 8 && C <= 512, "bn_act: C must be a multiple of 8 in [8, 512]");
  return 0;
}

int cvx_bn_finalize(float* part, int P, int C, long long count, float momentum, float eps, float* mean, float* invstd, float* rmean,
                    float* rvar, hipStream_t st) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cvx_cdiv(C, 16)), dim3(256), 0, st, part, P, C, (double)count, momentum, eps, mean, invstd, rmean,
                     rvar);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_bn_fold(int n, const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps, float* scale, float* shift,
                hipStream_t st) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3(cvx_cdiv(n, 256)), dim3(256), 0, st, n, gamma, beta, rmean, rvar, eps, scale, shift);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_bn_silu_apply(const half_t* y, long long M, int C, int hw, const BnCoef& k, const ViewDesc& out, const ViewDesc& res, hipStream_t st) {
  CVX_TRY(check_c(C));
  int rows = cvx_stream_rows_per_block(M, C);
  hipLaunchKernelGGL(bn_silu_apply_kernel, dim3(cvx_stream_blocks(M, C)), dim3(256), 0, st, y, M, C, hw, k, out, res, rows);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_bn_bwd_reduce(const half_t* y, long long M, int C, int hw, const BnCoef& k, const ViewDesc& gout, float* part, hipStream_t st) {
  CVX_TRY(check_c(C));
  int rows = cvx_stream_rows_per_block(M, C);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(cvx_stream_blocks(M, C)), dim3(256), 0, st, y, M, C, hw, k, gout, part, rows);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_bn_bwd_finalize(float* part, int P, int C, long long count, float inv_scale, float* c1, float* c2, float* dgamma, float* dbeta,
                        hipStream_t st) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cvx_cdiv(C, 16)), dim3(256), 0, st, part, P, C, (double)count, inv_scale, c1, c2, dgamma,
                     dbeta);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_bn_bwd_apply(const half_t* y, long long M, int C, int hw, const BnCoef& k, const float* c1, const float* c2, const ViewDesc& gout,
                     half_t* dy, const ViewDesc& gres, int res_accumulate, hipStream_t st) {
  CVX_TRY(check_c(C));
  int rows = cvx_stream_rows_per_block(M, C);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(cvx_stream_blocks(M, C)), dim3(256), 0, st, y, M, C, hw, k, c1, c2, gout, dy, gres,
                     res_accumulate, rows);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_colsum(long long M, int C, int hw, const ViewDesc& g, float* part, float inv_scale, float* dbias, hipStream_t st) {
  CVX_TRY(check_c(C));
  int rows = cvx_stream_rows_per_block(M, C);
  int P = cvx_stream_blocks(M, C);
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3(P), dim3(256), 0, st, M, C, hw, g, part, rows);
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cvx_cdiv(C, 16)), dim3(256), 0, st, part, CVX_STAT_REPLICAS, C, inv_scale, dbias);
  CVX_HIP(hipGetLastError());
  return 0;
}
