// BatchNorm(train/eval) + SiLU elementwise passes around the MFMA convolutions (gfx950).
//
// Forward (train):  the conv epilogue stores the raw output y in FP32 (a transient buffer shared by all layers) and adds
//                   per-channel (sum, sumsq) into 16 replica slabs (fixed-point integer atomics) ->
//                   bn_silu_apply: every block folds the replicas into mean/invstd (fp64, identical in all blocks),
//                   block 0 also stores them and updates the running statistics, then
//                   xhat = (y-mean)*invstd  (kept, fp16, for the backward pass),  a = silu(gamma*xhat+beta) (+res).
//                   The normalisation therefore reads the un-rounded accumulators: rounding y to fp16 first was the
//                   largest single contribution to the forward error against the fp32 reference.
// Backward:         bn_bwd_reduce (sum dz, sum dz*xhat into replica slabs) -> bn_bwd_apply: every block folds them
//                   into c1/c2, block 0 accumulates dgamma/dbeta, dy = gamma*invstd*(dz - c1 - xhat*c2),
//                   residual gradient pass-through.
// All passes are HBM-bound streams: 16-byte accesses, one fixed channel group per thread so the per-channel
// coefficients live in registers.  The replica slabs are zeroed once per step by the engine.
#include "bn_common.h"

namespace {
using namespace cvx_bn;

constexpr int UNR = 4;  // rows in flight per thread in the streaming passes

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
    if (d.bias_only) {
      d.scale[i] = 1.f;
      d.shift[i] = params[d.beta_off + i];
      continue;
    }
    const float sc = params[d.gamma_off + i] / sqrtf(stats[d.rvar_off + i] + eps);
    d.scale[i] = sc;
    const float cb = d.cbias_off >= 0 ? params[d.cbias_off + i] : 0.f;  // bn(conv + b) = sc * conv + (beta + sc * (b - mean))
    d.shift[i] = params[d.beta_off + i] + (cb - stats[d.rmean_off + i]) * sc;
  }
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const long long* part, int C, float inv_scale, float* dbias) {
  extern __shared__ __attribute__((aligned(16))) long long ws[];
  fold_replicas(part, C, ws);
  const double* s0 = reinterpret_cast<const double*>(ws);
  for (int c = threadIdx.x; c < C; c += 256) dbias[c] += (float)(s0[c] * inv_scale);
}

// ---------------------------------------------------------------------------------------------
// streaming passes.  Thread layout: CG = C/8 channel groups; thread -> (row slot r, group cg);
// a block walks `rows_per_block` consecutive rows in steps of RP = 256 / CG.
// ---------------------------------------------------------------------------------------------
struct f8 {
  f4 lo, hi;
  __device__ __forceinline__ float operator[](int i) const { return i < 4 ? lo[i] : hi[i - 4]; }
};
__device__ __forceinline__ f8 load_f8(const float* p) { return f8{*reinterpret_cast<const f4*>(p), *reinterpret_cast<const f4*>(p + 4)}; }

// activation kinds of the training passes (cvx_op_desc.act -> kind): 0 SiLU, 1 ReLU, 2 none
template <int ACT>
__device__ __forceinline__ float act_fwd(float z) {
  if constexpr (ACT == 0) return cvx_silu(z);
  if constexpr (ACT == 1) return fmaxf(z, 0.f);
  return z;
}

// RES_PRE: the residual joins the pre-activation (ResNet Bottleneck: relu(bn(conv) + identity)), else it is added to the output
template <int ACT, bool RES_PRE>
__global__ __launch_bounds__(256) void bn_act_apply_kernel(const float* y, long long M, int C, int hw, BnTrainArgs a, ViewDesc out,
                                                           ViewDesc res, half_t* xhat, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) long long ws[];  // fold workspace | mean, invstd (2*C floats)
  float* s_mu = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + fold_ws_bytes(C));
  float* s_is = s_mu + C;
  fold_replicas(a.stats, C, ws);
  for (int c = threadIdx.x; c < C; c += 256) {
    double var;
    const BnMoments mo = moments_of(ws, C, c, M, a.eps, &var);
    s_mu[c] = mo.mean;
    s_is[c] = mo.invstd;
    if (blockIdx.x == 0) {
      const double cnt = (double)M;
      const double mu = reinterpret_cast<const double*>(ws)[c] / cnt;
      a.mean[c] = mo.mean;
      a.invstd[c] = mo.invstd;
      const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
      a.rmean[c] = (float)((1.0 - a.momentum) * (double)a.rmean[c] + a.momentum * (mu + (a.cbias ? (double)a.cbias[c] : 0.0)));
      a.rvar[c] = (float)((1.0 - a.momentum) * (double)a.rvar[c] + a.momentum * unbiased);
    }
  }
  __syncthreads();
  const int CG = C >> 3;
  const int RP = 256 / CG;
  const int cg = threadIdx.x % CG, r = threadIdx.x / CG;
  if (r >= RP) return;
  float mu[8], is[8], ga[8], be[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    mu[i] = s_mu[cg * 8 + i];
    is[i] = s_is[cg * 8 + i];
    ga[i] = a.gamma[cg * 8 + i];
    be[i] = a.beta[cg * 8 + i];
  }
  const long long m0 = (long long)blockIdx.x * rows_per_block;
  const long long m1 = min(M, m0 + rows_per_block);
  auto one = [&](long long m, const f8& v, const h8& rr) {
    float f[8];
    h8 xh;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = (v[i] - mu[i]) * is[i];
      xh[i] = (half_t)x;
      if constexpr (RES_PRE)
        f[i] = act_fwd<ACT>(x * ga[i] + be[i] + (float)rr[i]);
      else
        f[i] = act_fwd<ACT>(x * ga[i] + be[i]);
    }
    if (!RES_PRE && res.p) {
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] += (float)rr[i];
    }
    h8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (half_t)f[i];
    if constexpr (ACT == 1) {
      // ReLU: the backward mask [out > 0] rides in the lowest mantissa bit of the kept xhat (one ulp of a value the backward only
      // uses in sums): the backward passes then need neither the output nor the pre-activation
      unsigned short* xb = reinterpret_cast<unsigned short*>(&xh);
#pragma unroll
      for (int i = 0; i < 8; ++i) xb[i] = (unsigned short)((xb[i] & 0xFFFEu) | ((float)o[i] > 0.f ? 1u : 0u));
    }
    *reinterpret_cast<h8*>(xhat + m * C + cg * 8) = xh;
    *reinterpret_cast<h8*>(out.p + view_off(out, m, hw) + cg * 8) = o;
  };
  // UNR rows per trip with every load issued before the first use: the passes are latency-bound otherwise
  long long m = m0 + r;
  for (; m + (long long)(UNR - 1) * RP < m1; m += (long long)UNR * RP) {
    f8 v[UNR];
    h8 rr[UNR] = {};
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      v[u] = load_f8(y + (m + u * RP) * C + cg * 8);
      if (res.p) rr[u] = *reinterpret_cast<const h8*>(res.p + view_off(res, m + u * RP, hw) + cg * 8);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) one(m + u * RP, v[u], rr[u]);
  }
  for (; m < m1; m += RP) {
    f8 v = load_f8(y + m * C + cg * 8);
    h8 rr = {};
    if (res.p) rr = *reinterpret_cast<const h8*>(res.p + view_off(res, m, hw) + cg * 8);
    one(m, v, rr);
  }
}

struct Coef8 {
  float a[8], b[8];
};

// dz = g * act'(pre).  SiLU: pre = gamma * xhat + beta is recomputed from the kept xhat.  ReLU: the mask is [out > 0] of the layer's
// own fp16 forward output, kept in the lowest mantissa bit of xhat by the forward pass: recomputing the pre-activation from the
// rounded xhat would flip the mask of ~2e-4 of the elements (|pre| below the fp16 rounding of xhat), a 1 % gradient error per layer,
// and reading the output tensor again would cost 2 of the pass's ~8 bytes per element.
// SiLU with a pre-activation residual (YOLOv7's RepConv, yolov7_model.py: silu(bn(conv3x3) + bn(conv1x1))): `fo` carries the residual
// VALUE (the other branch's output) instead, and pre = gamma * xhat + beta + residual.
template <int ACT>
__device__ __forceinline__ float act_dz(float g, float xh, float ga, float be, float fo) {
  if constexpr (ACT == 0) return g * cvx_silu_grad(xh * ga + be);
  if constexpr (ACT == 3) return g * cvx_silu_grad(xh * ga + be + fo);  // kernel-internal kind: SiLU whose pre-activation holds a residual
  if constexpr (ACT == 1) return fo > 0.f ? g : 0.f;
  return g;
}

template <int ACT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const half_t* xhat, long long M, int C, int hw, BnCoef k, ViewDesc gout, ViewDesc fout,
                                                            long long* part, int rows_per_block) {
  __shared__ float sacc[256 * 16];
  const int CG = C >> 3;
  const int RP = 256 / CG;
  const int cg = threadIdx.x % CG, r = threadIdx.x / CG;
  const bool active = r < RP;
  float acc[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[0][i] = acc[1][i] = 0.f;
  if (active) {
    Coef8 s;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s.a[i] = k.gamma[cg * 8 + i];
      s.b[i] = k.beta[cg * 8 + i];
    }
    const long long m0 = (long long)blockIdx.x * rows_per_block;
    const long long m1 = min(M, m0 + rows_per_block);
    auto one = [&](const h8& v, const h8& g, const h8& fo) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float xh = (float)v[i];
        const float fb = ACT == 1 ? (float)(reinterpret_cast<const unsigned short*>(&v)[i] & 1u) : (float)fo[i];  // ReLU: the mask bit of xhat
        float dz = act_dz<ACT>((float)g[i], xh, s.a[i], s.b[i], fb);
        acc[0][i] += dz;
        acc[1][i] += dz * xh;
      }
    };
    long long m = m0 + r;
    for (; m + (long long)(UNR - 1) * RP < m1; m += (long long)UNR * RP) {
      h8 v[UNR], g[UNR], fo[UNR] = {};
#pragma unroll
      for (int q = 0; q < UNR; ++q) {
        v[q] = *reinterpret_cast<const h8*>(xhat + (m + q * RP) * C + cg * 8);
        g[q] = *reinterpret_cast<const h8*>(gout.p + view_off(gout, m + q * RP, hw) + cg * 8);
        if constexpr (ACT == 3) fo[q] = *reinterpret_cast<const h8*>(fout.p + view_off(fout, m + q * RP, hw) + cg * 8);
      }
#pragma unroll
      for (int q = 0; q < UNR; ++q) one(v[q], g[q], fo[q]);
    }
    for (; m < m1; m += RP) {
      h8 v = *reinterpret_cast<const h8*>(xhat + m * C + cg * 8);
      h8 g = *reinterpret_cast<const h8*>(gout.p + view_off(gout, m, hw) + cg * 8);
      h8 fo = {};
      if constexpr (ACT == 3) fo = *reinterpret_cast<const h8*>(fout.p + view_off(fout, m, hw) + cg * 8);
      one(v, g, fo);
    }
  }
  block_channel_sums<2>(acc, C, CG, cg, active, sacc, part, blockIdx.x);
}

// RES_PRE: the residual branch receives dz (the gradient of the shared pre-activation), else the incoming gradient g
template <int ACT, bool RES_PRE>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const half_t* xhat, long long M, int C, int hw, BnCoef k, const long long* part,
                                                           float inv_scale, float* dgamma, float* dbeta, ViewDesc gout, ViewDesc fout, half_t* dy,
                                                           ViewDesc gres, int res_accumulate, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) long long ws[];
  fold_replicas(part, C, ws);
  const double* s0 = reinterpret_cast<const double*>(ws);
  const double* s1 = s0 + C;
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
  Coef8 s;
  float k1[8], k2[8], gi[8];
  const double cnt = (double)M;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s.a[i] = k.gamma[cg * 8 + i];
    s.b[i] = k.beta[cg * 8 + i];
    k1[i] = (float)(s0[cg * 8 + i] / cnt);
    k2[i] = (float)(s1[cg * 8 + i] / cnt);
    gi[i] = s.a[i] * k.invstd[cg * 8 + i];
  }
  const long long m0 = (long long)blockIdx.x * rows_per_block;
  const long long m1 = min(M, m0 + rows_per_block);
  auto one = [&](long long m, const h8& v, h8 g, const h8& old, const h8& fo) {
    h8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float xh = (float)v[i];
      const float fb = ACT == 1 ? (float)(reinterpret_cast<const unsigned short*>(&v)[i] & 1u) : (float)fo[i];  // ReLU: the mask bit of xhat
      float dz = act_dz<ACT>((float)g[i], xh, s.a[i], s.b[i], fb);
      o[i] = (half_t)(gi[i] * (dz - k1[i] - xh * k2[i]));
      if constexpr (RES_PRE) g[i] = (half_t)dz;
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
    h8 v[UNR], g[UNR], old[UNR] = {}, fo[UNR] = {};
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      v[q] = *reinterpret_cast<const h8*>(xhat + (m + q * RP) * C + cg * 8);
      g[q] = *reinterpret_cast<const h8*>(gout.p + view_off(gout, m + q * RP, hw) + cg * 8);
      if (rd_old) old[q] = *reinterpret_cast<const h8*>(gres.p + view_off(gres, m + q * RP, hw) + cg * 8);
      if constexpr (ACT == 3) fo[q] = *reinterpret_cast<const h8*>(fout.p + view_off(fout, m + q * RP, hw) + cg * 8);
    }
#pragma unroll
    for (int q = 0; q < UNR; ++q) one(m + q * RP, v[q], g[q], old[q], fo[q]);
  }
  for (; m < m1; m += RP) {
    h8 v = *reinterpret_cast<const h8*>(xhat + m * C + cg * 8);
    h8 g = *reinterpret_cast<const h8*>(gout.p + view_off(gout, m, hw) + cg * 8);
    h8 old = {}, fo = {};
    if (rd_old) old = *reinterpret_cast<const h8*>(gres.p + view_off(gres, m, hw) + cg * 8);
    if constexpr (ACT == 3) fo = *reinterpret_cast<const h8*>(fout.p + view_off(fout, m, hw) + cg * 8);
    one(m, v, g, old, fo);
  }
}

// per-channel (sum, sumsq) of an fp32 [M][C] tensor into the replica slabs: what the conv epilogues do in the engine;
// stand-alone for the single-op entry point (unit tests, other callers with an fp32 pre-activation of their own)
__global__ __launch_bounds__(256) void bn_stats_f32_kernel(const float* y, long long M, int C, long long* part, int rows_per_block) {
  __shared__ float sacc[256 * 16];
  const int CG = C >> 3;
  const int RP = 256 / CG;
  const int cg = threadIdx.x % CG, r = threadIdx.x / CG;
  const bool active = r < RP;
  float acc[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[0][i] = acc[1][i] = 0.f;
  if (active) {
    const long long m0 = (long long)blockIdx.x * rows_per_block;
    const long long m1 = min(M, m0 + rows_per_block);
    for (long long m = m0 + r; m < m1; m += RP) {
      const f8 v = load_f8(y + m * C + cg * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[0][i] += v[i];
        acc[1][i] = fmaf(v[i], v[i], acc[1][i]);
      }
    }
  }
  block_channel_sums<2>(acc, C, CG, cg, active, sacc, part, blockIdx.x);
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
  block_channel_sums<1>(acc, C, CG, cg, active, sacc, part, blockIdx.x);
}

// every bias gradient of the network in two launches (the head's six 1x1 output convs: twelve launches one by one, at the
// very start of the backward pass): block -> (tensor, row chunk) through a table, one finalize block per tensor
__global__ __launch_bounds__(256) void colsum_multi_reduce_kernel(const half_t* base, const ColsumDesc* descs, const ColsumBlock* blocks) {
  __shared__ float sacc[256 * 8];
  const ColsumBlock br = blocks[blockIdx.x];
  const ColsumDesc d = descs[br.desc];
  const ViewDesc g{const_cast<half_t*>(base) + d.off, d.bstride, d.ld};
  const int C = d.C, hw = d.hw;
  const int CG = C >> 3;
  const int RP = 256 / CG;
  const int cg = threadIdx.x % CG, r = threadIdx.x / CG;
  const bool active = r < RP;
  float acc[1][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[0][i] = 0.f;
  if (active) {
    const long long m0 = (long long)br.block * d.rows_per_block;
    const long long m1 = min(d.M, m0 + d.rows_per_block);
    long long m = m0 + r;
    for (; m + (long long)(UNR - 1) * RP < m1; m += (long long)UNR * RP) {
      h8 v[UNR];
#pragma unroll
      for (int q = 0; q < UNR; ++q) v[q] = *reinterpret_cast<const h8*>(g.p + view_off(g, m + q * RP, hw) + cg * 8);
#pragma unroll
      for (int q = 0; q < UNR; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[0][i] += (float)v[q][i];
    }
    for (; m < m1; m += RP) {
      h8 v = *reinterpret_cast<const h8*>(g.p + view_off(g, m, hw) + cg * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[0][i] += (float)v[i];
    }
  }
  block_channel_sums<1>(acc, C, CG, cg, active, sacc, d.part, br.block);
}
__global__ __launch_bounds__(256) void colsum_multi_finalize_kernel(const ColsumDesc* descs, float inv_scale, float* grads) {
  extern __shared__ __attribute__((aligned(16))) long long ws[];
  const ColsumDesc d = descs[blockIdx.x];
  fold_replicas(d.part, d.C, ws);
  const double* s0 = reinterpret_cast<const double*>(ws);
  for (int c = threadIdx.x; c < d.C; c += 256) grads[d.dbias_off + c] += (float)(s0[c] * inv_scale);
}

}  // namespace

int cvx_stream_rows_per_block(long long M, int C, int kb_per_block) {
  const int CG = C / 8;
  const int RP = 256 / CG;
  static const int kb_env = cvx_tune_int("CVX_BN_KB", 0);
  if (kb_env > 0) kb_per_block = kb_env;
  // wide layers (ResNet's 512..2048 channels): every block folds R*C*32 bytes of statistic slabs before it streams, so its share of
  // the tensor must be several times that -- at least 64 rows
  if (C > 256 && kb_per_block < C / 8) kb_per_block = C / 8;
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
  CVX_CHECK(C % 8 == 0 && C >= 8 && C <= CVX_BN_MAX_C, "bn_act: C must be a multiple of 8 in [8, 2048]");
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
static int check_act(int act, int res_pre, bool has_res) {
  CVX_CHECK(act >= 0 && act <= 2, "bn_act: activation kind must be 0 (SiLU), 1 (ReLU) or 2 (none)");
  CVX_CHECK(!res_pre || has_res, "bn_act: res_pre without a residual");
  CVX_CHECK(!(has_res && act == 1 && !res_pre), "bn_act: ReLU with a post-activation residual is not built (the backward mask is the output's sign)");
  return 0;
}
int cvx_bn_act_apply(const float* y, long long M, int C, int hw, const BnTrainArgs& a, const ViewDesc& out, const ViewDesc& res, int act,
                     int res_pre, half_t* xhat, hipStream_t st) {
  CVX_TRY(check_c(C, M));
  CVX_TRY(check_act(act, res_pre, res.p != nullptr));
  int rows = cvx_stream_rows_per_block(M, C, 16);
  const dim3 grid(blocks_for(M, rows)), block(256);
  const size_t lds = fold_ws_bytes(C) + 2 * C * 4;
#define CVX_LAUNCH_APPLY(A, R) hipLaunchKernelGGL((bn_act_apply_kernel<A, R>), grid, block, lds, st, y, M, C, hw, a, out, res, xhat, rows)
  if (act == 0 && res_pre)
    CVX_LAUNCH_APPLY(0, true);
  else if (act == 0)
    CVX_LAUNCH_APPLY(0, false);
  else if (act == 1 && res_pre)
    CVX_LAUNCH_APPLY(1, true);
  else if (act == 1)
    CVX_LAUNCH_APPLY(1, false);
  else if (res_pre)
    CVX_LAUNCH_APPLY(2, true);
  else
    CVX_LAUNCH_APPLY(2, false);
#undef CVX_LAUNCH_APPLY
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_bn_silu_apply(const float* y, long long M, int C, int hw, const BnTrainArgs& a, const ViewDesc& out, const ViewDesc& res,
                      half_t* xhat, hipStream_t st) {
  return cvx_bn_act_apply(y, M, C, hw, a, out, res, 0, 0, xhat, st);
}
int cvx_bn_stats_f32(const float* y, long long M, int C, long long* part, hipStream_t st) {
  CVX_TRY(check_c(C, M));
  int rows = cvx_stream_rows_per_block(M, C, 32);
  hipLaunchKernelGGL(bn_stats_f32_kernel, dim3(blocks_for(M, rows)), dim3(256), 0, st, y, M, C, part, rows);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_bn_bwd_reduce(const half_t* xhat, long long M, int C, int hw, const BnCoef& k, const ViewDesc& gout, const BnActKind& ak, long long* part,
                      hipStream_t st) {
  CVX_TRY(check_c(C, M));
  CVX_CHECK(ak.act >= 0 && ak.act <= 2, "bn_bwd: activation kind");
  CVX_CHECK(!(ak.act == 0 && ak.res_pre) || ak.fout.p, "bn_bwd: SiLU with a pre-activation residual needs the residual's forward value");
  int rows = cvx_stream_rows_per_block(M, C, 32);
  const dim3 grid(blocks_for(M, rows)), block(256);
  if (ak.act == 0 && ak.res_pre)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<3>, grid, block, 0, st, xhat, M, C, hw, k, gout, ak.fout, part, rows);
  else if (ak.act == 0)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<0>, grid, block, 0, st, xhat, M, C, hw, k, gout, ak.fout, part, rows);
  else if (ak.act == 1)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, grid, block, 0, st, xhat, M, C, hw, k, gout, ak.fout, part, rows);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<2>, grid, block, 0, st, xhat, M, C, hw, k, gout, ak.fout, part, rows);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_bn_bwd_apply(const half_t* xhat, long long M, int C, int hw, const BnCoef& k, const long long* part, float inv_scale, float* dgamma,
                     float* dbeta, const ViewDesc& gout, const BnActKind& ak, half_t* dy, const ViewDesc& gres, int res_accumulate, hipStream_t st) {
  CVX_TRY(check_c(C, M));
  CVX_TRY(check_act(ak.act, ak.res_pre, gres.p != nullptr));
  CVX_CHECK(!(ak.act == 0 && ak.res_pre) || ak.fout.p, "bn_bwd: SiLU with a pre-activation residual needs the residual's forward value");
  int rows = cvx_stream_rows_per_block(M, C, 32);
  const dim3 grid(blocks_for(M, rows)), block(256);
  const size_t lds = fold_ws_bytes(C);
#define CVX_LAUNCH_BWD(A, R)                                                                                                              \
  hipLaunchKernelGGL((bn_bwd_apply_kernel<A, R>), grid, block, lds, st, xhat, M, C, hw, k, part, inv_scale, dgamma, dbeta, gout, ak.fout, dy, gres, \
                     res_accumulate, rows)
  if (ak.act == 0 && ak.res_pre)
    CVX_LAUNCH_BWD(3, true);
  else if (ak.act == 0)
    CVX_LAUNCH_BWD(0, false);
  else if (ak.act == 1 && ak.res_pre)
    CVX_LAUNCH_BWD(1, true);
  else if (ak.act == 1)
    CVX_LAUNCH_BWD(1, false);
  else if (ak.res_pre)
    CVX_LAUNCH_BWD(2, true);
  else
    CVX_LAUNCH_BWD(2, false);
#undef CVX_LAUNCH_BWD
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_colsum_multi(const half_t* base, const ColsumDesc* descs, int ndesc, int max_c, const ColsumBlock* blocks, int nblocks, float inv_scale,
                     float* grads, hipStream_t st) {
  if (ndesc <= 0 || nblocks <= 0) return 0;
  CVX_TRY(check_c(max_c));
  hipLaunchKernelGGL(colsum_multi_reduce_kernel, dim3(nblocks), dim3(256), 0, st, base, descs, blocks);
  hipLaunchKernelGGL(colsum_multi_finalize_kernel, dim3(ndesc), dim3(256), fold_ws_bytes(max_c), st, descs, inv_scale, grads);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_colsum_finalize(const long long* part, int C, float inv_scale, float* dbias, hipStream_t st) {
  CVX_TRY(check_c(C));
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(1), dim3(256), fold_ws_bytes(C), st, part, C, inv_scale, dbias);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_colsum(long long M, int C, int hw, const ViewDesc& g, long long* part, float inv_scale, float* dbias, hipStream_t st) {
  CVX_TRY(check_c(C, M));
  int rows = cvx_stream_rows_per_block(M, C, 32);
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3(blocks_for(M, rows)), dim3(256), 0, st, M, C, hw, g, part, rows);
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(1), dim3(256), fold_ws_bytes(C), st, part, C, inv_scale, dbias);
  CVX_HIP(hipGetLastError());
  return 0;
}
