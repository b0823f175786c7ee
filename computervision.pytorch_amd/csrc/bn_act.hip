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
#include <mutex>
#include <type_traits>

namespace {
using namespace cvx_bn;

#ifndef CVX_BN_UNR
#define CVX_BN_UNR 4
#endif
#ifndef CVX_BN_RAW_MULT
#define CVX_BN_RAW_MULT 1
#endif
constexpr int UNR = CVX_BN_UNR;  // rows in flight per thread in the streaming passes

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
// RAW16: y is the raw output rounded to fp16 (CVX_OPF_RAW_F16 layers): 2 instead of 4 bytes read per element, and no xhat is written -- the
// backward passes normalise y themselves
__device__ __forceinline__ f8 load_row(const float* p) { return load_f8(p); }
__device__ __forceinline__ h8 load_row(const half_t* p) { return *reinterpret_cast<const h8*>(p); }
template <int ACT, bool RES_PRE, bool RAW16 = false>
__global__ __launch_bounds__(256) void bn_act_apply_kernel(const typename std::conditional<RAW16, half_t, float>::type* y, long long M, int C, int hw, BnTrainArgs a, ViewDesc out,
                                                           ViewDesc res, half_t* xhat, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) long long ws[];  // fold workspace | mean, invstd (2*C floats)
  float* s_mu = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + fold_ws_bytes(C));
  float* s_is = s_mu + C;
  const int CG = C >> 3;
  const int RP = 256 / CG;
  const int cg = threadIdx.x % CG, r = threadIdx.x / CG;
  const bool active = r < RP;
  const long long m0 = (long long)blockIdx.x * rows_per_block;
  const long long m1 = min(M, m0 + rows_per_block);
  long long m = m0 + r;
  // rows in flight per thread.  A RAW16 row is 16 bytes per thread instead of 32, but more rows do not pay: CVX_BN_RAW_MULT 1 / 2 / 4 ->
  // 5.911 / 5.933 / 5.938 ms per step, same box (round 5)
  constexpr int RIF = RAW16 ? CVX_BN_RAW_MULT * UNR : UNR;
  using YV = typename std::conditional<RAW16, h8, f8>::type;  // a row as loaded: converted where it is used
  // The first trip's rows and the coefficients are requested BEFORE the statistics are folded: the fold (slab reads, LDS atomics, three
  // barriers, fp64 conversion) is 2-3 us of every block's life that the rows' memory latency now runs under.
  const bool first = active && m + (long long)(RIF - 1) * RP < m1;
  YV v0[RIF];
  h8 rr0[RIF] = {};
  float ga[8], be[8];
  if (active) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      ga[i] = a.gamma[cg * 8 + i];
      be[i] = a.beta[cg * 8 + i];
    }
  }
  if (first) {
#pragma unroll
    for (int u = 0; u < RIF; ++u) {
      v0[u] = load_row(y + (m + u * RP) * C + cg * 8);
      if (res.p) rr0[u] = *reinterpret_cast<const h8*>(res.p + view_off(res, m + u * RP, hw) + cg * 8);
    }
  }
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
  if (!active) return;
  float mu[8], is[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    mu[i] = s_mu[cg * 8 + i];
    is[i] = s_is[cg * 8 + i];
  }
  auto one = [&](long long m, const YV& v, const h8& rr) {
    float f[8];
    h8 xh;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = ((float)v[i] - mu[i]) * is[i];
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
    if constexpr (!RAW16) *reinterpret_cast<h8*>(xhat + m * C + cg * 8) = xh;
    *reinterpret_cast<h8*>(out.p + view_off(out, m, hw) + cg * 8) = o;
  };
  if (first) {
#pragma unroll
    for (int u = 0; u < RIF; ++u) one(m + u * RP, v0[u], rr0[u]);
    m += (long long)RIF * RP;
  }
  // RIF rows per trip with every load issued before the first use: the passes are latency-bound otherwise
  for (; m + (long long)(RIF - 1) * RP < m1; m += (long long)RIF * RP) {
    YV v[RIF];
    h8 rr[RIF] = {};
#pragma unroll
    for (int u = 0; u < RIF; ++u) {
      v[u] = load_row(y + (m + u * RP) * C + cg * 8);
      if (res.p) rr[u] = *reinterpret_cast<const h8*>(res.p + view_off(res, m + u * RP, hw) + cg * 8);
    }
#pragma unroll
    for (int u = 0; u < RIF; ++u) one(m + u * RP, v[u], rr[u]);
  }
  for (; m < m1; m += RP) {
    YV v = load_row(y + m * C + cg * 8);
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

template <int ACT, bool RAW = false>
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
    float mu[8], is[8];  // RAW: the kept tensor is y, xhat = (y - mean) * invstd
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s.a[i] = k.gamma[cg * 8 + i];
      s.b[i] = k.beta[cg * 8 + i];
      if constexpr (RAW) {
        mu[i] = k.mean[cg * 8 + i];
        is[i] = k.invstd[cg * 8 + i];
      }
    }
    const long long m0 = (long long)blockIdx.x * rows_per_block;
    const long long m1 = min(M, m0 + rows_per_block);
    auto one = [&](const h8& v, const h8& g, const h8& fo) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float xh = (float)v[i];
        if constexpr (RAW) xh = (xh - mu[i]) * is[i];
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
// (requesting the first trip's rows before the fold, as the forward pass does, measured SLOWER here: 64 more live registers across the fold)
template <int ACT, bool RES_PRE, bool RAW = false>
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
  float k1[8], k2[8], gi[8], mu[8], is[8];
  const double cnt = (double)M;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s.a[i] = k.gamma[cg * 8 + i];
    s.b[i] = k.beta[cg * 8 + i];
    k1[i] = (float)(s0[cg * 8 + i] / cnt);
    k2[i] = (float)(s1[cg * 8 + i] / cnt);
    is[i] = k.invstd[cg * 8 + i];
    gi[i] = s.a[i] * is[i];
    if constexpr (RAW) mu[i] = k.mean[cg * 8 + i];
  }
  const long long m0 = (long long)blockIdx.x * rows_per_block;
  const long long m1 = min(M, m0 + rows_per_block);
  auto one = [&](long long m, const h8& v, h8 g, const h8& old, const h8& fo) {
    h8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float xh = (float)v[i];
      if constexpr (RAW) xh = (xh - mu[i]) * is[i];
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

#ifdef CVX_TUNING
// ---------------------------------------------------------------------------------------------
// One-launch backward (round 4, TUNING BUILD ONLY -- measured slower, see the end of this comment): reduce + grid gate + apply.  A thread keeps its KR rows of (xhat, g) in registers across the gate, so the
// pass reads 4 and writes 2 bytes per element instead of 4 + 4 + 2, and a layer costs one launch instead of two -- on the <= 40x40 layers
// the two launches were 10-18 us each for 2-8 us of traffic.  The gate is a counter behind the layer's replica slabs (zeroed with them once
// per pass): every block adds its partial sums (integer atomics, exact), bumps the counter and waits until all `nblocks` have.  That needs
// every block of the grid resident at once: the host picks KR so that the grid has at most 512 blocks of 256 threads (two per CU) and
// the engine runs the kernel on its main stream only -- blocks of OTHER kernels that share the CUs (the weight-gradient stream) finish on
// their own, a second gated kernel beside this one could starve both.  The wait is bounded: a block that gives up poisons its output with
// NaN (the loss-scale check then skips the step) instead of hanging the queue.
// Measured (MI355X, batch 32; rocprofv3 kernel durations): ALONE on the device the gated launch takes 18.9 / 21.2 / 28.0 / 42.8 us on
// 20x20x128 / 40x40x64 / 40x40x128 / 80x80x64 against 15.1 / 16.6 / 21.7 / 38.5 us for the two passes together -- every block reaches its
// atomics at the same moment and then reads the slabs back past the L2 (device-scope loads), ~10 us that the two-pass form hides behind
// other blocks' streaming, and at KR = 16 the 229 registers leave one wave per SIMD; inside the training step (weight-gradient stream
// beside it) 6.93 against 6.62 ms.  So the release library does not carry it: CVX_BN_FUSED=1 enables it in the tuning build.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void fold_replicas_coherent(const long long* part, int C, long long* ws) {  // fold_replicas, reading at device scope
  const int nacc = C * 2 * CVX_FIX_WORDS;
  for (int i = threadIdx.x; i < nacc; i += 256) ws[i] = 0;
  __syncthreads();
  const int total = cvx_stat_replicas(C) * C * 2;
  for (int e = threadIdx.x; e < total; e += 256) {
    const long long* src = part + (long long)e * CVX_FIX_WORDS;
    const long long q0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long q1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int cv = e % (C * 2);
    atomicAdd(reinterpret_cast<unsigned long long*>(&ws[cv * 2]), (unsigned long long)q0);
    atomicAdd(reinterpret_cast<unsigned long long*>(&ws[cv * 2 + 1]), (unsigned long long)q1);
  }
  __syncthreads();
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

constexpr int kGateSpins = 1 << 22;  // x (sleep + one coherent load, ~1 us): seconds -- far beyond any wait of a healthy grid

// waves per SIMD the compiler must leave room for (it schedules for instruction-level parallelism and spends 16+ registers per row otherwise):
// sets the blocks a CU holds, i.e. the largest grid the gate admits (the host asks the runtime for the real figure)
template <int KR, bool RES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KR == 4 ? 4 : KR == 8 ? 3 : 2))) void bn_bwd_fused_kernel(const half_t* xhat, long long M, int C, int hw, BnCoef k, long long* part,
                                                           unsigned long long* gate, unsigned nblocks, float inv_scale, float* dgamma, float* dbeta,
                                                           ViewDesc gout, half_t* dy, ViewDesc gres, int res_accumulate) {
  extern __shared__ __attribute__((aligned(16))) long long ws[];  // 16 KB of block-sum scratch first, then the fold workspace
  __shared__ int s_gave_up;
  const int CG = C >> 3;
  const int RP = 256 / CG;
  const int cg = threadIdx.x % CG, r = threadIdx.x / CG;
  const bool active = r < RP;
  long long mb = (long long)blockIdx.x * RP * KR + r;
  Coef8 s;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s.a[i] = active ? k.gamma[cg * 8 + i] : 0.f;
    s.b[i] = active ? k.beta[cg * 8 + i] : 0.f;
  }
  // ---- phase 1: this thread's rows into registers (every load issued before the first use), channel sums of dz and dz * xhat ----
  h8 v[KR], g[KR];
#pragma unroll
  for (int q = 0; q < KR; ++q) {
    const long long m = mb + (long long)q * RP;
    const long long mc = m < M ? m : M - 1;  // branch-free: rows past the end re-read the last row, and their g is zeroed below
    v[q] = h8{};
    g[q] = h8{};
    if (active) {
      v[q] = *reinterpret_cast<const h8*>(xhat + mc * C + cg * 8);
      g[q] = *reinterpret_cast<const h8*>(gout.p + view_off(gout, mc, hw) + cg * 8);
    }
  }
#pragma unroll
  for (int q = 0; q < KR; ++q)
    if (mb + (long long)q * RP >= M) g[q] = h8{};
  float acc[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[0][i] = acc[1][i] = 0.f;
#pragma unroll
  for (int q = 0; q < KR; ++q) {
    __builtin_amdgcn_sched_barrier(0);  // row by row: interleaved, the rows' unpacked floats cost 16+ registers each (spills at KR = 8)
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // rows past the end hold g = 0: they add nothing
      const float xh = (float)v[q][i];
      const float dz = act_dz<0>((float)g[q][i], xh, s.a[i], s.b[i], 0.f);
      acc[0][i] += dz;
      acc[1][i] += dz * xh;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // the rows stay PACKED (8 registers per row) across the gate: without this the compiler keeps phase 1's converted floats and dz values
  // alive for phase 2 -- 16+ registers per row, which at KR = 16 halves the blocks a CU holds and with them the largest grid the gate admits
  // (the sums as inputs tie the statement BEHIND phase 1: moved in front of it -- where the compiler put it at first -- the laundered rows
  // are a second copy beside the ones phase 1 still reads)
#pragma unroll
  for (int q = 0; q < KR; ++q) asm volatile("" : "+v"(v[q]), "+v"(g[q]) : "v"(acc[0][q & 7]), "v"(acc[1][q & 7]));
  asm volatile("" : "+v"(mb));  // ... and phase 2 works its addresses out again: kept from phase 1 they are 4 more registers per row
  if (threadIdx.x == 0) s_gave_up = 0;
  block_channel_sums<2>(acc, C, CG, cg, active, reinterpret_cast<float*>(ws), part, blockIdx.x);
  // ---- the gate ----
  // Everything blocks exchange goes through device-scope ATOMICS (the partial sums: integer RMWs; the counter; the slabs are read back
  // with device-scope loads), which are performed at the coherence point, so no cache maintenance is needed -- and none must be used: the
  // first version's __threadfence() / acquire loads compiled to buffer_wbl2 / buffer_inv sc1 per thread and per poll, i.e. an L2
  // write-back or invalidate of the whole XCD each, and ran 2-4x SLOWER than the two-launch path while slowing the weight-gradient
  // stream's kernels by 1.7x.  What remains is order: a wave's atomics have been performed once its vmcnt is back at zero.
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(gate, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)nblocks) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > kGateSpins) {
        s_gave_up = 1;
        break;
      }
    }
  }
  __syncthreads();
  const bool gave_up = s_gave_up != 0;
  fold_replicas_coherent(part, C, ws);
  const double* s0 = reinterpret_cast<const double*>(ws);
  const double* s1 = s0 + C;
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      dgamma[c] += gave_up ? __builtin_nanf("") : (float)(s1[c] * inv_scale);
      dbeta[c] += (float)(s0[c] * inv_scale);
    }
  }
  if (!active) return;
  // ---- phase 2: dy = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)) from the registers ----
  float k1[8], k2[8], gi[8];
  const double cnt = (double)M;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    k1[i] = gave_up ? __builtin_nanf("") : (float)(s0[cg * 8 + i] / cnt);
    k2[i] = (float)(s1[cg * 8 + i] / cnt);
    gi[i] = s.a[i] * k.invstd[cg * 8 + i];
  }
  constexpr int QB = 4;  // rows per trip: the residual's old values are loaded QB rows ahead of their use
#pragma unroll
  for (int q0 = 0; q0 < KR; q0 += QB) {
    h8 old[QB];
    if constexpr (RES) {
#pragma unroll
      for (int j = 0; j < QB; ++j) {
        const long long m = mb + (long long)(q0 + j) * RP;
        old[j] = h8{};
        if (res_accumulate && m < M) old[j] = *reinterpret_cast<const h8*>(gres.p + view_off(gres, m, hw) + cg * 8);
      }
    }
#pragma unroll
    for (int j = 0; j < QB; ++j) {
      const int q = q0 + j;
      const long long m = mb + (long long)q * RP;
      if (m >= M) return;
      __builtin_amdgcn_sched_barrier(0);
      h8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (float)v[q][i];
        const float dz = act_dz<0>((float)g[q][i], xh, s.a[i], s.b[i], 0.f);
        o[i] = (half_t)(gi[i] * (dz - k1[i] - xh * k2[i]));
      }
      *reinterpret_cast<h8*>(dy + m * C + cg * 8) = o;
      if constexpr (RES) {
        h8 gg = g[q];
        if (res_accumulate) {
#pragma unroll
          for (int i = 0; i < 8; ++i) gg[i] = (half_t)((float)gg[i] + (float)old[j][i]);
        }
        *reinterpret_cast<h8*>(gres.p + view_off(gres, m, hw) + cg * 8) = gg;
      }
    }
  }
}

#endif  // CVX_TUNING

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
  // big tensors: no more than 512 blocks (in the step, same box: cap 0 / 256 / 384 / 512 / 768 / 1024 -> 6.54 / 6.45 / 6.43 / 6.41 / 6.47 / 6.52 ms).
  // Measured alone on the device (rocprofv3, 13.1 M elements, tools/sweeps/prof_bn_kb.sh): 800 blocks of 32 KB 20.9 / 19.0 us (reduce /
  // apply), 400 of 64 KB 14.5 / 16.3, 200 of 128 KB 16.2 / 17.0 -- every block pays its coefficient loads, slab fold and atomics once,
  // whatever it streams; the small layers (<= 200 blocks at 32 KB) keep their size.
  static const int max_blocks = cvx_tune_int("CVX_BN_BLOCKS", 512);
  if (kb_env <= 0 && max_blocks > 0) {
    const long long total_kb = (M * C * 2) >> 10;
    if (total_kb / max_blocks > kb_per_block) kb_per_block = (int)(total_kb / max_blocks);
  }
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
__global__ __launch_bounds__(256) void raw16_to_xhat_kernel(const half_t* y, long long n, int C, const float* mean, const float* invstd, half_t* xhat) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const int c = (int)(i % C);
    xhat[i] = (half_t)(((float)y[i] - mean[c]) * invstd[c]);
  }
}
__global__ __launch_bounds__(256) void kept_to_xhat_f32_kernel(const half_t* y, long long n, int C, const float* mean, const float* invstd, float* xhat) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const int c = (int)(i % C);
    xhat[i] = mean ? ((float)y[i] - mean[c]) * invstd[c] : (float)y[i];
  }
}
int cvx_kept_to_xhat_f32(const half_t* kept, long long M, int C, const float* mean, const float* invstd, float* xhat, hipStream_t st) {
  const long long n = M * C;
  hipLaunchKernelGGL(kept_to_xhat_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, kept, n, C, mean, invstd, xhat);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_raw16_to_xhat(const half_t* y16, long long M, int C, const float* mean, const float* invstd, half_t* xhat, hipStream_t st) {
  const long long n = M * C;
  hipLaunchKernelGGL(raw16_to_xhat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y16, n, C, mean, invstd, xhat);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_bn_silu_apply_raw16(const half_t* y16, long long M, int C, int hw, const BnTrainArgs& a, const ViewDesc& out, hipStream_t st) {
  CVX_TRY(check_c(C, M));
  int rows = cvx_stream_rows_per_block(M, C, 16);
  const dim3 grid(blocks_for(M, rows)), block(256);
  const size_t lds = fold_ws_bytes(C) + 2 * (size_t)C * 4;
  hipLaunchKernelGGL((bn_act_apply_kernel<0, false, true>), grid, block, lds, st, y16, M, C, hw, a, out, ViewDesc{nullptr, 0, 0}, (half_t*)nullptr, rows);
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
  CVX_CHECK(!k.mean || (ak.act == 0 && !ak.res_pre), "bn_bwd: a raw-fp16 layer (BnCoef::mean) is SiLU without a pre-activation residual");
  int rows = cvx_stream_rows_per_block(M, C, 32);
  const dim3 grid(blocks_for(M, rows)), block(256);
  if (ak.act == 0 && ak.res_pre)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<3>, grid, block, 0, st, xhat, M, C, hw, k, gout, ak.fout, part, rows);
  else if (ak.act == 0 && k.mean)
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<0, true>), grid, block, 0, st, xhat, M, C, hw, k, gout, ak.fout, part, rows);
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
  CVX_CHECK(!k.mean || (ak.act == 0 && !ak.res_pre), "bn_bwd: a raw-fp16 layer (BnCoef::mean) is SiLU without a pre-activation residual");
  int rows = cvx_stream_rows_per_block(M, C, 32);
  const dim3 grid(blocks_for(M, rows)), block(256);
  const size_t lds = fold_ws_bytes(C);
#define CVX_LAUNCH_BWD(A, R)                                                                                                              \
  hipLaunchKernelGGL((bn_bwd_apply_kernel<A, R>), grid, block, lds, st, xhat, M, C, hw, k, part, inv_scale, dgamma, dbeta, gout, ak.fout, dy, gres, \
                     res_accumulate, rows)
  if (ak.act == 0 && ak.res_pre)
    CVX_LAUNCH_BWD(3, true);
  else if (ak.act == 0 && k.mean)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<0, false, true>), grid, block, lds, st, xhat, M, C, hw, k, part, inv_scale, dgamma, dbeta, gout, ak.fout, dy, gres,
                       res_accumulate, rows);
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
// The one-launch form.  Returns 1 when the layer does not qualify (the caller then takes cvx_bn_bwd_reduce + cvx_bn_bwd_apply): SiLU
// without a pre-activation residual only, at most 512 blocks at 16 rows per thread.  `gate`: one 64-bit counter, zero on entry.
int cvx_bn_bwd_fused(const half_t* xhat, long long M, int C, int hw, const BnCoef& k, long long* part, unsigned long long* gate, float inv_scale,
                     float* dgamma, float* dbeta, const ViewDesc& gout, const BnActKind& ak, half_t* dy, const ViewDesc& gres, int res_accumulate,
                     hipStream_t st) {
#ifndef CVX_TUNING
  return 1;  // the gated kernel measured slower than the two passes (see its comment): tuning build only
#else
  static const bool on = cvx_tune_int("CVX_BN_FUSED", 0) != 0;
  if (!on || ak.act != 0 || ak.res_pre || C > 512 || C % 8 != 0 || !gate) return 1;
  CVX_TRY(check_c(C, M));
  static const int max_g = cvx_tune_int("CVX_BN_FUSED_G", 256), max_g16 = cvx_tune_int("CVX_BN_FUSED_G16", 512);
  const size_t lds = std::max<size_t>(256 * 16 * 4, fold_ws_bytes(C));
  const bool res = gres.p != nullptr;
  // blocks of each variant the device holds at once, from the runtime (registers as compiled, 16 KB of LDS): the gate's hard limit
  static int resident[3][2] = {{0, 0}, {0, 0}, {0, 0}};
  static std::mutex mu;
  auto capacity = [&](int idx, const void* fn) -> int {
    std::lock_guard<std::mutex> lock(mu);
    if (resident[idx][res] == 0) {
      int per_cu = 0, dev = 0;
      hipDeviceProp_t prop;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 16384) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
          hipGetDeviceProperties(&prop, dev) != hipSuccess)
        resident[idx][res] = -1;
      else
        resident[idx][res] = per_cu * prop.multiProcessorCount;
    }
    return resident[idx][res];
  };
  const int RP = 256 / (C / 8);
  int KR = 0;
  long long G = 0;
  int idx = 0;
  for (int kr : {4, 8, 16}) {
    G = (M + (long long)RP * kr - 1) / ((long long)RP * kr);
    const void* fn = kr == 4 ? (res ? (const void*)bn_bwd_fused_kernel<4, true> : (const void*)bn_bwd_fused_kernel<4, false>)
                   : kr == 8 ? (res ? (const void*)bn_bwd_fused_kernel<8, true> : (const void*)bn_bwd_fused_kernel<8, false>)
                             : (res ? (const void*)bn_bwd_fused_kernel<16, true> : (const void*)bn_bwd_fused_kernel<16, false>);
    if (G <= (kr == 16 ? max_g16 : max_g) && G <= capacity(idx, fn)) {
      KR = kr;
      break;
    }
    ++idx;
  }
  if (!KR) return 1;
  const dim3 grid((unsigned)G), block(256);
#define CVX_LAUNCH_FUSED(K_, R_)                                                                                                              \
  hipLaunchKernelGGL((bn_bwd_fused_kernel<K_, R_>), grid, block, lds, st, xhat, M, C, hw, k, part, gate, (unsigned)G, inv_scale, dgamma, dbeta, gout, \
                     dy, gres, res_accumulate)
  if (KR == 4) {
    if (res) CVX_LAUNCH_FUSED(4, true); else CVX_LAUNCH_FUSED(4, false);
  } else if (KR == 8) {
    if (res) CVX_LAUNCH_FUSED(8, true); else CVX_LAUNCH_FUSED(8, false);
  } else {
    if (res) CVX_LAUNCH_FUSED(16, true); else CVX_LAUNCH_FUSED(16, false);
  }
#undef CVX_LAUNCH_FUSED
  CVX_HIP(hipGetLastError());
  return 0;
#endif
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
