// The stem convolution (3 -> Cout, 3x3, stride 2, pad 1: model.0 of every YOLOv8 scale) in FP32, straight from the
// caller's NCHW fp32 images (gfx950).
//
// Why a kernel of its own: K = 27 wastes an MFMA K-step, the layer is purely HBM-bound, and its arithmetic decides the
// forward error of the whole network -- the image has a large mean, so rounding the image, the stem weights or the raw
// stem output to fp16 costs more than all later layers together.  Here nothing is rounded before
// the BatchNorm: images and master weights are read as fp32, 27 FMAs per output value on the vector ALUs
// (432 per pixel at 16 channels: ~3 GFLOP per 32-image batch, far below the memory time), and training mode runs TWO
// passes over the images -- statistics, then recompute + normalise + SiLU -- instead of storing the raw output:
//   pass 1  stem_stats   images (157 MB at batch 32)                      -> per-channel (sum, sumsq), fixed-point slabs
//   pass 2  stem_apply   images again                                     -> xhat fp16 (backward operand) + activation fp16
// i.e. 0.52 GB where NCHW->NHWC8 conversion + MFMA conv + BN/SiLU pass moved 0.89 GB.  Eval mode is pass 2 alone with the
// folded running statistics.  The weight gradient (stem_wgrad) reads the fp32 images too: 4 x 27 accumulators per lane,
// lane = pixel, wave = 4 output channels, fixed-order reductions into the engine's fp32 gradient slabs.
#include "stem.h"

#include "bn_common.h"

namespace {
using namespace cvx_bn;

constexpr int KT = 27;  // (kh*3 + kw)*3 + ci

struct PixelId {
  int b, oy, ox;
};
__device__ __forceinline__ PixelId pixel_of(const StemParams& p, long long m) {
  const unsigned mu = (unsigned)m;  // M < 2^31 (launcher)
  const unsigned t = mu / (unsigned)p.OW;
  const int ox = (int)(mu - t * (unsigned)p.OW);
  const int b = (int)(t / (unsigned)p.OH);
  const int oy = (int)(t - (unsigned)b * (unsigned)p.OH);
  return PixelId{b, oy, ox};
}

// the 3x3x3 input window of output pixel (oy, ox): columns 2ox-1, 2ox, 2ox+1 of rows 2oy-1 .. 2oy+1 (H, W even: only
// the top row and the left column can fall outside).  One 8-byte load covers (2ox, 2ox+1); consecutive lanes read
// consecutive 8-byte pairs, the left neighbour is a second (cache-resident) 4-byte load.
__device__ __forceinline__ void load_window(const StemParams& p, const PixelId& id, float (&x)[KT]) {
  const long long plane = (long long)p.H * p.W;
  const float* img = p.img + (long long)id.b * 3 * plane;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int iy = 2 * id.oy - 1 + kh;
    const bool rok = iy >= 0;  // iy <= 2*OH - 1 + 1 - 1 = H - 1 always
    const int iyc = rok ? iy : 0;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
      const float* row = img + ci * plane + (long long)iyc * p.W + 2 * id.ox;
      const float2 c = *reinterpret_cast<const float2*>(row);
      const float l = id.ox > 0 ? row[-1] : 0.f;
      x[(kh * 3 + 0) * 3 + ci] = rok ? l : 0.f;
      x[(kh * 3 + 1) * 3 + ci] = rok ? c.x : 0.f;
      x[(kh * 3 + 2) * 3 + ci] = rok ? c.y : 0.f;
    }
  }
}

// weights fp32 [Cout][kh][kw][ci] (the master layout) -> LDS [27][Cout]: a lane group reads 16 consecutive channels of
// one k as four broadcast ds_read_b128
__device__ __forceinline__ void stage_weights(const StemParams& p, float* sW) {
  for (int i = threadIdx.x; i < KT * p.Cout; i += blockDim.x) {
    const int co = i / KT, k = i - co * KT;
    sW[k * p.Cout + co] = p.w[i];
  }
}

// 16 output channels [g*16, g*16+16) of one pixel; the same instruction order in every pass (bit-identical recompute)
__device__ __forceinline__ void conv16(const float (&x)[KT], const float* sW, int Cout, int g, float (&acc)[16]) {
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.f;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const f4* w4 = reinterpret_cast<const f4*>(sW + k * Cout + g * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f4 w = w4[q];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q * 4 + r] = fmaf(x[k], w[r], acc[q * 4 + r]);
    }
  }
}

// ---- pass 1: batch statistics ------------------------------------------------------------------------------
template <int NG>
__global__ __launch_bounds__(256) void stem_stats_kernel(const StemParams p, long long M, long long* stats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW = smem;                 // [27][Cout]
  float* sred = smem + KT * p.Cout;  // [4 waves][2][Cout]
  stage_weights(p, sW);
  __syncthreads();
  float s1[NG][16], s2[NG][16];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int c = 0; c < 16; ++c) s1[g][c] = s2[g][c] = 0.f;
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
    asm volatile("" ::: "memory");  // keeps the compiler from hoisting all 27 x Cout LDS weights into registers (spills)
    const PixelId id = pixel_of(p, m);
    float x[KT];
    load_window(p, id, x);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float acc[16];
      conv16(x, sW, p.Cout, g, acc);
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        s1[g][c] += acc[c];
        s2[g][c] = fmaf(acc[c], acc[c], s2[g][c]);
      }
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float a = cvx_wave_sum64(s1[g][c]), b = cvx_wave_sum64(s2[g][c]);
      if (lane == 0) {
        sred[(wave * 2 + 0) * p.Cout + g * 16 + c] = a;
        sred[(wave * 2 + 1) * p.Cout + g * 16 + c] = b;
      }
    }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * p.Cout; t += 256) {
    const int which = t / p.Cout, c = t - which * p.Cout;
    const float v = (sred[(0 * 2 + which) * p.Cout + c] + sred[(1 * 2 + which) * p.Cout + c]) +
                    (sred[(2 * 2 + which) * p.Cout + c] + sred[(3 * 2 + which) * p.Cout + c]);
    cvx_fix_atomic_add(stats, ((long long)(blockIdx.x % cvx_stat_replicas(p.Cout)) * p.Cout + c) * 2 + which, v);
  }
}

// ---- pass 2: recompute, normalise from the fp32 values, SiLU, store ---------------------------------------------
template <int NG, bool TRAIN>
__global__ __launch_bounds__(256) void stem_apply_kernel(const StemParams p, long long M, BnTrainArgs a, const float* scale,
                                                         const float* shift, ViewDesc out, half_t* xhat) {
  extern __shared__ __attribute__((aligned(16))) long long ws[];  // [fold workspace (train)] | 4 x Cout coefficients | weights
  float* sA = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + (TRAIN ? fold_ws_bytes(p.Cout) : 0));
  float* sB = sA + p.Cout;   // train: (mean, invstd, gamma, beta); eval: (scale, shift, -, -)
  float* sG = sB + p.Cout;
  float* sBe = sG + p.Cout;
  float* sW = sBe + p.Cout;
  if (TRAIN) {
    fold_replicas(a.stats, p.Cout, ws);
    for (int c = threadIdx.x; c < p.Cout; c += 256) {
      double var;
      const BnMoments mo = moments_of(ws, p.Cout, c, M, a.eps, &var);
      sA[c] = mo.mean;
      sB[c] = mo.invstd;
      sG[c] = a.gamma[c];
      sBe[c] = a.beta[c];
      if (blockIdx.x == 0) {
        const double cnt = (double)M;
        const double mu = reinterpret_cast<const double*>(ws)[c] / cnt;
        a.mean[c] = mo.mean;
        a.invstd[c] = mo.invstd;
        const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
        a.rmean[c] = (float)((1.0 - a.momentum) * (double)a.rmean[c] + a.momentum * mu);
        a.rvar[c] = (float)((1.0 - a.momentum) * (double)a.rvar[c] + a.momentum * unbiased);
      }
    }
  } else {
    for (int c = threadIdx.x; c < p.Cout; c += 256) {
      sA[c] = scale[c];
      sB[c] = shift[c];
    }
  }
  stage_weights(p, sW);
  __syncthreads();
  const int ohw = p.OH * p.OW;
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
    asm volatile("" ::: "memory");  // keeps the compiler from hoisting all 27 x Cout LDS weights into registers (spills)
    const PixelId id = pixel_of(p, m);
    float x[KT];
    load_window(p, id, x);
    half_t* o = out.p + (long long)id.b * out.bstride + ((long long)id.oy * p.OW + id.ox) * out.ld;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float acc[16];
      conv16(x, sW, p.Cout, g, acc);
      h8 av[2], xv[2];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int ch = g * 16 + c;
        float z;
        if (TRAIN) {
          const float xh = (acc[c] - sA[ch]) * sB[ch];
          xv[c >> 3][c & 7] = (half_t)xh;
          z = fmaf(xh, sG[ch], sBe[ch]);
        } else {
          z = fmaf(acc[c], sA[ch], sB[ch]);
        }
        av[c >> 3][c & 7] = (half_t)cvx_silu(z);
      }
      *reinterpret_cast<h8*>(o + g * 16) = av[0];
      *reinterpret_cast<h8*>(o + g * 16 + 8) = av[1];
      if (TRAIN && xhat) {  // (nullptr: the backward pass recomputes it, cvx_stem_keeps_xhat)
        half_t* xo = xhat + m * p.Cout + g * 16;
        *reinterpret_cast<h8*>(xo) = xv[0];
        *reinterpret_cast<h8*>(xo + 8) = xv[1];
      }
    }
  }
  (void)ohw;
}

// ---- weight gradient: dW[co][tap][ci] = sum over pixels of dy[pixel][co] * x[pixel][tap][ci] -------------------------
// blockIdx.y = slice of 16 output channels; wave w of a workgroup owns channels slice*16 + 4w .. +3; all four waves walk
// the same pixels (lane = pixel).  Slab layout = the engine's [split][Cout][9 taps * 16 (padded ci)] fp32.
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const StemParams p, long long M, const half_t* dy, float* slabs) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int co0 = blockIdx.y * 16 + wave * 4;
  float acc[4][KT];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < KT; ++k) acc[c][k] = 0.f;
  // the loop is latency-bound (19 loads, then 108 FMAs per pixel): WG_UNR pixels per lane are in flight per trip
  constexpr int WG_UNR = 3;
  const long long stride = (long long)gridDim.x * 64;
  for (long long m0 = (long long)blockIdx.x * 64 + lane; m0 < M; m0 += stride * WG_UNR) {
    asm volatile("" ::: "memory");
    float x[WG_UNR][KT];
    h4 g[WG_UNR];
#pragma unroll
    for (int u = 0; u < WG_UNR; ++u) {
      const long long m = m0 + u * stride;
      const bool ok = m < M;
      const long long mm = ok ? m : m0;  // clamped address, zero gradient: contributes nothing
      load_window(p, pixel_of(p, mm), x[u]);
      g[u] = *reinterpret_cast<const h4*>(dy + mm * p.Cout + co0);
      if (!ok) g[u] = h4{(half_t)0, (half_t)0, (half_t)0, (half_t)0};
    }
#pragma unroll
    for (int u = 0; u < WG_UNR; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float gc = (float)g[u][c];
#pragma unroll
        for (int k = 0; k < KT; ++k) acc[c][k] = fmaf(gc, x[u][k], acc[c][k]);
      }
  }
  float* slab = slabs + (long long)blockIdx.x * p.Cout * 144;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const float v = cvx_wave_sum64(acc[c][k]);
      if (lane == 0) slab[(co0 + c) * 144 + (k / 3) * 16 + (k % 3)] = v;
    }
}

// ---- fused backward of the stem: BatchNorm backward "apply" + weight gradient in one pass ------------------------------
// The stem's dy (gradient w.r.t. its raw conv output) has exactly one consumer, the weight gradient (there is no data
// gradient into the image), and both kernels sit at the very end of the backward pass where nothing overlaps them.  Here
// dy = gamma * invstd * (dz - c1 - xhat * c2), dz = gout * silu'(gamma * xhat + beta), is formed in registers from xhat and
// gout and goes straight into the 4 x 27 accumulators: dy is never written or re-read (0.21 GB at batch 32) and the step's
// tail is one kernel shorter.  c1 / c2 come from the replica slabs bn_bwd_reduce filled (folded by every block, as in
// bn_bwd_apply); block (0, 0) accumulates dgamma / dbeta.
__global__ __launch_bounds__(256) void stem_bwd_kernel(const StemParams p, long long M, const half_t* xhat, ViewDesc gout, BnCoef k,
                                                       const long long* part, float inv_scale, float* dgamma, float* dbeta, float* slabs,
                                                       int tiles_x, int tiles_y, int total_tiles) {
  // Workgroup = one 8 x 32 tile of output pixels at a time (grid-stride over the tiles): the 17 x 65 x 3 fp32 input window of
  // the tile is staged in LDS ONCE (coalesced loads; even / odd columns kept apart so that the stride-2 window reads are
  // conflict-free) and read by all four waves (4 output channels each) -- the first version let every wave fetch its pixels'
  // windows from global memory itself: 4 x redundant L1 traffic, 1.4 TB/s.
  constexpr int TH = 8, TW = 32, WR = 2 * TH + 1, WE = TW + 1, WO = TW;  // window rows; even-index / odd-index columns per row
  constexpr int ROWP = WE + WO + 1;                                      // floats per (channel, row): [even | odd | pad]
  extern __shared__ __attribute__((aligned(16))) long long ws[];  // fold workspace | 5 x Cout coefficients | window
  float* sG = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + fold_ws_bytes(p.Cout));
  float* sB = sG + p.Cout;
  float* sK1 = sB + p.Cout;
  float* sK2 = sK1 + p.Cout;
  float* sGi = sK2 + p.Cout;
  float* sX = sGi + p.Cout;  // [3][WR][ROWP]
  fold_replicas(part, p.Cout, ws);
  {
    const double* s0 = reinterpret_cast<const double*>(ws);
    const double* s1 = s0 + p.Cout;
    const double cnt = (double)M;
    for (int c = threadIdx.x; c < p.Cout; c += 256) {
      const float g = k.gamma[c];
      sG[c] = g;
      sB[c] = k.beta[c];
      sK1[c] = (float)(s0[c] / cnt);
      sK2[c] = (float)(s1[c] / cnt);
      sGi[c] = g * k.invstd[c];
      if (blockIdx.x == 0 && blockIdx.y == 0) {
        dgamma[c] += (float)(s1[c] * inv_scale);
        dbeta[c] += (float)(s0[c] * inv_scale);
      }
    }
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int co0 = blockIdx.y * 16 + wave * 4;
  float ga[4], be[4], k1[4], k2[4], gi[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    ga[c] = sG[co0 + c];
    be[c] = sB[co0 + c];
    k1[c] = sK1[co0 + c];
    k2[c] = sK2[co0 + c];
    gi[c] = sGi[co0 + c];
  }
  float acc[4][KT];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) acc[c][kk] = 0.f;
  const long long plane = (long long)p.H * p.W;
  const int lx = lane & 31, ly = lane >> 5;  // a wave covers two tile rows per group
  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const int tx = tile % tiles_x;
    const int t2 = tile / tiles_x;
    const int ty = t2 % tiles_y;
    const int b = t2 / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    // xhat / gout of the wave's 4 channels, one pixel group (2 rows x 32 columns) at a time, the next group's in flight
    auto load_group = [&](int q, h4& xq, h4& gq, bool& okq) {
      const int oy = oy0 + q * 2 + ly, ox = ox0 + lx;
      okq = oy < p.OH && ox < p.OW;
      const long long pix = okq ? (long long)oy * p.OW + ox : 0;
      xq = *reinterpret_cast<const h4*>(xhat + ((long long)b * p.OH * p.OW + pix) * p.Cout + co0);
      gq = *reinterpret_cast<const h4*>(gout.p + (long long)b * gout.bstride + pix * gout.ld + co0);
    };
    h4 xn, gn;
    bool okn;
    load_group(0, xn, gn, okn);
    // stage the input window: rows 2*oy0-1 .. 2*oy0+2*TH-1, columns 2*ox0-1 .. 2*ox0+2*TW-1 (window index j = col - (2*ox0-1))
    const float* img = p.img + (long long)b * 3 * plane;
    __syncthreads();  // the previous tile's window is no longer read
    {
      constexpr int NE = 3 * WR * (WE + WO), NST = (NE + 255) / 256;
      float v[NST];
      int dst[NST];
#pragma unroll
      for (int it = 0; it < NST; ++it) {  // every load is issued before the first LDS store waits for one
        const int e = it * 256 + threadIdx.x;
        const int j = e % (WE + WO);
        const int t3 = e / (WE + WO);
        const int r = t3 % WR, ci = t3 / WR;
        const int iy = 2 * oy0 - 1 + r, ix = 2 * ox0 - 1 + j;
        const bool in = e < NE && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        v[it] = in ? img[ci * plane + (long long)iy * p.W + ix] : 0.f;
        dst[it] = e < NE ? (ci * WR + r) * ROWP + ((j & 1) ? WE + (j >> 1) : (j >> 1)) : -1;
      }
#pragma unroll
      for (int it = 0; it < NST; ++it)
        if (dst[it] >= 0) sX[dst[it]] = v[it];
    }
    __syncthreads();
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      const h4 xq = xn, gq = gn;
      const bool okq = okn;
      if (q < 3) load_group(q + 1, xn, gn, okn);
      float dyc[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float xv = (float)xq[c];
        const float dz = (float)gq[c] * cvx_silu_grad(xv * ga[c] + be[c]);
        dyc[c] = okq ? (float)(half_t)(gi[c] * (dz - k1[c] - xv * k2[c])) : 0.f;  // rounded to fp16 once, as bn_bwd_apply stores it
      }
      const float* rowq = sX + (2 * (q * 2 + ly)) * ROWP;  // window row of kh = 0
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
          const float* row = rowq + (ci * WR + kh) * ROWP;
          // window columns j = 2*lx + kw: kw = 0 -> even[lx], kw = 1 -> odd[lx], kw = 2 -> even[lx + 1]
          const float x0 = row[lx], x1 = row[WE + lx], x2 = row[lx + 1];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            acc[c][(kh * 3 + 0) * 3 + ci] = fmaf(dyc[c], x0, acc[c][(kh * 3 + 0) * 3 + ci]);
            acc[c][(kh * 3 + 1) * 3 + ci] = fmaf(dyc[c], x1, acc[c][(kh * 3 + 1) * 3 + ci]);
            acc[c][(kh * 3 + 2) * 3 + ci] = fmaf(dyc[c], x2, acc[c][(kh * 3 + 2) * 3 + ci]);
          }
        }
    }
  }
  float* slab = slabs + (long long)blockIdx.x * p.Cout * 144;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
      const float v = cvx_wave_sum64(acc[c][kk]);
      if (lane == 0) slab[(co0 + c) * 144 + (kk / 3) * 16 + (kk % 3)] = v;
    }
}

// ---- the same fused backward on the matrix cores (round 5) ---------------------------------------------------------------
// dW[co][tap] = sum over pixels of dy[co][pixel] * window[pixel][tap] is a [16 x pixels] . [pixels x 27] product: v_mfma_f32_16x16x4_f32 takes four
// pixels per instruction (fp32 operands, products and sums: the arithmetic of the VALU kernel above in another order), two N tiles
// cover the 27 taps.  What the VALU kernel spent its time on is gone: 108 FMAs per pixel and wave, window reads once per wave (here one
// ds_read_b32 per lane and N tile), 8-byte strided loads of xhat / gout (here the tile's rows come in by LDS-DMA, 1 KiB per instruction,
// the next tile's while this one is being multiplied).
//   workgroup = 4 waves, one 8 x 32 tile of output pixels at a time (grid-stride), wave w owns tile rows 2w, 2w + 1 = 16 steps of 4 pixels;
//   lane (g = lane / 16, m = lane % 16): A operand = dy of channel m at the step's pixel g (formed in registers from xhat / gout as the
//   VALU kernel forms it, rounded to fp16 once), B operand = the window value of tap n = m (N tile 0) / 16 + m (N tile 1) at that pixel;
//   LDS per stage: window [3][17 rows][72] fp32 from column 2*ox0 - 4 on (16-byte chunks: whole chunks are in or out of the image; out
//   of range = zero fill by the buffer bounds check), xhat and gout tiles [8][32][16 channels] fp16.  Two stages.
constexpr int SM_TH = 8, SM_TW = 32, SM_WR = 2 * SM_TH + 1, SM_P = 72, SM_PL = SM_WR * SM_P + 16;  // row pitch / plane pitch (floats)
constexpr int SM_WIN_CHUNKS = 16 * 64;                                                             // 16 DMA instructions of 64 chunks (3 * SM_PL / 4 = 930 used)
constexpr int SM_STAGE_BYTES = SM_WIN_CHUNKS * 16 + 2 * SM_TH * SM_TW * 32;
static_assert(3 * SM_PL / 4 <= SM_WIN_CHUNKS, "window chunks");

typedef __attribute__((address_space(3))) void* stem_lds_ptr;

// ONEPASS: the BatchNorm-backward sums are not an input (`part`) but an OUTPUT.  dy = gi (dz - mean(dz) - xhat mean(dz xhat)) is linear in the
// two means, so  dW = gi (S1 - mean(dz) S2 - mean(dz xhat) S3)  with  S1 = sum dz x,  S2 = sum x,  S3 = sum xhat x  over the pixels: the
// workgroup accumulates S1 and S3 (four MFMAs per step instead of two: dz and xhat as two A operands), S2 and the two sums on the vector
// ALUs, and stores ONE partial block [S1 16 x 32 | S3 16 x 32 | S2 32 | sum dz 16 | sum dz xhat 16] (SM_PART floats) in place of its slab;
// stem_onepass_fold_kernel combines the workgroups' blocks in fp64.  The separate bn_bwd_reduce pass over gout and xhat (0.21 GB, 45-70 us
// in the step's tail) is gone; dy is not rounded to fp16 on the way (it was, "as bn_bwd_apply stores it": one rounding less).
constexpr int SM_PART = 16 * 32 * 2 + 32 + 16 + 16;

// RECOMP (with ONEPASS): xhat is not read either -- the tile's input window is in LDS anyway, so the conv output of a 16-pixel block is
// recomputed as a [16 pixels x 28 taps] . [28 x 16 channels] product (seven more MFMAs per 16 pixels: lane (g, m) ends with pixels 4 g + i,
// channel m, exactly the operand layout of the four steps that follow), normalised with the forward's published mean / invstd and rounded to
// fp16 as the forward pass would have stored it.  The forward then does not store xhat at all: 0.1 GB less written there, 0.1 GB less read
// here, in the step's tail.
template <bool ONEPASS, bool RECOMP = false>
__global__ __launch_bounds__(256) void stem_bwd_mfma_kernel(const StemParams p, long long M, const half_t* xhat, ViewDesc gout, BnCoef k,
                                                            const long long* part, float inv_scale, float* dgamma, float* dbeta, float* slabs,
                                                            int tiles_x, int tiles_y, int total_tiles) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) long long ws[];  // fold workspace | 5 x Cout coefficients | 2 stages
  float* sG = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + fold_ws_bytes(p.Cout));
  float* sB = sG + p.Cout;
  float* sK1 = sB + p.Cout;
  float* sK2 = sK1 + p.Cout;
  float* sGi = sK2 + p.Cout;
  unsigned char* stage0 = reinterpret_cast<unsigned char*>(reinterpret_cast<uintptr_t>(sGi + p.Cout + 3) & ~(uintptr_t)15);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  const int co0 = blockIdx.y * 16;
  const long long plane = (long long)p.H * p.W;

  const __amdgpu_buffer_rsrc_t rs_img = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.img), (short)0, (int)(unsigned)((long long)p.B * 3 * plane * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_xh = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(xhat), (short)0, (int)(unsigned)(M * p.Cout * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(gout.p, (short)0, (int)(unsigned)((long long)p.B * gout.bstride * 2), 0x00020000);

  // per-lane DMA tables: window instruction q = 4 * j + wave (j = 0..3) -> chunk q * 64 + lane of the stage's window area
  int w_rel[4], w_r[4], w_j[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int chunk = (4 * j + wave) * 64 + lane;
    const int ci = chunk / (SM_PL / 4), rem = chunk - ci * (SM_PL / 4);
    const int r = rem / (SM_P / 4), cj = rem - r * (SM_P / 4);
    const bool used = ci < 3 && r < SM_WR && cj < 17;
    w_r[j] = used ? r : -100000;  // (never in range)
    w_j[j] = cj;
    w_rel[j] = (int)((ci * (long long)p.H + (r - 1)) * p.W + (4 * cj - 4)) * 4;
  }
  // xhat / gout: instruction = tile row 2 * wave + j (j = 0, 1), lane -> pixel lane / 2, 16-byte half lane % 2
  const int t_px = lane >> 1, t_half = lane & 1;

  auto issue = [&](int tile, unsigned char* st) {
    const int tx = tile % tiles_x;
    const int t2 = tile / tiles_x;
    const int ty = t2 % tiles_y;
    const int b = t2 / tiles_y;
    const int oy0 = ty * SM_TH, ox0 = tx * SM_TW;
    const unsigned base = (unsigned)((((long long)b * 3 * p.H + 2 * oy0) * p.W + 2 * ox0) * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int iy = 2 * oy0 - 1 + w_r[j], ix = 2 * ox0 - 4 + 4 * w_j[j];
      const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const unsigned vo = ok ? base + (unsigned)w_rel[j] : 0xffffffffu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_img, (stem_lds_ptr)(st + (4 * j + wave) * 1024), 16, vo, 0, 0, 0);
    }
    unsigned char* sx = st + SM_WIN_CHUNKS * 16;
    unsigned char* sg = sx + (RECOMP ? 0 : SM_TH * SM_TW * 32);  // (RECOMP: no xhat tile, the stage is one tile smaller)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = 2 * wave + j;
      const int oy = oy0 + row, ox = ox0 + t_px;
      const bool ok = oy < p.OH && ox < p.OW;
      const long long pix = (long long)oy * p.OW + ox;
      const unsigned vx = ok ? (unsigned)((((long long)b * p.OH * p.OW + pix) * p.Cout + co0) * 2 + t_half * 16) : 0xffffffffu;
      const unsigned vg = ok ? (unsigned)(((long long)b * gout.bstride + pix * gout.ld + co0) * 2 + t_half * 16) : 0xffffffffu;
      if constexpr (!RECOMP) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_xh, (stem_lds_ptr)(sx + row * 1024), 16, vx, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (stem_lds_ptr)(sg + row * 1024), 16, vg, 0, 0, 0);
    }
  };

  int tile = blockIdx.x;
  if (tile < total_tiles) issue(tile, stage0);  // in flight while the statistics are folded

  float k1 = 0.f, k2 = 0.f, gi = 0.f;
  if constexpr (!ONEPASS) {
    fold_replicas(part, p.Cout, ws);
    const double* s0 = reinterpret_cast<const double*>(ws);
    const double* s1 = s0 + p.Cout;
    const double cnt = (double)M;
    for (int c = threadIdx.x; c < p.Cout; c += 256) {
      const float gm = k.gamma[c];
      sK1[c] = (float)(s0[c] / cnt);
      sK2[c] = (float)(s1[c] / cnt);
      sGi[c] = gm * k.invstd[c];
      if (blockIdx.x == 0 && blockIdx.y == 0) {
        dgamma[c] += (float)(s1[c] * inv_scale);
        dbeta[c] += (float)(s0[c] * inv_scale);
      }
    }
    __syncthreads();
    k1 = sK1[co0 + m];
    k2 = sK2[co0 + m];
    gi = sGi[co0 + m];
  }
  const float ga = k.gamma[co0 + m], be = k.beta[co0 + m];

  // B operand: tap n of N tile t is k = 16 t + m = (kh * 3 + kw) * 3 + ci (k >= 27: a padding column, never stored)
  int boff[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int kk = min(16 * t + m, KT - 1);
    const int ci = kk % 3, kw = (kk / 3) % 3, kh = kk / 9;
    // window column of output column c, tap kw: 2 * (ox0 + c) + kw - 1 - (2 * ox0 - 4) = 2 c + kw + 3; row of tile row r, tap kh: 2 r + kh
    boff[t] = (ci * SM_PL + (4 * wave + kh) * SM_P + kw + 3 + 2 * g) * 4;
  }
  const int aoff = (2 * wave * SM_TW + g) * 32 + m * 2;
  // RECOMP: conv operands (pixel m of the 16-pixel block, tap 4 t + g) and the step operands for pixels 4 g + i
  int toff7[7], boffr[2];
  float wreg[7];
  float mean_c = 0.f, invstd_c = 0.f;
  if constexpr (RECOMP) {
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      const int kq = 4 * t + g, kk = min(kq, KT - 1);
      const int ci = kk % 3, kw = (kk / 3) % 3, kh = kk / 9;
      toff7[t] = (ci * SM_PL + (4 * wave + kh) * SM_P + kw + 3 + 2 * m) * 4;
      wreg[t] = kq < KT ? p.w[(long long)(co0 + m) * KT + kq] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) boffr[t] = boff[t] + (2 * (4 * g) - 2 * g) * 4;
    mean_c = k.mean[co0 + m];
    invstd_c = k.invstd[co0 + m];
  }
  const int aoffr = (2 * wave * SM_TW + 4 * g) * 32 + m * 2;

  f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  f4 acx0 = {0.f, 0.f, 0.f, 0.f}, acx1 = {0.f, 0.f, 0.f, 0.f};  // ONEPASS: S3 (xhat rows)
  float sdz = 0.f, sdzx = 0.f, sb0 = 0.f, sb1 = 0.f;            // ONEPASS: per lane (pixel group g, channel / tap m)
  int cur = 0;
  for (; tile < total_tiles; tile += gridDim.x) {
    constexpr int STAGE = SM_STAGE_BYTES - (RECOMP ? SM_TH * SM_TW * 32 : 0);
    unsigned char* st = stage0 + cur * STAGE;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();  // this tile has landed for every wave, and every wave is done with the other stage
    asm volatile("" ::: "memory");
    if (tile + (int)gridDim.x < total_tiles) issue(tile + gridDim.x, stage0 + (cur ^ 1) * STAGE);
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y;
    const int oy0 = ty * SM_TH + 2 * wave, ox0 = tx * SM_TW + g;
    const unsigned char* sw = st;
    const unsigned char* sx = st + SM_WIN_CHUNKS * 16 + aoff;
    const unsigned char* sg = sx + SM_TH * SM_TW * 32;
    if constexpr (ONEPASS && RECOMP) {
      const unsigned char* sgr = st + SM_WIN_CHUNKS * 16 + aoffr;
      const int oxb = tx * SM_TW + 4 * g;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = q >> 1, cb = (q & 1) * 16;  // tile row 2 * wave + row, columns cb .. cb + 15
        f4 y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 7; ++t) {
          const float a = *reinterpret_cast<const float*>(sw + toff7[t] + (row * 2 * SM_P + 2 * cb) * 4);
          y = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[t], y, 0, 0, 0);
        }
        // the block's 16 pixels are ONE K = 16 step of v_mfma_f32_16x16x16_f16: lane (g, m) holds K = pixels 4 g + i of both operands.  dz and the
        // window values are rounded to fp16 here -- what every other layer's weight gradient does with its dy and its input (xhat is fp16 already)
        typedef _Float16 hv4 __attribute__((ext_vector_type(4)));
        hv4 dz4, xv4, b04, b14;
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // pixel cb + 4 g + i
          const bool ok = oy0 + row < p.OH && oxb + cb + i < p.OW;
          const half_t xh = (half_t)((y[i] - mean_c) * invstd_c);  // as the forward pass would have stored it
          const float xv = ok ? (float)xh : 0.f;
          const float gv = (float)*reinterpret_cast<const half_t*>(sgr + (row * SM_TW + cb + i) * 32);
          const half_t b0 = (half_t)*reinterpret_cast<const float*>(sw + boffr[0] + (row * 2 * SM_P + 2 * (cb + i)) * 4);
          const half_t b1 = (half_t)*reinterpret_cast<const float*>(sw + boffr[1] + (row * 2 * SM_P + 2 * (cb + i)) * 4);
          const half_t dzh = (half_t)(gv * cvx_silu_grad(xv * ga + be));
          const float dz = (float)dzh;
          dz4[i] = dzh;
          xv4[i] = (half_t)xv;
          b04[i] = b0;
          b14[i] = b1;
          sdz += dz;
          sdzx = fmaf(dz, xv, sdzx);
          sb0 += ok ? (float)b0 : 0.f;
          sb1 += ok ? (float)b1 : 0.f;
        }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x16f16(dz4, b04, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x16f16(dz4, b14, acc1, 0, 0, 0);
        acx0 = __builtin_amdgcn_mfma_f32_16x16x16f16(xv4, b04, acx0, 0, 0, 0);
        acx1 = __builtin_amdgcn_mfma_f32_16x16x16f16(xv4, b14, acx1, 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int row = s >> 3, c4 = (s & 7) * 4;  // tile row 2 * wave + row, column c4 + g
      const float xv = (float)*reinterpret_cast<const half_t*>(sx + s * 128);
      const float gv = (float)*reinterpret_cast<const half_t*>(sg + s * 128);
      const float b0 = *reinterpret_cast<const float*>(sw + boff[0] + (row * 2 * SM_P + 2 * c4) * 4);
      const float b1 = *reinterpret_cast<const float*>(sw + boff[1] + (row * 2 * SM_P + 2 * c4) * 4);
      const bool ok = oy0 + row < p.OH && ox0 + c4 < p.OW;
      const float dz = gv * cvx_silu_grad(xv * ga + be);
      if constexpr (ONEPASS) {
        // pixels outside the map: xhat and gout were zero-filled (dz = 0, xv = 0), only the window sum needs the mask
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dz, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dz, b1, acc1, 0, 0, 0);
        acx0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, b0, acx0, 0, 0, 0);
        acx1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, b1, acx1, 0, 0, 0);
        sdz += dz;
        sdzx = fmaf(dz, xv, sdzx);
        sb0 += ok ? b0 : 0.f;
        sb1 += ok ? b1 : 0.f;
      } else {
        const float dy = ok ? (float)(half_t)(gi * (dz - k1 - xv * k2)) : 0.f;  // rounded to fp16 once, as bn_bwd_apply stores it
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dy, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dy, b1, acc1, 0, 0, 0);
      }
    }
    }
    cur ^= 1;
  }
  if constexpr (ONEPASS) {
    // ---- the workgroup's partial block.  D row 4 g + i = channel, column m = tap within the N tile; the lane sums run over g and the waves ----
    __syncthreads();
    float* red = reinterpret_cast<float*>(stage0);  // [wave][SM_PART]
    float* mine = red + wave * SM_PART;
    sdz += __shfl_xor(sdz, 16);
    sdz += __shfl_xor(sdz, 32);
    sdzx += __shfl_xor(sdzx, 16);
    sdzx += __shfl_xor(sdzx, 32);
    sb0 += __shfl_xor(sb0, 16);
    sb0 += __shfl_xor(sb0, 32);
    sb1 += __shfl_xor(sb1, 16);
    sb1 += __shfl_xor(sb1, 32);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mine[(4 * g + i) * 32 + m] = acc0[i];
      mine[(4 * g + i) * 32 + 16 + m] = acc1[i];
      mine[512 + (4 * g + i) * 32 + m] = acx0[i];
      mine[512 + (4 * g + i) * 32 + 16 + m] = acx1[i];
    }
    if (g == 0) {
      mine[1024 + m] = sb0;
      mine[1024 + 16 + m] = sb1;
      mine[1056 + m] = sdz;
      mine[1072 + m] = sdzx;
    }
    __syncthreads();
    float* out = slabs + (long long)blockIdx.x * p.Cout * 144 + blockIdx.y * (16 * 144);
    for (int e = threadIdx.x; e < SM_PART; e += 256) out[e] = ((red[e] + red[e + SM_PART]) + red[e + 2 * SM_PART]) + red[e + 3 * SM_PART];
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned*>(out)[2 * SM_PART + 64] = 0u;  // the fold kernel's arrival counter
    return;
  }
  // the four waves' partial blocks -> one slab per workgroup: D row 4 g + i = channel, column m = tap within the N tile
  __syncthreads();
  float* red = reinterpret_cast<float*>(stage0);  // [wave][2 tiles][4 i][64 lanes]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    red[((wave * 2 + 0) * 4 + i) * 64 + lane] = acc0[i];
    red[((wave * 2 + 1) * 4 + i) * 64 + lane] = acc1[i];
  }
  __syncthreads();
  float* slab = slabs + (long long)blockIdx.x * p.Cout * 144;
  for (int e = threadIdx.x; e < 512; e += 256) {
    const int l = e & 63, i = (e >> 6) & 3, t = e >> 8;
    const int kk = 16 * t + (l & 15), co = 4 * (l >> 4) + i;
    if (kk < KT) {
      const int q = (t * 4 + i) * 64 + l;
      const float v = ((red[q] + red[q + 512]) + red[q + 1024]) + red[q + 1536];
      slab[(co0 + co) * 144 + (kk / 3) * 16 + (kk % 3)] = v;
    }
  }
#endif
}

// ---- fold of the one-pass partial blocks: column sums over the splits in fp64 (one workgroup per 16 columns), then the LAST workgroup to
// arrive forms  dW = gi (S1 - k1 S2 - k2 S3),  dgamma = sum dz xhat,  dbeta = sum dz  and adds them to the gradient arena.
// Column sums live in the unused part of the split rows' strides (row r, floats [SM_PART, 2 SM_PART) = 544 doubles): needs >= 2 splits.
constexpr int SM_FOLD_BLOCKS = SM_PART / 16;  // 68
__global__ __launch_bounds__(256) void stem_onepass_fold_kernel(float* slabs, int nsplit, int Cout, long long M, const float* gamma, const float* invstd,
                                                                float inv_scale, float* dw, float* dgamma, float* dbeta) {
  __shared__ double sd[256];
  __shared__ double col[SM_PART];
  __shared__ bool last;
  const int slice = blockIdx.y;
  const long long stride = (long long)Cout * 144;
  float* base = slabs + slice * (16 * 144);
  const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int column = blockIdx.x * 16 + c;
  // (eight independent loads per trip: one dependent 4-byte load per addition made this 17 us of the step's tail)
  double a = 0.0;
  int sp = r;
  for (; sp + 7 * 16 < nsplit; sp += 8 * 16) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = base[(long long)(sp + 16 * u) * stride + column];
#pragma unroll
    for (int u = 0; u < 8; ++u) a += (double)v[u];
  }
  for (; sp < nsplit; sp += 16) a += (double)base[(long long)sp * stride + column];
  sd[threadIdx.x] = a;
  __syncthreads();
  for (int off = 128; off >= 16; off >>= 1) {
    if ((int)threadIdx.x < off) sd[threadIdx.x] += sd[threadIdx.x + off];
    __syncthreads();
  }
  auto colsum_at = [&](int q) -> double* { return reinterpret_cast<double*>(base + (long long)(q / 544) * stride + SM_PART) + (q % 544); };
  if (threadIdx.x < 16) __hip_atomic_store(colsum_at(column), sd[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    unsigned* ctr = reinterpret_cast<unsigned*>(base) + 2 * SM_PART + 64;
    last = atomicAdd(ctr, 1u) == gridDim.x - 1;
    if (last) __threadfence();
  }
  __syncthreads();
  if (!last) return;
  for (int q = threadIdx.x; q < SM_PART; q += 256) col[q] = __hip_atomic_load(colsum_at(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const double cnt = (double)M;
  for (int e = threadIdx.x; e < 16 * KT; e += 256) {
    const int co = e / KT, kk = e - co * KT;
    const double k1 = col[1056 + co] / cnt, k2 = col[1072 + co] / cnt;
    const double g = (double)gamma[slice * 16 + co] * (double)invstd[slice * 16 + co];
    const double v = g * (col[co * 32 + kk] - k1 * col[1024 + kk] - k2 * col[512 + co * 32 + kk]);
    dw[(long long)(slice * 16 + co) * KT + kk] += (float)(v * inv_scale);
  }
  if (threadIdx.x < 16) {
    dgamma[slice * 16 + threadIdx.x] += (float)(col[1072 + threadIdx.x] * inv_scale);
    dbeta[slice * 16 + threadIdx.x] += (float)(col[1056 + threadIdx.x] * inv_scale);
  }
}

// plain slabs [nsplit][Cout][9 * 16] -> dw [Cout][27] (+=): what the engine's table-driven reducer does for every other layer
__global__ __launch_bounds__(256) void stem_slab_fold_kernel(const float* slabs, int nsplit, int Cout, float inv_scale, float* dw) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= Cout * KT) return;
  const int co = e / KT, kk = e - co * KT;
  const float* s = slabs + (long long)co * 144 + (kk / 3) * 16 + (kk % 3);
  float a = 0.f;
  for (int sp = 0; sp < nsplit; ++sp) a += s[(long long)sp * Cout * 144];
  dw[e] += a * inv_scale;
}

int stem_grid(long long M, int per_block) {
  long long blocks = (M + per_block - 1) / per_block;
  const long long cap = 256 * 8;  // persistent: grid-stride over the pixels, 8 workgroups per CU at most
  return (int)(blocks < cap ? blocks : cap);
}

int check(const StemParams& p) {
  CVX_CHECK(p.img && p.w, "stem: null pointers");
  CVX_CHECK(p.H % 2 == 0 && p.W % 2 == 0 && p.OH == p.H / 2 && p.OW == p.W / 2, "stem: even input sizes, output = input / 2");
  CVX_CHECK(p.Cout % 16 == 0 && p.Cout >= 16 && p.Cout <= 80, "stem: 16..80 output channels in multiples of 16");
  CVX_CHECK((long long)p.B * p.OH * p.OW < (1LL << 31), "stem: too many output pixels");
  CVX_CHECK(((uintptr_t)p.img % 8) == 0, "stem: images must be 8-byte aligned");
  return 0;
}

}  // namespace

int cvx_stem_stats(const StemParams& p, long long* stats, hipStream_t st) {
  CVX_TRY(check(p));
  const long long M = (long long)p.B * p.OH * p.OW;
  const int lds = (KT + 8) * p.Cout * 4;
  const dim3 grid(stem_grid(M, 256));
  switch (p.Cout / 16) {
    case 1: hipLaunchKernelGGL(stem_stats_kernel<1>, grid, dim3(256), lds, st, p, M, stats); break;
    case 2: hipLaunchKernelGGL(stem_stats_kernel<2>, grid, dim3(256), lds, st, p, M, stats); break;
    case 3: hipLaunchKernelGGL(stem_stats_kernel<3>, grid, dim3(256), lds, st, p, M, stats); break;
    case 4: hipLaunchKernelGGL(stem_stats_kernel<4>, grid, dim3(256), lds, st, p, M, stats); break;
    default: hipLaunchKernelGGL(stem_stats_kernel<5>, grid, dim3(256), lds, st, p, M, stats); break;
  }
  CVX_HIP(hipGetLastError());
  return 0;
}

template <bool TRAIN>
static int launch_apply(const StemParams& p, const BnTrainArgs& a, const float* scale, const float* shift, const ViewDesc& out,
                        half_t* xhat, hipStream_t st) {
  const long long M = (long long)p.B * p.OH * p.OW;
  const int lds = (TRAIN ? fold_ws_bytes(p.Cout) : 0) + (4 + KT) * p.Cout * 4;
  const dim3 grid(stem_grid(M, 256));
  switch (p.Cout / 16) {
    case 1: hipLaunchKernelGGL((stem_apply_kernel<1, TRAIN>), grid, dim3(256), lds, st, p, M, a, scale, shift, out, xhat); break;
    case 2: hipLaunchKernelGGL((stem_apply_kernel<2, TRAIN>), grid, dim3(256), lds, st, p, M, a, scale, shift, out, xhat); break;
    case 3: hipLaunchKernelGGL((stem_apply_kernel<3, TRAIN>), grid, dim3(256), lds, st, p, M, a, scale, shift, out, xhat); break;
    case 4: hipLaunchKernelGGL((stem_apply_kernel<4, TRAIN>), grid, dim3(256), lds, st, p, M, a, scale, shift, out, xhat); break;
    default: hipLaunchKernelGGL((stem_apply_kernel<5, TRAIN>), grid, dim3(256), lds, st, p, M, a, scale, shift, out, xhat); break;
  }
  CVX_HIP(hipGetLastError());
  return 0;
}

int cvx_stem_apply_train(const StemParams& p, const BnTrainArgs& a, const ViewDesc& out, half_t* xhat, hipStream_t st) {
  CVX_TRY(check(p));
  CVX_CHECK(out.p && ((uintptr_t)out.p % 16) == 0 && out.ld % 8 == 0, "stem: output view");  // (xhat == nullptr: not stored, cvx_stem_keeps_xhat)
  return launch_apply<true>(p, a, nullptr, nullptr, out, xhat, st);
}

int cvx_stem_apply_eval(const StemParams& p, const float* scale, const float* shift, const ViewDesc& out, hipStream_t st) {
  CVX_TRY(check(p));
  CVX_CHECK(out.p && scale && shift && ((uintptr_t)out.p % 16) == 0 && out.ld % 8 == 0, "stem: output view");
  BnTrainArgs none{};
  return launch_apply<false>(p, none, scale, shift, out, nullptr, st);
}

int cvx_stem_wgrad_splits(long long M) {
  static const int cap = cvx_tune_int("CVX_STEM_SPLITS", 512);
  return stem_grid(M, 64 * 16) < cap ? stem_grid(M, 64 * 16) : cap;
}

int cvx_stem_wgrad(const StemParams& p, const half_t* dy, float* slabs, int nsplit, hipStream_t st) {
  CVX_TRY(check(p));
  const long long M = (long long)p.B * p.OH * p.OW;
  CVX_CHECK(dy && slabs && nsplit >= 1 && nsplit == cvx_stem_wgrad_splits(M), "stem wgrad: split count must come from cvx_stem_wgrad_splits");
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(nsplit, p.Cout / 16), dim3(256), 0, st, p, M, dy, slabs);
  CVX_HIP(hipGetLastError());
  return 0;
}

int cvx_stem_backward(const StemParams& p, const half_t* xhat, const ViewDesc& gout, const BnCoef& k, const long long* part, float inv_scale,
                      float* dgamma, float* dbeta, float* slabs, int nsplit, hipStream_t st) {
  CVX_TRY(check(p));
  const long long M = (long long)p.B * p.OH * p.OW;
  CVX_CHECK(xhat && gout.p && part && dgamma && dbeta && slabs && nsplit == cvx_stem_wgrad_splits(M), "stem backward: bad arguments");
  CVX_CHECK(gout.ld % 4 == 0 && ((uintptr_t)gout.p % 8) == 0, "stem backward: gradient view alignment");
  const int tiles_x = (p.OW + 31) / 32, tiles_y = (p.OH + 7) / 8;
  // the matrix-core kernel moves 16-byte pieces: image rows and the two gradient-side tensors must be 16-byte granular and within a buffer
  // descriptor's 4 GiB
  static const bool mfma_on = cvx_tune_int("CVX_STEM_BWD_MFMA", 1) != 0;
  const bool mfma_ok = mfma_on && p.W % 4 == 0 && ((uintptr_t)p.img % 16) == 0 && ((uintptr_t)xhat % 16) == 0 && ((uintptr_t)gout.p % 16) == 0 &&
                       gout.ld % 8 == 0 && gout.bstride % 8 == 0 && (long long)p.B * 3 * p.H * p.W * 4 < (1LL << 32) && M * p.Cout * 2 < (1LL << 32) &&
                       (long long)p.B * gout.bstride * 2 < (1LL << 32);
  if (mfma_ok) {
    const int lds_m = fold_ws_bytes(p.Cout) + 5 * p.Cout * 4 + 32 + 2 * SM_STAGE_BYTES;
    hipLaunchKernelGGL(stem_bwd_mfma_kernel<false>, dim3(nsplit, p.Cout / 16), dim3(256), lds_m, st, p, M, xhat, gout, k, part, inv_scale, dgamma, dbeta,
                       slabs, tiles_x, tiles_y, tiles_x * tiles_y * p.B);
    CVX_HIP(hipGetLastError());
    return 0;
  }
  const int lds = fold_ws_bytes(p.Cout) + 5 * p.Cout * 4 + 3 * 17 * 66 * 4;
  hipLaunchKernelGGL(stem_bwd_kernel, dim3(nsplit, p.Cout / 16), dim3(256), lds, st, p, M, xhat, gout, k, part, inv_scale, dgamma, dbeta, slabs,
                     tiles_x, tiles_y, tiles_x * tiles_y * p.B);
  CVX_HIP(hipGetLastError());
  return 0;
}

// ---- the whole backward of the stem with the fold into the gradient arena (what the engine and cvx_stem_backward_nchw call) ----
static bool stem_dma_ok(const StemParams& p, const half_t* xhat, const ViewDesc& gout, long long M) {
  return p.W % 4 == 0 && ((uintptr_t)p.img % 16) == 0 && ((uintptr_t)xhat % 16) == 0 && ((uintptr_t)gout.p % 16) == 0 && gout.ld % 8 == 0 &&
         gout.bstride % 8 == 0 && (long long)p.B * 3 * p.H * p.W * 4 < (1LL << 32) && M * p.Cout * 2 < (1LL << 32) &&
         (long long)p.B * gout.bstride * 2 < (1LL << 32);
}
bool cvx_stem_keeps_xhat(const StemParams& p, const ViewDesc& gout, int nsplit) {
  static const bool recomp = cvx_tune_int("CVX_STEM_RECOMP", 1) != 0;
  // (the one-pass conditions with a 16-byte aligned dummy in place of the xhat pointer: xhat is what would not exist)
  return !(recomp && cvx_stem_backward_onepass_ok(p, reinterpret_cast<const half_t*>((uintptr_t)16), gout, nsplit));
}
bool cvx_stem_backward_onepass_ok(const StemParams& p, const half_t* xhat, const ViewDesc& gout, int nsplit) {
  static const bool on = cvx_tune_int("CVX_STEM_ONEPASS", 1) != 0;
  const long long M = (long long)p.B * p.OH * p.OW;
  return on && nsplit >= 2 && p.Cout % 16 == 0 && stem_dma_ok(p, xhat, gout, M);
}
int cvx_stem_backward_fold(const StemParams& p, const half_t* xhat, const ViewDesc& gout, const BnCoef& k, const long long* part, float inv_scale,
                           float* dgamma, float* dbeta, float* dw, float* slabs, int nsplit, hipStream_t st) {
  CVX_TRY(check(p));
  const long long M = (long long)p.B * p.OH * p.OW;
  CVX_CHECK(xhat && gout.p && dgamma && dbeta && dw && slabs && nsplit == cvx_stem_wgrad_splits(M), "stem backward: bad arguments");
  if (cvx_stem_backward_onepass_ok(p, xhat, gout, nsplit)) {
    const int tiles_x = (p.OW + 31) / 32, tiles_y = (p.OH + 7) / 8;
    const int lds_m = fold_ws_bytes(p.Cout) + 5 * p.Cout * 4 + 32 + 2 * SM_STAGE_BYTES;
    if (k.mean && !cvx_stem_keeps_xhat(p, gout, nsplit)) {  // the forward pass stored no xhat: recomputed from the window
      hipLaunchKernelGGL((stem_bwd_mfma_kernel<true, true>), dim3(nsplit, p.Cout / 16), dim3(256), lds_m - 2 * SM_TH * SM_TW * 32, st, p, M, xhat, gout, k, (const long long*)nullptr,
                         inv_scale, dgamma, dbeta, slabs, tiles_x, tiles_y, tiles_x * tiles_y * p.B);
    } else
    hipLaunchKernelGGL(stem_bwd_mfma_kernel<true>, dim3(nsplit, p.Cout / 16), dim3(256), lds_m, st, p, M, xhat, gout, k, (const long long*)nullptr, inv_scale,
                       dgamma, dbeta, slabs, tiles_x, tiles_y, tiles_x * tiles_y * p.B);
    hipLaunchKernelGGL(stem_onepass_fold_kernel, dim3(SM_FOLD_BLOCKS, p.Cout / 16), dim3(256), 0, st, slabs, nsplit, p.Cout, M, k.gamma, k.invstd, inv_scale,
                       dw, dgamma, dbeta);
    CVX_HIP(hipGetLastError());
    return 0;
  }
  CVX_CHECK(part, "stem backward: the two-pass path needs the sums of bn_bwd_reduce");
  CVX_TRY(cvx_stem_backward(p, xhat, gout, k, part, inv_scale, dgamma, dbeta, slabs, nsplit, st));
  hipLaunchKernelGGL(stem_slab_fold_kernel, dim3((p.Cout * KT + 255) / 256), dim3(256), 0, st, slabs, nsplit, p.Cout, inv_scale, dw);
  CVX_HIP(hipGetLastError());
  return 0;
}
