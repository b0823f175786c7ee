// HBM-bound helper kernels (gfx950): layout change, SPPF max-pool, nearest upsample, weight shadow
// packing, gradient-slab reduction, fused Adam.  16-byte accesses wherever the layout allows.
#include <algorithm>
#include "misc_ops.h"
#include "bn_common.h"

namespace {

__device__ __forceinline__ long long voff(const ViewDesc& v, int b, long long pix) { return (long long)b * v.bstride + pix * v.ld; }

// ---- NCHW fp32 image -> NHWC fp16 with C padded 3 -> 8 (16 B per pixel): first layers other than the YOLO stem ----
__global__ void image_to_nhwc8_kernel(const float* img, int B, int HW, half_t* out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)B * HW;
  if (i >= n) return;
  long long b = i / HW, pix = i - b * HW;
  const float* src = img + b * 3 * HW + pix;
  h8 o = {(half_t)src[0], (half_t)src[HW], (half_t)src[2LL * HW], (half_t)0, (half_t)0, (half_t)0, (half_t)0, (half_t)0};
  *reinterpret_cast<h8*>(out + i * 8) = o;
}

// ---- max pool 2x2 stride 2 (DLA Tree.downsample, centernet_model.py:128-129) ----
__global__ void maxpool2_kernel(ViewDesc in, ViewDesc out, int B, int IH, int IW, int OH, int OW, int CG) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)B * OH * OW * CG;
  if (i >= n) return;
  int cg = (int)(i % CG);
  long long t = i / CG;
  int w = (int)(t % OW);
  t /= OW;
  int h = (int)(t % OH);
  int b = (int)(t / OH);
  float best[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) best[k] = -INFINITY;
  // ceil_mode: the last window of an odd size is clipped -- its missing taps are clamped onto the window's own last row / column (the
  // maximum does not change), so the four loads are unconditional and issued together
  h8 v4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int hh = min(2 * h + (q >> 1), IH - 1), ww = min(2 * w + (q & 1), IW - 1);
    v4[q] = *reinterpret_cast<const h8*>(in.p + voff(in, b, (long long)hh * IW + ww) + cg * 8);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int k = 0; k < 8; ++k) best[k] = fmaxf(best[k], (float)v4[q][k]);
  h8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (half_t)best[k];
  *reinterpret_cast<h8*>(out.p + voff(out, b, (long long)h * OW + w) + cg * 8) = o;
}

// gradient of the 2x2 stride-2 max pool: the windows do not overlap, so one thread owns a window -- it reads the four forward inputs once,
// re-derives the first maximum (row-major scan, torch's rule; no argmax tensor) and writes all four input gradients.  Pixels outside
// every window (floor mode on an odd size) belong to the threads of an extra window row / column and receive zero.
__global__ void maxpool2_bwd_kernel(ViewDesc in, ViewDesc gout, ViewDesc gin, int B, int IH, int IW, int OH, int OW, int CG, int accumulate) {
  const int WH = (IH + 1) >> 1, WW = (IW + 1) >> 1;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)B * WH * WW * CG;
  if (i >= n) return;
  int cg = (int)(i % CG);
  long long t = i / CG;
  int w = (int)(t % WW);
  t /= WW;
  int h = (int)(t % WH);
  int b = (int)(t / WH);
  const bool window = h < OH && w < OW;
  int bi[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bi[k] = -1;
  h8 g;
  if (window) {
    float best[8];
    bool first = true;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int y = 2 * h + (q >> 1), x = 2 * w + (q & 1);
      if (y >= IH || x >= IW) continue;
      const h8 v = *reinterpret_cast<const h8*>(in.p + voff(in, b, (long long)y * IW + x) + cg * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float f = (float)v[k];
        if (first || f > best[k]) {
          best[k] = f;
          bi[k] = q;
        }
      }
      first = false;
    }
    g = *reinterpret_cast<const h8*>(gout.p + voff(gout, b, (long long)h * OW + w) + cg * 8);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int y = 2 * h + (q >> 1), x = 2 * w + (q & 1);
    if (y >= IH || x >= IW) continue;
    half_t* dst = gin.p + voff(gin, b, (long long)y * IW + x) + cg * 8;
    h8 o;
    if (accumulate) {
      if (!window) continue;
      o = *reinterpret_cast<const h8*>(dst);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (bi[k] == q) o[k] = (half_t)((float)o[k] + (float)g[k]);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (bi[k] == q) ? g[k] : (half_t)0.f;
    }
    *reinterpret_cast<h8*>(dst) = o;
  }
}

// ---- max pool 3x3 stride 2 pad 1 (ResNet stem, core/models/resnet.py:163) ----
// idx (training): one byte per output element, the window tap dy*3+dx of the first maximum in row-major scan order (torch's rule)
__global__ void maxpool3_kernel(ViewDesc in, ViewDesc out, int B, int IH, int IW, int OH, int OW, int CG, int stride, uint8_t* idx) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)B * OH * OW * CG;
  if (i >= n) return;
  int cg = (int)(i % CG);
  long long t = i / CG;
  int w = (int)(t % OW);
  t /= OW;
  int h = (int)(t % OH);
  int b = (int)(t / OH);
  float best[8];
  int bi[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    best[k] = -INFINITY;
    bi[k] = 0;
  }
  bool first = true;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int hh = stride * h - 1 + dy;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int ww = stride * w - 1 + dx;
      if (hh < 0 || hh >= IH || ww < 0 || ww >= IW) continue;
      const h8 v = *reinterpret_cast<const h8*>(in.p + voff(in, b, (long long)hh * IW + ww) + cg * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float f = (float)v[k];
        if (first || f > best[k]) {
          best[k] = f;
          bi[k] = dy * 3 + dx;
        }
      }
      first = false;
    }
  }
  h8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (half_t)best[k];
  const long long pix = (long long)h * OW + w;
  *reinterpret_cast<h8*>(out.p + voff(out, b, pix) + cg * 8) = o;
  if (idx) {
    unsigned long long pk = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) pk |= (unsigned long long)(bi[k] & 0xff) << (8 * k);
    *reinterpret_cast<unsigned long long*>(idx + (((long long)b * OH * OW + pix) * CG + cg) * 8) = pk;
  }
}

// gradient of the 3x3 pad-1 max pool: a gather over the (at most 3 x 3 / stride^2) windows that hold the input pixel
__global__ void maxpool3_bwd_kernel(ViewDesc gout, ViewDesc gin, int B, int IH, int IW, int OH, int OW, int CG, int stride, const uint8_t* idx,
                                    int accumulate) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)B * IH * IW * CG;
  if (i >= n) return;
  int cg = (int)(i % CG);
  long long t = i / CG;
  int ww = (int)(t % IW);
  t /= IW;
  int hh = (int)(t % IH);
  int b = (int)(t / IH);
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int nh = hh + 1 - dy;  // stride * h = hh + 1 - dy
    if (nh < 0 || nh % stride != 0 || nh / stride >= OH) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int nw = ww + 1 - dx;
      if (nw < 0 || nw % stride != 0 || nw / stride >= OW) continue;
      const long long pix = (long long)(nh / stride) * OW + nw / stride;
      const unsigned long long pk = *reinterpret_cast<const unsigned long long*>(idx + (((long long)b * OH * OW + pix) * CG + cg) * 8);
      const h8 g = *reinterpret_cast<const h8*>(gout.p + voff(gout, b, pix) + cg * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if ((int)((pk >> (8 * k)) & 0xff) == dy * 3 + dx) acc[k] += (float)g[k];
    }
  }
  half_t* q = gin.p + voff(gin, b, (long long)hh * IW + ww) + cg * 8;
  if (accumulate) {
    const h8 old = *reinterpret_cast<const h8*>(q);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += (float)old[k];
  }
  h8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (half_t)acc[k];
  *reinterpret_cast<h8*>(q) = o;
}

__global__ void zero_slice_kernel(ViewDesc v, int B, int HW, int CG) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)B * HW * CG;
  if (i >= n) return;
  int cg = (int)(i % CG);
  long long t = i / CG;
  int pix = (int)(t % HW);
  int b = (int)(t / HW);
  const h8 z = {};
  *reinterpret_cast<h8*>(v.p + voff(v, b, pix) + cg * 8) = z;
}

// gradient of the global average pool: every pixel of the map receives gout / HW
__global__ void avgpool_global_bwd_kernel(ViewDesc gout, ViewDesc gin, int B, int HW, int CG, int accumulate) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)B * HW * CG;
  if (i >= n) return;
  int cg = (int)(i % CG);
  long long t = i / CG;
  int pix = (int)(t % HW);
  int b = (int)(t / HW);
  const h8 g = *reinterpret_cast<const h8*>(gout.p + voff(gout, b, 0) + cg * 8);
  half_t* q = gin.p + voff(gin, b, pix) + cg * 8;
  const float inv = 1.f / (float)HW;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = (float)g[k] * inv;
  if (accumulate) {
    const h8 old = *reinterpret_cast<const h8*>(q);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += (float)old[k];
  }
  h8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (half_t)acc[k];
  *reinterpret_cast<h8*>(q) = o;
}

// inverted dropout (nn.Dropout, deeplabv3plus.py:67): out = keep ? in / (1 - p) : 0 with keep drawn per element from a counter-based
// hash of (seed, element index) -- the backward pass re-derives the same mask from the same seed instead of storing it.
// (torch draws its mask from its own Philox stream: the two masks are different samples of the same distribution.)
__device__ __forceinline__ unsigned dropout_hash(unsigned long long seed, unsigned long long i) {
  unsigned long long z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return (unsigned)((z ^ (z >> 31)) >> 32);
}
__global__ void dropout_kernel(ViewDesc in, ViewDesc out, int B, int HW, int CG, unsigned thresh, float scale, unsigned long long seed, int accumulate) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)B * HW * CG;
  if (i >= n) return;
  int cg = (int)(i % CG);
  long long t = i / CG;
  int pix = (int)(t % HW);
  int b = (int)(t / HW);
  const h8 v = *reinterpret_cast<const h8*>(in.p + voff(in, b, pix) + cg * 8);
  half_t* q = out.p + voff(out, b, pix) + cg * 8;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = dropout_hash(seed, (unsigned long long)i * 8 + k) >= thresh ? (float)v[k] * scale : 0.f;
  if (accumulate) {
    const h8 old = *reinterpret_cast<const h8*>(q);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += (float)old[k];
  }
  h8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (half_t)acc[k];
  *reinterpret_cast<h8*>(q) = o;
}

// ---- L2Normalize (core/models/ssd_model.py:113-128): x / (sqrt(sum_c x^2) + 1e-10) * weight[c]; one wave per pixel ----
__global__ __launch_bounds__(256) void l2norm_kernel(ViewDesc in, ViewDesc out, const float* weight, long long npix, int hw, int C) {
  const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= npix) return;
  const int lane = threadIdx.x & 63;
  const int b = (int)(pix / hw);
  const long long p = pix - (long long)b * hw;
  const half_t* src = in.p + voff(in, b, p);
  float ss = 0.f;
  for (int c = lane * 8; c < C; c += 512) {
    const h8 v = *reinterpret_cast<const h8*>(src + c);
#pragma unroll
    for (int k = 0; k < 8; ++k) ss += (float)v[k] * (float)v[k];
  }
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const float inv = 1.f / (sqrtf(ss) + 1e-10f);
  half_t* dst = out.p + voff(out, b, p);
  for (int c = lane * 8; c < C; c += 512) {
    const h8 v = *reinterpret_cast<const h8*>(src + c);
    h8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (half_t)(weight[c + k] * ((float)v[k] * inv));
    *reinterpret_cast<h8*>(dst + c) = o;
  }
}

// gradient of L2Normalize, y = w * x / (n + eps), n = ||x||_2 over the channels of a pixel:
//   dx = w * g / (n + eps) - x * (sum_c w g x) / (n (n + eps)^2),   dw_c = sum over pixels of g_c x_c / (n + eps).
// One wave per pixel (C <= 512: one 8-channel group per lane); every workgroup parks its four waves' dw partial sums, a second
// launch adds the workgroups' partials in fixed order (deterministic).
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(ViewDesc in, ViewDesc gout, ViewDesc gin, const float* weight, long long npix, int hw, int C,
                                                         int pix_per_wave, int accumulate, float* partial) {
  __shared__ float sdw[4 * 512];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c0 = lane * 8;
  const bool on = c0 < C;
  float wv[8], dw[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    wv[k] = on ? weight[c0 + k] : 0.f;
    dw[k] = 0.f;
  }
  const long long p0 = ((long long)blockIdx.x * 4 + wave) * pix_per_wave;
  for (long long pix = p0; pix < min(npix, p0 + pix_per_wave); ++pix) {
    const int b = (int)(pix / hw);
    const long long p = pix - (long long)b * hw;
    h8 xv = {}, gv = {};
    if (on) {
      xv = *reinterpret_cast<const h8*>(in.p + voff(in, b, p) + c0);
      gv = *reinterpret_cast<const h8*>(gout.p + voff(gout, b, p) + c0);
    }
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      ss += (float)xv[k] * (float)xv[k];
      dot += wv[k] * (float)gv[k] * (float)xv[k];
    }
    for (int o = 32; o > 0; o >>= 1) {
      ss += __shfl_xor(ss, o);
      dot += __shfl_xor(dot, o);
    }
    const float n = sqrtf(ss), inv = 1.f / (n + 1e-10f);
    const float coef = n > 0.f ? dot * inv * inv / n : 0.f;
    if (on) {
      half_t* q = gin.p + voff(gin, b, p) + c0;
      h8 o;
      h8 old = {};
      if (accumulate) old = *reinterpret_cast<const h8*>(q);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        o[k] = (half_t)(wv[k] * (float)gv[k] * inv - (float)xv[k] * coef + (accumulate ? (float)old[k] : 0.f));
        dw[k] += (float)gv[k] * (float)xv[k] * inv;
      }
      *reinterpret_cast<h8*>(q) = o;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) sdw[wave * 512 + c0 + k] = dw[k];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256)
    partial[(long long)blockIdx.x * C + c] = (sdw[c] + sdw[512 + c]) + (sdw[1024 + c] + sdw[1536 + c]);
}
// 16 channels x 16 lanes per workgroup: lane j adds the workgroups j, j + 16, ... (four independent chains), the lanes are folded in order
__global__ __launch_bounds__(256) void l2norm_dw_kernel(const float* __restrict__ partial, int nblocks, int C, float inv_scale, float* dw) {
  __shared__ float red[256];
  const int cl = threadIdx.x & 15, lane = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < C) {
    int b = lane;
    for (; b + 48 < nblocks; b += 64) {
      a0 += partial[(long long)b * C + c];
      a1 += partial[(long long)(b + 16) * C + c];
      a2 += partial[(long long)(b + 32) * C + c];
      a3 += partial[(long long)(b + 48) * C + c];
    }
    for (; b < nblocks; b += 16) a0 += partial[(long long)b * C + c];
  }
  red[lane * 16 + cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (threadIdx.x < 16 && c < C) {
    float acc = 0.f;
    for (int j = 0; j < 16; ++j) acc += red[j * 16 + cl];
    dw[c] += acc * inv_scale;
  }
}

// ---- global average pool (ASPPPooling's AdaptiveAvgPool2d(1), core/models/deeplabv3plus.py:30): one workgroup per
// (image, 8-channel group), fp32 sums in a fixed order ----
__global__ __launch_bounds__(256) void avgpool_global_kernel(ViewDesc in, ViewDesc out, int HW, int CG) {
  __shared__ float red[256 * 8];
  const int b = blockIdx.x / CG, cg = blockIdx.x - b * CG;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  for (int pix = threadIdx.x; pix < HW; pix += 256) {
    const h8 v = *reinterpret_cast<const h8*>(in.p + voff(in, b, pix) + cg * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += (float)v[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = acc[k];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s)
#pragma unroll
      for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] += red[(threadIdx.x + s) * 8 + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    h8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (half_t)(red[k] / (float)HW);
    *reinterpret_cast<h8*>(out.p + voff(out, b, 0) + cg * 8) = o;
  }
}

// ---- bilinear resize, align_corners = False (F.interpolate(..., mode="bilinear"), deeplabv3plus.py:38,117-122,147):
// src = (dst + 0.5) * (in / out) - 0.5, clamped at 0; a 1x1 input degenerates to a broadcast ----
__device__ __forceinline__ void bilinear_src(int d, float scale, int in_size, int* i0, int* i1, float* lam) {
  float s = ((float)d + 0.5f) * scale - 0.5f;
  if (s < 0.f) s = 0.f;
  int a = (int)s;
  if (a > in_size - 1) a = in_size - 1;
  *i0 = a;
  *i1 = a + (a < in_size - 1 ? 1 : 0);
  *lam = s - (float)a;
}
__global__ void resize_bilinear_kernel(ViewDesc in, ViewDesc out, int B, int IH, int IW, int OH, int OW, int CG, float sh, float sw) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)B * OH * OW * CG;
  if (i >= n) return;
  int cg = (int)(i % CG);
  long long t = i / CG;
  int w = (int)(t % OW);
  t /= OW;
  int h = (int)(t % OH);
  int b = (int)(t / OH);
  int y0, y1, x0, x1;
  float ly, lx;
  bilinear_src(h, sh, IH, &y0, &y1, &ly);
  bilinear_src(w, sw, IW, &x0, &x1, &lx);
  const h8 v00 = *reinterpret_cast<const h8*>(in.p + voff(in, b, (long long)y0 * IW + x0) + cg * 8);
  const h8 v01 = *reinterpret_cast<const h8*>(in.p + voff(in, b, (long long)y0 * IW + x1) + cg * 8);
  const h8 v10 = *reinterpret_cast<const h8*>(in.p + voff(in, b, (long long)y1 * IW + x0) + cg * 8);
  const h8 v11 = *reinterpret_cast<const h8*>(in.p + voff(in, b, (long long)y1 * IW + x1) + cg * 8);
  h8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float top = (float)v00[k] * (1.f - lx) + (float)v01[k] * lx;
    const float bot = (float)v10[k] * (1.f - lx) + (float)v11[k] * lx;
    o[k] = (half_t)(top * (1.f - ly) + bot * ly);
  }
  *reinterpret_cast<h8*>(out.p + voff(out, b, (long long)h * OW + w) + cg * 8) = o;
}
// gradient of the bilinear resize as a gather (deterministic, no atomics): input pixel (y, x) collects every output pixel whose
// two source rows / columns include it.  The candidate range is a superset derived from src = (dst + 0.5) * scale - 0.5; each
// candidate re-derives its (i0, i1, lambda) with the forward's arithmetic, so the weights are the forward's bit for bit.
__device__ __forceinline__ void bilinear_dst_range(int s, float scale, int out_size, int* lo, int* hi) {
  const float a = ((float)s - 0.5f) / scale - 0.5f, b = ((float)s + 1.5f) / scale - 0.5f;
  int l = (int)floorf(a) - 1, h = (int)ceilf(b) + 1;
  *lo = l < 0 ? 0 : l;
  *hi = h > out_size - 1 ? out_size - 1 : h;
}
__global__ void resize_bilinear_bwd_kernel(ViewDesc gout, ViewDesc gin, int B, int IH, int IW, int OH, int OW, int CG, float sh, float sw,
                                           int accumulate) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)B * IH * IW * CG;
  if (i >= n) return;
  int cg = (int)(i % CG);
  long long t = i / CG;
  int x = (int)(t % IW);
  t /= IW;
  int y = (int)(t % IH);
  int b = (int)(t / IH);
  int oy0, oy1, ox0, ox1;
  bilinear_dst_range(y, sh, OH, &oy0, &oy1);
  bilinear_dst_range(x, sw, OW, &ox0, &ox1);
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  for (int oy = oy0; oy <= oy1; ++oy) {
    int y0, y1;
    float ly;
    bilinear_src(oy, sh, IH, &y0, &y1, &ly);
    const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
    if (wy == 0.f) continue;
    for (int ox = ox0; ox <= ox1; ++ox) {
      int x0, x1;
      float lx;
      bilinear_src(ox, sw, IW, &x0, &x1, &lx);
      const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
      if (wx == 0.f) continue;
      const h8 g = *reinterpret_cast<const h8*>(gout.p + voff(gout, b, (long long)oy * OW + ox) + cg * 8);
      const float wgt = wy * wx;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = fmaf(wgt, (float)g[k], acc[k]);
    }
  }
  half_t* q = gin.p + voff(gin, b, (long long)y * IW + x) + cg * 8;
  if (accumulate) {
    const h8 old = *reinterpret_cast<const h8*>(q);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += (float)old[k];
  }
  h8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (half_t)acc[k];
  *reinterpret_cast<h8*>(q) = o;
}
// ... from a 1 x 1 input (ASPPPooling's broadcast): the gradient is the sum over all output pixels -- one workgroup per (image, channel
// group), fixed-order tree (the gather above would walk the whole map in every thread)
__global__ __launch_bounds__(256) void resize_from_1x1_bwd_kernel(ViewDesc gout, ViewDesc gin, int HW, int CG, int accumulate) {
  __shared__ float red[256 * 8];
  const int b = blockIdx.x / CG, cg = blockIdx.x - b * CG;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  for (int pix = threadIdx.x; pix < HW; pix += 256) {
    const h8 v = *reinterpret_cast<const h8*>(gout.p + voff(gout, b, pix) + cg * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += (float)v[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = acc[k];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s)
#pragma unroll
      for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] += red[(threadIdx.x + s) * 8 + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    half_t* q = gin.p + voff(gin, b, 0) + cg * 8;
    h8 o;
    if (accumulate) {
      const h8 old = *reinterpret_cast<const h8*>(q);
#pragma unroll
      for (int k = 0; k < 8; ++k) red[k] += (float)old[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (half_t)red[k];
    *reinterpret_cast<h8*>(q) = o;
  }
}
// fp32 rows (B, IH*IW, ld) -> NCHW fp32 (B, C, OH, OW): the segmentation logits back at input resolution (deeplabv3plus.py:147)
// One thread per output pixel walks the classes: the source indices and weights are computed once, the four source rows are read with
// 16-byte loads (their C floats are contiguous), every class plane is written with consecutive lanes on consecutive x.
__global__ __launch_bounds__(256) void resize_bilinear_f32_nchw_kernel(const float* in, int ld, int B, int C, int IH, int IW, int OH, int OW, float sh,
                                                                       float sw, float* out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)B * OH * OW;
  if (i >= n) return;
  int w = (int)(i % OW);
  long long t = i / OW;
  int h = (int)(t % OH);
  int b = (int)(t / OH);
  int y0, y1, x0, x1;
  float ly, lx;
  bilinear_src(h, sh, IH, &y0, &y1, &ly);
  bilinear_src(w, sw, IW, &x0, &x1, &lx);
  const float* base = in + (long long)b * IH * IW * ld;
  const float* r00 = base + ((long long)y0 * IW + x0) * ld;
  const float* r01 = base + ((long long)y0 * IW + x1) * ld;
  const float* r10 = base + ((long long)y1 * IW + x0) * ld;
  const float* r11 = base + ((long long)y1 * IW + x1) * ld;
  const long long plane = (long long)OH * OW;
  float* o = out + (long long)b * C * plane + (long long)h * OW + w;
  int c = 0;
  if ((ld & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0) {
    for (; c + 4 <= C; c += 4) {
      const float4 a = *reinterpret_cast<const float4*>(r00 + c), bq = *reinterpret_cast<const float4*>(r01 + c);
      const float4 cq = *reinterpret_cast<const float4*>(r10 + c), d = *reinterpret_cast<const float4*>(r11 + c);
      const float va[4] = {a.x, a.y, a.z, a.w}, vb[4] = {bq.x, bq.y, bq.z, bq.w}, vc[4] = {cq.x, cq.y, cq.z, cq.w}, vd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float top = va[k] * (1.f - lx) + vb[k] * lx, bot = vc[k] * (1.f - lx) + vd[k] * lx;
        o[(long long)(c + k) * plane] = top * (1.f - ly) + bot * ly;
      }
    }
  }
  for (; c < C; ++c) {
    const float top = r00[c] * (1.f - lx) + r01[c] * lx, bot = r10[c] * (1.f - lx) + r11[c] * lx;
    o[(long long)c * plane] = top * (1.f - ly) + bot * ly;
  }
}

// ---- depthwise ConvTranspose2d, kernel 2f, stride f, padding f/2 (IDAUp.up_i, centernet_model.py:256): every output pixel
// receives exactly 2 x 2 taps.  w: fp32 [C][2f][2f] (the master tensor), out = sum_{ky,kx} in[(oy + p - ky) / f][..] * w[c][ky][kx] ----
// The taps live in LDS as [tap][C] (a thread's 8 channels contiguous) when they fit 32 KB; the master layout [C][K][K] would cost 8 scattered
// 4-byte loads per tap.
__global__ __launch_bounds__(256) void dwconvt_kernel(ViewDesc in, ViewDesc out, const float* __restrict__ w, int B, int IH, int IW, int CG, int f) {
  __shared__ float sw[8192];
  const int OH = IH * f, OW = IW * f, K = 2 * f, P = f / 2, KK = K * K, C = CG * 8;
  const bool staged = C * KK <= 8192;
  if (staged) {
    for (int j = threadIdx.x; j < C * KK; j += 256) sw[(j % KK) * C + j / KK] = w[j];
    __syncthreads();
  }
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)B * OH * OW * CG;
  if (i >= n) return;
  int cg = (int)(i % CG);
  long long t = i / CG;
  int ox = (int)(t % OW);
  t /= OW;
  int oy = (int)(t % OH);
  int b = (int)(t / OH);
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  const int ky0 = (oy + P) % f, kx0 = (ox + P) % f;
  // the 2 x 2 taps with out-of-range ones clamped onto a valid pixel / tap and given the factor 0: the four pixel loads are unconditional
  // (issued together; a branch per tap waits for memory once per tap), an absent tap adds exactly +0
  int kys[2], kxs[2], iys[2], ixs[2];
  float my[2], mx[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int ky = ky0 + a * f, iy = (oy + P - ky) / f;  // exact
    const bool ok = ky < K && iy >= 0 && iy < IH;
    kys[a] = ok ? ky : ky0;
    iys[a] = ok ? iy : (iy < 0 ? 0 : IH - 1);
    my[a] = ok ? 1.f : 0.f;
    const int kx = kx0 + a * f, ix = (ox + P - kx) / f;
    const bool okx = kx < K && ix >= 0 && ix < IW;
    kxs[a] = okx ? kx : kx0;
    ixs[a] = okx ? ix : (ix < 0 ? 0 : IW - 1);
    mx[a] = okx ? 1.f : 0.f;
  }
  h8 v[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) v[a][c] = *reinterpret_cast<const h8*>(in.p + voff(in, b, (long long)iys[a] * IW + ixs[c]) + cg * 8);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float m = my[a] * mx[c];
      if (staged) {
        const float* wl = sw + (kys[a] * K + kxs[c]) * C + cg * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = fmaf((float)v[a][c][k], wl[k] * m, acc[k]);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = fmaf((float)v[a][c][k], w[((long long)(cg * 8 + k) * K + kys[a]) * K + kxs[c]] * m, acc[k]);
      }
    }
  h8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (half_t)acc[k];
  *reinterpret_cast<h8*>(out.p + voff(out, b, (long long)oy * OW + ox) + cg * 8) = o;
}

// ---- channel-slice copy (a tensor that must also live in a second concat buffer) ----
// gradients of the depthwise transposed convolution (IDAUp.up_i): out[oy][ox] takes in[iy][ix] * w[ky][kx] with oy = iy*f - P + ky.
// Data gradient: a gather over the K x K taps of every input pixel.
__global__ void dwconvt_dgrad_kernel(ViewDesc gout, ViewDesc gin, const float* w, int B, int IH, int IW, int CG, int f, int accumulate) {
  const int OH = IH * f, OW = IW * f, K = 2 * f, P = f / 2;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)B * IH * IW * CG;
  if (i >= n) return;
  int cg = (int)(i % CG);
  long long t = i / CG;
  int ix = (int)(t % IW);
  t /= IW;
  int iy = (int)(t % IH);
  int b = (int)(t / IH);
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  for (int ky = 0; ky < K; ++ky) {
    const int oy = iy * f - P + ky;
    if (oy < 0 || oy >= OH) continue;
    for (int kx = 0; kx < K; ++kx) {
      const int ox = ix * f - P + kx;
      if (ox < 0 || ox >= OW) continue;
      const h8 g = *reinterpret_cast<const h8*>(gout.p + voff(gout, b, (long long)oy * OW + ox) + cg * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = fmaf((float)g[k], w[((long long)(cg * 8 + k) * K + ky) * K + kx], acc[k]);
    }
  }
  half_t* q = gin.p + voff(gin, b, (long long)iy * IW + ix) + cg * 8;
  if (accumulate) {
    const h8 old = *reinterpret_cast<const h8*>(q);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += (float)old[k];
  }
  h8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (half_t)acc[k];
  *reinterpret_cast<h8*>(q) = o;
}
// Weight gradient: dw[c][ky][kx] += inv_scale * sum over (b, iy, ix) of in * gout at the tap's offset.  Two launches, deterministic:
// (1) a workgroup takes a run of input pixels, thread = (pixel lane, channel group) so that a pixel's channel row is one coalesced read,
// 16 taps per blockIdx.y in registers, lanes folded through LDS in a fixed order -> part[block][tap][C]; (2) the blocks are summed in
// index order.  With all taps in one group (f = 2, every up-layer of DLA-34 but one) the same pass also produces the data gradient
// (DGRAD): both walk gout at the same K x K offsets of an input pixel, so gout is read once instead of twice.
template <bool DGRAD>
__global__ __launch_bounds__(256) void dwconvt_wgrad_part_kernel(ViewDesc in, ViewDesc gout, ViewDesc gin, const float* __restrict__ w, float* part,
                                                                 int B, int IH, int IW, int CG, int f, int rows, int accumulate) {
  __shared__ float sacc[4 * 256 * 8];
  const int OH = IH * f, OW = IW * f, K = 2 * f, P = f / 2, KK = K * K, C = CG * 8;
  const int RP = 256 / CG;
  const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
  const int tap0 = blockIdx.y * 16;
  const long long npix = (long long)B * IH * IW;
  const long long p0 = (long long)blockIdx.x * rows, p1 = p0 + rows < npix ? p0 + rows : npix;
  float acc[16][8];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[t][k] = 0.f;
  if (DGRAD) {  // the 16 taps as [tap][C] in the LDS the fold uses afterwards (C <= 512, checked by the launcher)
    for (int j = threadIdx.x; j < C * 16; j += 256) sacc[(j & 15) * C + (j >> 4)] = w[j];
    __syncthreads();
  }
  if (pl < RP)
    for (long long p = p0 + pl; p < p1; p += RP) {
      const int ix = (int)(p % IW);
      const long long q = p / IW;
      const int iy = (int)(q % IH);
      const int b = (int)(q / IH);
      const h8 v = *reinterpret_cast<const h8*>(in.p + voff(in, b, (long long)iy * IW + ix) + cg * 8);
      float d[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) d[k] = 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int tap = tap0 + t;
        const int ky = tap / K, kx = tap - ky * K;
        const int oy = iy * f - P + ky, ox = ix * f - P + kx;
        if (oy < 0 || oy >= OH || ox < 0 || ox >= OW) continue;
        const h8 g = *reinterpret_cast<const h8*>(gout.p + voff(gout, b, (long long)oy * OW + ox) + cg * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[t][k] = fmaf((float)v[k], (float)g[k], acc[t][k]);
        if (DGRAD) {
          const float* wl = sacc + tap * C + cg * 8;
#pragma unroll
          for (int k = 0; k < 8; ++k) d[k] = fmaf((float)g[k], wl[k], d[k]);
        }
      }
      if (DGRAD) {
        half_t* gq = gin.p + voff(gin, b, (long long)iy * IW + ix) + cg * 8;
        if (accumulate) {
          const h8 old = *reinterpret_cast<const h8*>(gq);
#pragma unroll
          for (int k = 0; k < 8; ++k) d[k] += (float)old[k];
        }
        h8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (half_t)d[k];
        *reinterpret_cast<h8*>(gq) = o;
      }
    }
  if (DGRAD) __syncthreads();
  for (int r = 0; r < 4; ++r) {  // four taps per round through 32 KB of LDS
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) sacc[(j * 256 + threadIdx.x) * 8 + k] = acc[r * 4 + j][k];
    __syncthreads();
    for (int o = threadIdx.x; o < 4 * C; o += 256) {
      const int j = o / C, ch = o - j * C;
      float sum = 0.f;
      for (int l = 0; l < RP; ++l) sum += sacc[(j * 256 + l * CG + (ch >> 3)) * 8 + (ch & 7)];
      part[((long long)blockIdx.x * KK + tap0 + r * 4 + j) * C + ch] = sum;
    }
    __syncthreads();
  }
}
__global__ void dwconvt_wgrad_fin_kernel(const float* __restrict__ part, int nblk, int KK, int C, float inv_scale, float* dw) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;  // (tap, channel), channel fastest: coalesced over the partial rows
  if (o >= KK * C) return;
  const int tap = o / C, ch = o - tap * C;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int blk = 0;
  for (; blk + 4 <= nblk; blk += 4) {
    s0 += part[(long long)(blk + 0) * KK * C + o];
    s1 += part[(long long)(blk + 1) * KK * C + o];
    s2 += part[(long long)(blk + 2) * KK * C + o];
    s3 += part[(long long)(blk + 3) * KK * C + o];
  }
  for (; blk < nblk; ++blk) s0 += part[(long long)blk * KK * C + o];
  dw[(long long)ch * KK + tap] += ((s0 + s1) + (s2 + s3)) * inv_scale;
}

// gradient of conv + bias (+ ReLU) blocks without BatchNorm (CenterNet heads, SSD extras): dy = g * [out > 0] (relu) or g, dense fp16,
// and the bias gradient's column sums into the replica slabs (finalised by colsum_finalize)
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(ViewDesc gout, ViewDesc fout, int relu, long long M, int C, int hw, half_t* dy,
                                                           long long* part, int rows_per_block) {
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
      const unsigned mu = (unsigned)m, b = mu / (unsigned)hw, pix = mu - b * (unsigned)hw;
      h8 g = *reinterpret_cast<const h8*>(gout.p + voff(gout, (int)b, pix) + cg * 8);
      if (relu) {
        const h8 fo = *reinterpret_cast<const h8*>(fout.p + voff(fout, (int)b, pix) + cg * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (!((float)fo[i] > 0.f)) g[i] = (half_t)0.f;
      }
      *reinterpret_cast<h8*>(dy + m * C + cg * 8) = g;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[0][i] += (float)g[i];
    }
  }
  cvx_bn::block_channel_sums<1>(acc, C, CG, cg, active, sacc, part, blockIdx.x);
}

__global__ void copy_slice_kernel(ViewDesc in, ViewDesc out, int B, int HW, int CG) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)B * HW * CG;
  if (i >= n) return;
  int cg = (int)(i % CG);
  long long t = i / CG;
  int pix = (int)(t % HW);
  int b = (int)(t / HW);
  *reinterpret_cast<h8*>(out.p + voff(out, b, pix) + cg * 8) = *reinterpret_cast<const h8*>(in.p + voff(in, b, pix) + cg * 8);
}
__global__ void add_slice_kernel(ViewDesc in, ViewDesc out, int B, int HW, int CG) {  // out += in (gradient of a slice copy onto a written slice)
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)B * HW * CG;
  if (i >= n) return;
  int cg = (int)(i % CG);
  long long t = i / CG;
  int pix = (int)(t % HW);
  int b = (int)(t / HW);
  half_t* q = out.p + voff(out, b, pix) + cg * 8;
  const h8 a = *reinterpret_cast<const h8*>(in.p + voff(in, b, pix) + cg * 8), o = *reinterpret_cast<const h8*>(q);
  h8 r;
#pragma unroll
  for (int k = 0; k < 8; ++k) r[k] = (half_t)((float)a[k] + (float)o[k]);
  *reinterpret_cast<h8*>(q) = r;
}

// ---- max pool 5x5 s1 p2 ------------------------------------------------------------------------
__global__ void maxpool5_fwd_kernel(ViewDesc in, ViewDesc out, int B, int H, int W, int CG, uint8_t* idx) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)B * H * W * CG;
  if (i >= n) return;
  int cg = (int)(i % CG);
  long long t = i / CG;
  int w = (int)(t % W);
  t /= W;
  int h = (int)(t % H);
  int b = (int)(t / H);
  float best[8];
  int bi[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    best[k] = -INFINITY;
    bi[k] = 0;
  }
  bool first = true;
  // a window row at a time: five independent 16-byte loads from clamped (always valid) addresses, then the compares in
  // the reference's scan order -- no branch sits between a load and the next one, so they overlap
#pragma unroll
  for (int dy = 0; dy < 5; ++dy) {
    const int hh = h + dy - 2;
    const bool rok = hh >= 0 && hh < H;
    const int hc = min(max(hh, 0), H - 1);
    h8 v[5];
    bool ok[5];
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {
      const int ww = w + dx - 2;
      ok[dx] = rok && ww >= 0 && ww < W;
      const int wc = min(max(ww, 0), W - 1);
      v[dx] = *reinterpret_cast<const h8*>(in.p + voff(in, b, (long long)hc * W + wc) + cg * 8);
    }
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {
      if (!ok[dx]) continue;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float f = (float)v[dx][k];
        if (first || f > best[k]) {
          best[k] = f;
          bi[k] = dy * 5 + dx;
        }
      }
      first = false;
    }
  }
  h8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (half_t)best[k];
  long long pix = (long long)h * W + w;
  *reinterpret_cast<h8*>(out.p + voff(out, b, pix) + cg * 8) = o;
  if (idx) {
    unsigned long long pk = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) pk |= (unsigned long long)(bi[k] & 0xff) << (8 * k);
    *reinterpret_cast<unsigned long long*>(idx + (((long long)b * H * W + pix) * CG + cg) * 8) = pk;
  }
}

__global__ void maxpool5_bwd_kernel(ViewDesc gout, ViewDesc gin, int B, int H, int W, int CG, const uint8_t* idx, int accumulate) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)B * H * W * CG;
  if (i >= n) return;
  int cg = (int)(i % CG);
  long long t = i / CG;
  int w = (int)(t % W);
  t /= W;
  int h = (int)(t % H);
  int b = (int)(t / H);
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  // output (oh, ow) whose window contains (h, w): oh in [h-2, h+2]; its tap for this input is (h-oh+2, w-ow+2)
#pragma unroll
  for (int dy = 0; dy < 5; ++dy) {
    const int oh = h + dy - 2;
    const bool rok = oh >= 0 && oh < H;
    const int ohc = min(max(oh, 0), H - 1);
    unsigned long long pk[5];
    h8 g[5];
    bool ok[5];
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {  // loads from clamped addresses first, uses after (see the forward kernel)
      const int ow = w + dx - 2;
      ok[dx] = rok && ow >= 0 && ow < W;
      const long long pix = (long long)ohc * W + min(max(ow, 0), W - 1);
      pk[dx] = *reinterpret_cast<const unsigned long long*>(idx + (((long long)b * H * W + pix) * CG + cg) * 8);
      g[dx] = *reinterpret_cast<const h8*>(gout.p + voff(gout, b, pix) + cg * 8);
    }
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {
      if (!ok[dx]) continue;
      const int tapcode = (4 - dy) * 5 + (4 - dx);  // = (h - oh + 2) * 5 + (w - ow + 2)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if ((int)((pk[dx] >> (8 * k)) & 0xff) == tapcode) acc[k] += (float)g[dx][k];
    }
  }
  half_t* q = gin.p + voff(gin, b, (long long)h * W + w) + cg * 8;
  if (accumulate) {
    h8 old = *reinterpret_cast<const h8*>(q);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += (float)old[k];
  }
  h8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (half_t)acc[k];
  *reinterpret_cast<h8*>(q) = o;
}

// ---- SPPF: three chained 5x5 max pools in one launch ----------------------------------------------
// SPPF.forward (core/models/yolov8/modules.py:304-318) pools x three times, each result feeding the next: y1 = m(x), y2 = m(y1), y3 = m(y2).
// The maps are small (20 x 20 at 640 x 640 input), so the three launches were pure latency (46 + 35 + 35 us at batch 32).  Here a workgroup
// keeps the H x W map of one image and one 8-channel group in LDS, pools it three times (ping-pong) and writes every stage to its slice
// of the concat buffer (+ the argmax bytes in training).  Scan order and tie-breaking are those of maxpool5_fwd_kernel: same bits.
// One 5x5 pool as two separable passes with the 2-D scan's tie-breaking: the first maximum in (dy, dx) scan order is, among the rows in dy
// order, the first one whose row maximum equals the window maximum, and inside it the first dx that reaches it.  10 instead of 25 window
// reads and compare chains per pixel.
__device__ __forceinline__ void pool5_row(const h8* src, int h, int w, int W, h8* rm, unsigned* ri) {
  float best[8];
  int bi[8];
  bool first = true;
#pragma unroll
  for (int dx = 0; dx < 5; ++dx) {
    const int ww = w + dx - 2;
    const bool ok = ww >= 0 && ww < W;
    const h8 v = src[h * W + min(max(ww, 0), W - 1)];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float f = (float)v[k];
      const bool take = ok && (first || f > best[k]);
      best[k] = take ? f : (first ? -INFINITY : best[k]);
      bi[k] = take ? dx : (first ? 0 : bi[k]);
    }
    first = first && !ok;
  }
  h8 o;
  unsigned pk = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    o[k] = (half_t)best[k];
    pk |= (unsigned)bi[k] << (4 * k);
  }
  rm[h * W + w] = o;
  ri[h * W + w] = pk;
}
__device__ __forceinline__ h8 pool5_col(const h8* rm, const unsigned* ri, int h, int w, int H, int W, unsigned long long* pk_out) {
  float best[8];
  int bi[8];
  bool first = true;
#pragma unroll
  for (int dy = 0; dy < 5; ++dy) {
    const int hh = h + dy - 2;
    const bool ok = hh >= 0 && hh < H;
    const int q = min(max(hh, 0), H - 1) * W + w;
    const h8 v = rm[q];
    const unsigned di = ri[q];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float f = (float)v[k];
      const bool take = ok && (first || f > best[k]);
      best[k] = take ? f : (first ? -INFINITY : best[k]);
      bi[k] = take ? dy * 5 + (int)((di >> (4 * k)) & 7u) : (first ? 0 : bi[k]);
    }
    first = first && !ok;
  }
  h8 o;
  unsigned long long pk = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    o[k] = (half_t)best[k];
    pk |= (unsigned long long)(bi[k] & 0xff) << (8 * k);
  }
  *pk_out = pk;
  return o;
}

constexpr int SPPF_FWD_BYTES_PER_PIXEL = 16 + 16 + 4;  // stage map | row maxima | row argmax nibbles
constexpr int SPPF_BWD_BYTES_PER_PIXEL = 16 + 32;      // stage gradient (fp16) | fp32 accumulators of the stage's input gradient

__global__ __launch_bounds__(256) void sppf_pool3_fwd_kernel(ViewDesc in, ViewDesc o1, ViewDesc o2, ViewDesc o3, int H, int W, int CG, uint8_t* i1,
                                                             uint8_t* i2, uint8_t* i3) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sppf_lds[];
  const int HW = H * W;
  h8* cur = reinterpret_cast<h8*>(sppf_lds);
  h8* rm = cur + HW;
  unsigned* ri = reinterpret_cast<unsigned*>(rm + HW);
  const int b = blockIdx.x / CG, cg = blockIdx.x - b * CG;
  for (int p = threadIdx.x; p < HW; p += 256) cur[p] = *reinterpret_cast<const h8*>(in.p + voff(in, b, p) + cg * 8);
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const ViewDesc& out = s == 0 ? o1 : s == 1 ? o2 : o3;
    uint8_t* idx = s == 0 ? i1 : s == 1 ? i2 : i3;
    for (int p = threadIdx.x; p < HW; p += 256) {
      const int h = p / W;
      pool5_row(cur, h, p - h * W, W, rm, ri);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += 256) {
      const int h = p / W;
      unsigned long long pk;
      const h8 o = pool5_col(rm, ri, h, p - h * W, H, W, &pk);
      cur[p] = o;  // (the row pass of this stage is complete: the map can take the next stage's input)
      *reinterpret_cast<h8*>(out.p + voff(out, b, p) + cg * 8) = o;
      if (idx) *reinterpret_cast<unsigned long long*>(idx + (((long long)b * HW + p) * CG + cg) * 8) = pk;
    }
    __syncthreads();
  }
}

// The backward chain of the same three pools in one launch: g(y2) += route3(g(y3)), g(y1) += route2(g(y2)), g(x) (+)= route1(g(y1)), each stage
// rounded to fp16 like the tensors the three launches passed to each other.  Every output pixel sends its gradient to the ONE input pixel its
// argmax byte names (an fp32 LDS atomic per channel) instead of every input pixel searching the 25 outputs that could have picked it; the
// fp32 sums are the same up to the order of the additions.
__global__ __launch_bounds__(256) void sppf_pool3_bwd_kernel(ViewDesc g3, ViewDesc g2, ViewDesc g1, ViewDesc g0, int H, int W, int CG, const uint8_t* i1,
                                                             const uint8_t* i2, const uint8_t* i3, int acc_mask) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sppf_lds[];
  const int HW = H * W;
  h8* gcur = reinterpret_cast<h8*>(sppf_lds);              // gradient of the stage's output
  float* acc = reinterpret_cast<float*>(gcur + HW);        // [HW][8] gradient of its input, being summed
  const int b = blockIdx.x / CG, cg = blockIdx.x - b * CG;
  for (int p = threadIdx.x; p < HW; p += 256) gcur[p] = *reinterpret_cast<const h8*>(g3.p + voff(g3, b, p) + cg * 8);
#pragma unroll
  for (int s = 2; s >= 0; --s) {
    const uint8_t* idx = s == 2 ? i3 : s == 1 ? i2 : i1;
    const ViewDesc& gin = s == 2 ? g2 : s == 1 ? g1 : g0;
    const bool accumulate = ((acc_mask >> s) & 1) != 0;  // the stage's input gradient already holds the concat consumer's share
    for (int t = threadIdx.x; t < HW * 8; t += 256) acc[t] = 0.f;
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += 256) {
      const int h = p / W, w = p - h * W;
      const unsigned long long pk = *reinterpret_cast<const unsigned long long*>(idx + (((long long)b * HW + p) * CG + cg) * 8);
      const h8 g = gcur[p];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int tap = (int)((pk >> (8 * k)) & 0xff);
        const int dy = tap / 5, dx = tap - dy * 5;
        atomicAdd(&acc[((h + dy - 2) * W + (w + dx - 2)) * 8 + k], (float)g[k]);
      }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += 256) {
      half_t* q = gin.p + voff(gin, b, p) + cg * 8;
      float a[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] = acc[p * 8 + k];
      if (accumulate) {
        const h8 old = *reinterpret_cast<const h8*>(q);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] += (float)old[k];
      }
      h8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (half_t)a[k];
      gcur[p] = o;                         // the next stage's output gradient
      *reinterpret_cast<h8*>(q) = o;       // every stage's total goes back to its slice, like the three launches left them
    }
    __syncthreads();
  }
}

// ---- nearest 2x upsample -----------------------------------------------------------------------
__global__ void upsample2_fwd_kernel(ViewDesc in, ViewDesc out, int B, int H, int W, int CG) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over OUTPUT elements
  int OH = 2 * H, OW = 2 * W;
  long long n = (long long)B * OH * OW * CG;
  if (i >= n) return;
  int cg = (int)(i % CG);
  long long t = i / CG;
  int ow = (int)(t % OW);
  t /= OW;
  int oh = (int)(t % OH);
  int b = (int)(t / OH);
  h8 v = *reinterpret_cast<const h8*>(in.p + voff(in, b, (long long)(oh >> 1) * W + (ow >> 1)) + cg * 8);
  *reinterpret_cast<h8*>(out.p + voff(out, b, (long long)oh * OW + ow) + cg * 8) = v;
}

__global__ void upsample2_bwd_kernel(ViewDesc gout, ViewDesc gin, int B, int H, int W, int CG, int accumulate) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over INPUT elements
  long long n = (long long)B * H * W * CG;
  if (i >= n) return;
  int cg = (int)(i % CG);
  long long t = i / CG;
  int w = (int)(t % W);
  t /= W;
  int h = (int)(t % H);
  int b = (int)(t / H);
  int OW = 2 * W;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      h8 g = *reinterpret_cast<const h8*>(gout.p + voff(gout, b, (long long)(2 * h + dy) * OW + 2 * w + dx) + cg * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += (float)g[k];
    }
  half_t* q = gin.p + voff(gin, b, (long long)h * W + w) + cg * 8;
  if (accumulate) {
    h8 old = *reinterpret_cast<const h8*>(q);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += (float)old[k];
  }
  h8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (half_t)acc[k];
  *reinterpret_cast<h8*>(q) = o;
}

// ---- pred (B, A, no) <-> NCHW level tensors (API-compat path only) ------------------------------
__global__ void pred_to_nchw_kernel(const float* pred, int B, int A, int no, int a_off, int HW, float* out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over out elements (b, c, pix)
  long long n = (long long)B * no * HW;
  if (i >= n) return;
  int pix = (int)(i % HW);
  long long t = i / HW;
  int c = (int)(t % no);
  int b = (int)(t / no);
  out[i] = pred[((long long)b * A + a_off + pix) * no + c];
}
__global__ void nchw_to_pred_f16_kernel(const float* g, int B, int A, int no, int a_off, int HW, float scale, half_t* dpred) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over (b, pix, c)
  long long n = (long long)B * HW * no;
  if (i >= n) return;
  int c = (int)(i % no);
  long long t = i / no;
  int pix = (int)(t % HW);
  int b = (int)(t / HW);
  dpred[((long long)b * A + a_off + pix) * no + c] = (half_t)(g[((long long)b * no + c) * HW + pix] * scale);
}

// ---- weight shadow packing ------------------------------------------------------------------------
// block kinds of one launch: start >= 0 -- 1024 consecutive elements of the forward layout [Cout][T][Cin_pad] (fp32 -> fp16, padded
// input channels zero); start < 0 -- tile -(start + 1) of the transposed data-gradient layout [Cin][T][Cout]: a 32 x 32 (co, ci) tile of
// one tap goes through LDS so that both the fp32 reads (along ci) and the fp16 writes (along co) are contiguous (the element-wise
// scatter it replaces wrote 2 bytes per cache line: 0.8 ms for ResNet-101's 59 M parameters).
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* master, half_t* shadow, const PackDesc* descs, const BlockRef* blocks) {
  __shared__ float tile[32][33];
  const BlockRef br = blocks[blockIdx.x];
  const PackDesc d = descs[br.desc];
  if (br.start < 0) {
    const int lin = -(br.start + 1);
    const int tci = (d.Cin + 31) / 32, tco = (d.Cout + 31) / 32;
    const int t = lin / (tco * tci), rem = lin - t * (tco * tci);
    const int co0 = (rem / tci) * 32, ci0 = (rem % tci) * 32;
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = r0 + 8 * k;  // co within the tile
      const int co = co0 + r, ci = ci0 + c;
      tile[r][c] = (co < d.Cout && ci < d.Cin) ? master[d.src_off + ((long long)co * d.T + t) * d.Cin + ci] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = r0 + 8 * k;  // ci within the tile
      const int ci = ci0 + r, co = co0 + c;
      if (ci < d.Cin && co < d.Cout) shadow[d.dg_off + ((long long)ci * d.T + t) * d.Cout + co] = (half_t)tile[c][r];
    }
    return;
  }
  const int total = d.Cout * d.T * d.Cin_pad;
  for (int k = 0; k < 4; ++k) {
    int e = br.start + k * 256 + threadIdx.x;
    if (e >= total) return;
    int ci = e % d.Cin_pad;
    int row = e / d.Cin_pad;  // co*T + t
    float v = ci < d.Cin ? master[d.src_off + (long long)row * d.Cin + ci] : 0.f;
    shadow[d.fwd_off + e] = (half_t)v;
  }
}

__global__ __launch_bounds__(256) void ps_pack_kernel(const half_t* dg_base, half_t* ps_base, const PsPackDesc d) {
  const int k8 = (4 * d.C) / 8;  // 16-byte units per row
  const long long total = 4LL * d.cin_pad * k8;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int n = (int)(idx / k8), kk = (int)(idx - (long long)n * k8) * 8;
  const int phase = n / d.cin_pad, ci = n - phase * d.cin_pad;
  const int tau = kk / d.C, co = kk - tau * d.C;
  const int wt = d.wtap[phase * 4 + tau];
  h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (wt >= 0) v = *reinterpret_cast<const h8*>(dg_base + d.dg_off + ((long long)ci * d.T + wt) * d.C + co);
  *reinterpret_cast<h8*>(ps_base + d.ps_off + (long long)n * 4 * d.C + kk) = v;
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* slabs, float* grads, float inv_scale, const SlabDesc* descs,
                                                           const BlockRef* blocks) {
  // block = (256 / lanes) consecutive gradient elements x `lanes` split-lanes; each lane sums a strided
  // subset of the splits, then a fixed-order LDS tree adds the lanes (deterministic).
  __shared__ float sred[256];
  const BlockRef br = blocks[blockIdx.x];
  const SlabDesc d = descs[br.desc];
  const long long total = (long long)d.rows * d.Cin;
  const long long slab_elems = (long long)d.rows * d.Cin_pad;
  const int LR = d.lanes, COLS = 256 / LR;
  const int col = threadIdx.x % COLS, rl = threadIdx.x / COLS;
  const long long e = (long long)br.start + col;
  float acc = 0.f;
  if (e < total) {
    int ci = (int)(e % d.Cin);
    long long row = e / d.Cin;
    const float* s = slabs + d.slab_off + row * d.Cin_pad + ci;
    // four splits per trip, their loads issued together and added in the old order (one dependent 4-byte load per trip ran the pass at
    // 1.7 TB/s: nsplit / LR serial memory latencies per thread)
    int sp = rl;
    for (; sp + 3 * LR < d.nsplit; sp += 4 * LR) {
      const float a0 = s[(long long)sp * slab_elems], a1 = s[(long long)(sp + LR) * slab_elems];
      const float a2 = s[(long long)(sp + 2 * LR) * slab_elems], a3 = s[(long long)(sp + 3 * LR) * slab_elems];
      acc = (((acc + a0) + a1) + a2) + a3;
    }
    for (; sp < d.nsplit; sp += LR) acc += s[(long long)sp * slab_elems];
  }
  sred[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off >= COLS; off >>= 1) {
    if ((int)threadIdx.x < off) sred[threadIdx.x] += sred[threadIdx.x + off];
    __syncthreads();
  }
  if (rl == 0 && e < total) grads[d.dst_off + e] += sred[col] * inv_scale;
}

// ---- Adam ---------------------------------------------------------------------------------------
// state[0] = lr (host-written), state[1] = step count, state[2] = lr/(1-b1^t), state[3] = 1/sqrt(1-b2^t)
__global__ void adam_advance_kernel(float* state, float b1, float b2, const int* found_inf) {
  if (found_inf && *found_inf != 0) return;  // a skipped step does not advance the bias correction
  const float t = state[1] + 1.f;
  state[1] = t;
  state[2] = state[0] / (1.f - powf(b1, t));
  state[3] = rsqrtf(1.f - powf(b2, t));
}

__global__ __launch_bounds__(256) void adam_kernel(float* p, float* g, float* m, float* v, long long n, float lr_over_bc1, float b1, float b2,
                                                   float eps, float inv_sqrt_bc2, const int* found_inf, int zero_grad, const float* state, float gscale) {
  const bool skip = found_inf && *found_inf != 0;
  if (state) {  // device-resident step state (hipGraph replay: no host-side arguments change between steps)
    lr_over_bc1 = state[2];
    inv_sqrt_bc2 = state[3];
  }
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 4 <= n) {
    f4 gg = *reinterpret_cast<f4*>(g + i);
#pragma unroll
    for (int k = 0; k < 4; ++k) gg[k] *= gscale;  // 1/world_size after a SUM all-reduce (1 otherwise)
    if (!skip) {
      f4 pp = *reinterpret_cast<f4*>(p + i), mm = *reinterpret_cast<f4*>(m + i), vv = *reinterpret_cast<f4*>(v + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        mm[k] = b1 * mm[k] + (1.f - b1) * gg[k];
        vv[k] = b2 * vv[k] + (1.f - b2) * gg[k] * gg[k];
        pp[k] -= lr_over_bc1 * mm[k] / (sqrtf(vv[k]) * inv_sqrt_bc2 + eps);
      }
      *reinterpret_cast<f4*>(p + i) = pp;
      *reinterpret_cast<f4*>(m + i) = mm;
      *reinterpret_cast<f4*>(v + i) = vv;
    }
    if (zero_grad) *reinterpret_cast<f4*>(g + i) = f4{0.f, 0.f, 0.f, 0.f};
  } else {
    for (; i < n; ++i) {
      float gg = g[i] * gscale;
      if (!skip) {
        float mm = b1 * m[i] + (1.f - b1) * gg;
        float vv = b2 * v[i] + (1.f - b2) * gg * gg;
        m[i] = mm;
        v[i] = vv;
        p[i] -= lr_over_bc1 * mm / (sqrtf(vv) * inv_sqrt_bc2 + eps);
      }
      if (zero_grad) g[i] = 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void check_finite_kernel(const float* g, long long n, int* found_inf) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  bool bad = false;
  for (; i < n; i += stride) {
    float x = g[i];
    bad |= !(fabsf(x) <= 3.0e38f);
  }
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) *found_inf = 1;
}

template <typename K, typename... Args>
int launch1d(K kern, long long n, hipStream_t st, Args... args) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(kern, dim3(cvx_cdiv(n, 256)), dim3(256), 0, st, args...);
  CVX_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int cvx_image_to_nhwc8(const float* img, int B, int H, int W, half_t* out, hipStream_t st) {
  return launch1d(image_to_nhwc8_kernel, (long long)B * H * W, st, img, B, H * W, out);
}
int cvx_maxpool2(const ViewDesc& in, const ViewDesc& out, int B, int IH, int IW, int OH, int OW, int C, hipStream_t st) {
  CVX_CHECK((OH == IH / 2 || OH == (IH + 1) / 2) && (OW == IW / 2 || OW == (IW + 1) / 2), "maxpool2: output size must be floor or ceil of half the input");
  return launch1d(maxpool2_kernel, (long long)B * OH * OW * (C / 8), st, in, out, B, IH, IW, OH, OW, C / 8);
}
int cvx_maxpool2_bwd(const ViewDesc& in, const ViewDesc& gout, const ViewDesc& gin, int B, int IH, int IW, int OH, int OW, int C, int accumulate,
                     hipStream_t st) {
  CVX_CHECK(C % 8 == 0 && (OH == IH / 2 || OH == (IH + 1) / 2) && (OW == IW / 2 || OW == (IW + 1) / 2), "maxpool2_bwd: C % 8 / output size");
  return launch1d(maxpool2_bwd_kernel, (long long)B * ((IH + 1) / 2) * ((IW + 1) / 2) * (C / 8), st, in, gout, gin, B, IH, IW, OH, OW, C / 8, accumulate);
}
int cvx_l2norm(const ViewDesc& in, const ViewDesc& out, const float* weight, int B, int HW, int C, hipStream_t st) {
  CVX_CHECK(C % 8 == 0 && weight, "l2norm: C % 8 / weight");
  const long long npix = (long long)B * HW;
  hipLaunchKernelGGL(l2norm_kernel, dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, st, in, out, weight, npix, HW, C);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_l2norm_bwd_blocks(long long npix) { return (int)std::min<long long>(1024, (npix + 3) / 4); }
int cvx_l2norm_bwd(const ViewDesc& in, const ViewDesc& gout, const ViewDesc& gin, const float* weight, float* dweight, float inv_scale, int B, int HW,
                   int C, int accumulate, float* partial, hipStream_t st) {
  CVX_CHECK(C % 8 == 0 && C <= 512 && weight && dweight && partial, "l2norm_bwd: C % 8, C <= 512, weight / scratch");
  const long long npix = (long long)B * HW;
  const int nblocks = cvx_l2norm_bwd_blocks(npix);
  const int per_wave = (int)((npix + (long long)nblocks * 4 - 1) / ((long long)nblocks * 4));
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(nblocks), dim3(256), 0, st, in, gout, gin, weight, npix, HW, C, per_wave, accumulate, partial);
  hipLaunchKernelGGL(l2norm_dw_kernel, dim3((C + 15) / 16), dim3(256), 0, st, partial, nblocks, C, inv_scale, dweight);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_maxpool3(const ViewDesc& in, const ViewDesc& out, int B, int IH, int IW, int C, int stride, uint8_t* idx, hipStream_t st) {
  CVX_CHECK(C % 8 == 0 && (stride == 1 || stride == 2), "maxpool3: C % 8, stride 1 or 2");
  const int OH = (IH - 1) / stride + 1, OW = (IW - 1) / stride + 1;  // floor((I + 2 - 3) / stride) + 1
  return launch1d(maxpool3_kernel, (long long)B * OH * OW * (C / 8), st, in, out, B, IH, IW, OH, OW, C / 8, stride, idx);
}
int cvx_maxpool3_bwd(const ViewDesc& gout, const ViewDesc& gin, int B, int IH, int IW, int C, int stride, const uint8_t* idx, int accumulate,
                     hipStream_t st) {
  CVX_CHECK(C % 8 == 0 && (stride == 1 || stride == 2) && idx, "maxpool3_bwd: C % 8, stride 1 or 2, argmax bytes of the forward");
  const int OH = (IH - 1) / stride + 1, OW = (IW - 1) / stride + 1;
  return launch1d(maxpool3_bwd_kernel, (long long)B * IH * IW * (C / 8), st, gout, gin, B, IH, IW, OH, OW, C / 8, stride, idx, accumulate);
}
int cvx_zero_slice(const ViewDesc& v, int B, int HW, int C, hipStream_t st) {
  CVX_CHECK(C % 8 == 0, "zero_slice: C % 8");
  return launch1d(zero_slice_kernel, (long long)B * HW * (C / 8), st, v, B, HW, C / 8);
}
int cvx_avgpool_global_bwd(const ViewDesc& gout, const ViewDesc& gin, int B, int HW, int C, int accumulate, hipStream_t st) {
  CVX_CHECK(C % 8 == 0 && HW > 0, "avgpool_global_bwd: C % 8");
  return launch1d(avgpool_global_bwd_kernel, (long long)B * HW * (C / 8), st, gout, gin, B, HW, C / 8, accumulate);
}
int cvx_resize_bilinear_bwd(const ViewDesc& gout, const ViewDesc& gin, int B, int IH, int IW, int OH, int OW, int C, int accumulate,
                            hipStream_t st) {
  CVX_CHECK(C % 8 == 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, "resize_bilinear_bwd: C % 8 / sizes");
  if (IH == 1 && IW == 1) {
    hipLaunchKernelGGL(resize_from_1x1_bwd_kernel, dim3(B * (C / 8)), dim3(256), 0, st, gout, gin, OH * OW, C / 8, accumulate);
    CVX_HIP(hipGetLastError());
    return 0;
  }
  return launch1d(resize_bilinear_bwd_kernel, (long long)B * IH * IW * (C / 8), st, gout, gin, B, IH, IW, OH, OW, C / 8, (float)IH / (float)OH,
                  (float)IW / (float)OW, accumulate);
}
int cvx_dropout(const ViewDesc& in, const ViewDesc& out, int B, int HW, int C, float p, unsigned long long seed, int accumulate, hipStream_t st) {
  CVX_CHECK(C % 8 == 0 && p >= 0.f && p < 1.f, "dropout: C % 8, p in [0, 1)");
  const unsigned thresh = (unsigned)((double)p * 4294967296.0);
  return launch1d(dropout_kernel, (long long)B * HW * (C / 8), st, in, out, B, HW, C / 8, thresh, 1.f / (1.f - p), seed, accumulate);
}
int cvx_avgpool_global(const ViewDesc& in, const ViewDesc& out, int B, int HW, int C, hipStream_t st) {
  CVX_CHECK(C % 8 == 0 && HW > 0, "avgpool_global: C % 8");
  hipLaunchKernelGGL(avgpool_global_kernel, dim3(B * (C / 8)), dim3(256), 0, st, in, out, HW, C / 8);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_resize_bilinear(const ViewDesc& in, const ViewDesc& out, int B, int IH, int IW, int OH, int OW, int C, hipStream_t st) {
  CVX_CHECK(C % 8 == 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, "resize_bilinear: C % 8 / sizes");
  return launch1d(resize_bilinear_kernel, (long long)B * OH * OW * (C / 8), st, in, out, B, IH, IW, OH, OW, C / 8, (float)IH / (float)OH,
                  (float)IW / (float)OW);
}
int cvx_resize_bilinear_f32_nchw(const float* in, int ld, int B, int C, int IH, int IW, int OH, int OW, float* out, hipStream_t st) {
  CVX_CHECK(in && out && C > 0 && C <= ld && IH > 0 && IW > 0 && OH > 0 && OW > 0, "resize_bilinear_f32_nchw: bad arguments");
  return launch1d(resize_bilinear_f32_nchw_kernel, (long long)B * OH * OW, st, in, ld, B, C, IH, IW, OH, OW, (float)IH / (float)OH,
                  (float)IW / (float)OW, out);
}
int cvx_dwconvt(const ViewDesc& in, const ViewDesc& out, const float* w, int B, int IH, int IW, int C, int f, hipStream_t st) {
  CVX_CHECK(f >= 2 && f % 2 == 0, "dwconvt: stride must be even (kernel 2f, padding f/2)");
  return launch1d(dwconvt_kernel, (long long)B * IH * f * IW * f * (C / 8), st, in, out, w, B, IH, IW, C / 8, f);
}
int cvx_copy_slice(const ViewDesc& in, const ViewDesc& out, int B, int HW, int C, hipStream_t st) {
  return launch1d(copy_slice_kernel, (long long)B * HW * (C / 8), st, in, out, B, HW, C / 8);
}
int cvx_add_slice(const ViewDesc& in, const ViewDesc& out, int B, int HW, int C, hipStream_t st) {
  return launch1d(add_slice_kernel, (long long)B * HW * (C / 8), st, in, out, B, HW, C / 8);
}
// partial-sum geometry of the weight gradient: rows of input pixels per workgroup (a multiple of the pixel lanes), at most 256 workgroups
static int dwconvt_rows(long long npix, int C) {
  const int RP = 256 / (C / 8);
  long long rows = (npix + 255) / 256;
  rows = std::max<long long>(rows, 4LL * RP);
  return (int)((rows + RP - 1) / RP * RP);
}
long long cvx_dwconvt_bwd_scratch_floats(long long npix, int C, int f) {
  const int rows = dwconvt_rows(npix, C);
  return (npix + rows - 1) / rows * 4LL * f * f * C;
}
int cvx_dwconvt_bwd(const ViewDesc& in, const ViewDesc& gout, const ViewDesc& gin, const float* w, float* dw, float inv_scale, int B, int IH, int IW,
                    int C, int f, int accumulate, float* part, hipStream_t st) {
  CVX_CHECK(f >= 2 && f % 2 == 0 && C % 8 == 0 && C <= 2048 && part, "dwconvt_bwd: even stride, C % 8, C <= 2048, scratch");
  const long long npix = (long long)B * IH * IW;
  const int rows = dwconvt_rows(npix, C), nblk = (int)((npix + rows - 1) / rows), KK = 4 * f * f;
  if (KK == 16 && C <= 512) {
    hipLaunchKernelGGL(dwconvt_wgrad_part_kernel<true>, dim3(nblk, 1), dim3(256), 0, st, in, gout, gin, w, part, B, IH, IW, C / 8, f, rows, accumulate);
  } else {
    CVX_TRY(launch1d(dwconvt_dgrad_kernel, npix * (C / 8), st, gout, gin, w, B, IH, IW, C / 8, f, accumulate));
    hipLaunchKernelGGL(dwconvt_wgrad_part_kernel<false>, dim3(nblk, KK / 16), dim3(256), 0, st, in, gout, gin, w, part, B, IH, IW, C / 8, f, rows, 0);
  }
  hipLaunchKernelGGL(dwconvt_wgrad_fin_kernel, dim3((KK * C + 255) / 256), dim3(256), 0, st, part, nblk, KK, C, inv_scale, dw);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_bias_act_bwd(const ViewDesc& gout, const ViewDesc& fout, int relu, long long M, int C, int hw, half_t* dy, long long* part, float inv_scale,
                     float* dbias, hipStream_t st) {
  CVX_CHECK(C % 8 == 0 && C >= 8 && C <= CVX_BN_MAX_C && M < (1LL << 32) && (!relu || fout.p), "bias_act_bwd: C % 8, C <= 2048; ReLU needs the output");
  const int rows = cvx_stream_rows_per_block(M, C, 32);
  hipLaunchKernelGGL(bias_act_bwd_kernel, dim3((unsigned)((M + rows - 1) / rows)), dim3(256), 0, st, gout, fout, relu, M, C, hw, dy, part, rows);
  CVX_HIP(hipGetLastError());
  return cvx_colsum_finalize(part, C, inv_scale, dbias, st);
}
int cvx_maxpool5_fwd(const ViewDesc& in, const ViewDesc& out, int B, int H, int W, int C, uint8_t* idx, hipStream_t st) {
  CVX_CHECK(C % 8 == 0, "maxpool5: C % 8");
  return launch1d(maxpool5_fwd_kernel, (long long)B * H * W * (C / 8), st, in, out, B, H, W, C / 8, idx);
}
// maps the fused SPPF kernels hold in LDS (36 bytes per pixel forward, 48 backward)
bool cvx_sppf_pool3_fits(int H, int W) { return (long long)H * W * SPPF_BWD_BYTES_PER_PIXEL <= 60 * 1024; }
int cvx_sppf_pool3_fwd(const ViewDesc& in, const ViewDesc& o1, const ViewDesc& o2, const ViewDesc& o3, int B, int H, int W, int C, uint8_t* i1, uint8_t* i2,
                       uint8_t* i3, hipStream_t st) {
  CVX_CHECK(C % 8 == 0 && cvx_sppf_pool3_fits(H, W), "sppf_pool3: C % 8, map size");
  hipLaunchKernelGGL(sppf_pool3_fwd_kernel, dim3(B * (C / 8)), dim3(256), (size_t)H * W * SPPF_FWD_BYTES_PER_PIXEL, st, in, o1, o2, o3, H, W, C / 8, i1, i2, i3);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_sppf_pool3_bwd(const ViewDesc& g3, const ViewDesc& g2, const ViewDesc& g1, const ViewDesc& g0, int B, int H, int W, int C, const uint8_t* i1,
                       const uint8_t* i2, const uint8_t* i3, int acc_mask, hipStream_t st) {
  CVX_CHECK(C % 8 == 0 && cvx_sppf_pool3_fits(H, W) && i1 && i2 && i3, "sppf_pool3_bwd: C % 8, map size, idx");
  hipLaunchKernelGGL(sppf_pool3_bwd_kernel, dim3(B * (C / 8)), dim3(256), (size_t)H * W * SPPF_BWD_BYTES_PER_PIXEL, st, g3, g2, g1, g0, H, W, C / 8, i1, i2, i3, acc_mask);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_maxpool5_bwd(const ViewDesc& gout, const ViewDesc& gin, int B, int H, int W, int C, const uint8_t* idx, int accumulate, hipStream_t st) {
  CVX_CHECK(C % 8 == 0 && idx, "maxpool5_bwd: C % 8 / idx");
  return launch1d(maxpool5_bwd_kernel, (long long)B * H * W * (C / 8), st, gout, gin, B, H, W, C / 8, idx, accumulate);
}
int cvx_upsample2_fwd(const ViewDesc& in, const ViewDesc& out, int B, int H, int W, int C, hipStream_t st) {
  CVX_CHECK(C % 8 == 0, "upsample2: C % 8");
  return launch1d(upsample2_fwd_kernel, (long long)B * 4 * H * W * (C / 8), st, in, out, B, H, W, C / 8);
}
int cvx_upsample2_bwd(const ViewDesc& gout, const ViewDesc& gin, int B, int H, int W, int C, int accumulate, hipStream_t st) {
  CVX_CHECK(C % 8 == 0, "upsample2_bwd: C % 8");
  return launch1d(upsample2_bwd_kernel, (long long)B * H * W * (C / 8), st, gout, gin, B, H, W, C / 8, accumulate);
}
__global__ void pred_cols_to_nchw_kernel(const float* rows, int ld, int col0, int C, int B, int A, int a_off, int HW, float* out,
                                         long long out_bstride, long long out_off) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * C * HW) return;
  const int pix = (int)(i % HW);
  long long t = i / HW;
  const int c = (int)(t % C);
  const int b = (int)(t / C);
  out[(long long)b * out_bstride + out_off + (long long)c * HW + pix] = rows[((long long)b * A + a_off + pix) * ld + col0 + c];
}
// adjoint of pred_cols_to_nchw: a gradient in the flattened NCHW order of one level -> scale * gradient on the rows' columns (fp16)
__global__ void nchw_cols_grad_to_pred_kernel(const float* g, long long g_bstride, long long g_off, int C, int B, int A, int a_off, int HW, float scale,
                                              half_t* dpred, int ld, int col0) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * C * HW) return;
  const int c = (int)(i % C);
  long long t = i / C;
  const int pix = (int)(t % HW);
  const int b = (int)(t / HW);
  dpred[((long long)b * A + a_off + pix) * ld + col0 + c] = (half_t)(g[(long long)b * g_bstride + g_off + (long long)c * HW + pix] * scale);
}
int cvx_nchw_cols_grad_to_pred_launch(const float* g, long long g_bstride, long long g_off, int C, int B, int A, int a_off, int HW, float scale,
                                      half_t* dpred, int ld, int col0, hipStream_t st) {
  return launch1d(nchw_cols_grad_to_pred_kernel, (long long)B * C * HW, st, g, g_bstride, g_off, C, B, A, a_off, HW, scale, dpred, ld, col0);
}
int cvx_pred_cols_to_nchw_launch(const float* rows, int ld, int col0, int C, int B, int A, int a_off, int HW, float* out, long long out_bstride,
                                 long long out_off, hipStream_t st) {
  return launch1d(pred_cols_to_nchw_kernel, (long long)B * C * HW, st, rows, ld, col0, C, B, A, a_off, HW, out, out_bstride, out_off);
}
int cvx_pred_to_nchw(const float* pred, int B, int A, int no, int a_off, int H, int W, float* out, hipStream_t st) {
  return launch1d(pred_to_nchw_kernel, (long long)B * no * H * W, st, pred, B, A, no, a_off, H * W, out);
}
int cvx_nchw_to_pred_f16(const float* g, int B, int A, int no, int a_off, int H, int W, float scale, half_t* dpred, hipStream_t st) {
  return launch1d(nchw_to_pred_f16_kernel, (long long)B * no * H * W, st, g, B, A, no, a_off, H * W, scale, dpred);
}
int cvx_pack_weights(const float* master, half_t* shadow, const PackDesc* descs, const BlockRef* blocks, int nblocks, hipStream_t st) {
  if (nblocks <= 0) return 0;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(nblocks), dim3(256), 0, st, master, shadow, descs, blocks);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_pack_ps_weights(const half_t* dg_base, half_t* ps_base, const PsPackDesc& d, hipStream_t st) {
  CVX_CHECK(d.C % 8 == 0 && d.dg_off % 8 == 0 && d.ps_off % 8 == 0, "ps pack: 16-byte rows needed");
  const long long total = 4LL * d.cin_pad * (4 * d.C / 8);
  hipLaunchKernelGGL(ps_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dg_base, ps_base, d);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_reduce_slabs(const float* slabs, float* grads, float inv_scale, const SlabDesc* descs, const BlockRef* blocks, int nblocks,
                     hipStream_t st) {
  if (nblocks <= 0) return 0;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(nblocks), dim3(256), 0, st, slabs, grads, inv_scale, descs, blocks);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_adam(float* p, float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, int step, const int* found_inf,
             int zero_grad, hipStream_t st) {
  CVX_CHECK(step >= 1, "adam: step starts at 1");
  double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
  return launch1d(adam_kernel, (n + 3) / 4, st, p, g, m, v, n, (float)(lr / bc1), b1, b2, eps, (float)(1.0 / sqrt(bc2)), found_inf, zero_grad,
                  (const float*)nullptr, 1.0f);
}
int cvx_adam_dev(float* p, float* g, float* m, float* v, long long n, float b1, float b2, float eps, float* state, const int* found_inf,
                 int zero_grad, float grad_scale, hipStream_t st) {
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, st, state, b1, b2, found_inf);
  return launch1d(adam_kernel, (n + 3) / 4, st, p, g, m, v, n, 0.f, b1, b2, eps, 0.f, found_inf, zero_grad, (const float*)state, grad_scale);
}
int cvx_check_finite_launch(const float* g, long long n, int* found_inf, hipStream_t st) {
  if (n <= 0) return 0;
  int blocks = (int)std::min<long long>(1024, (n + 255) / 256);
  hipLaunchKernelGGL(check_finite_kernel, dim3(blocks), dim3(256), 0, st, g, n, found_inf);
  CVX_HIP(hipGetLastError());
  return 0;
}
