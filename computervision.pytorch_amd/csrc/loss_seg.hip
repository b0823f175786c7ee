// Segmentation loss of the DeepLabv3+ trainer on gfx950: bilinear upsampling of the decoder's logits to the label resolution,
// cross-entropy / focal loss per pixel, and the gradient w.r.t. the LOW-RESOLUTION logits rows the engine's backward pass starts
// from -- value and gradient in one pass chain, no autograd tape and no (B, nc, H, W) logits tensor in HBM.
//
// Reference semantics (file:line under the reference tree):
//   DeeplabV3Plus.forward        core/models/deeplabv3plus.py:143-148   x = F.interpolate(classifier(features), size=input, bilinear,
//                                                                       align_corners=False)
//   FocalLoss.forward            core/loss/focal_loss.py:14-22          ce = F.cross_entropy(x, t, ignore_index, reduction="none");
//                                                                       pt = exp(-ce); loss = alpha * (1 - pt)^gamma * ce; mean()
//   DeeplabV3PlusA.build_loss    core/algorithms/segmentation_2d.py:59-64  "focal" -> FocalLoss() (alpha 0.25, gamma 2, ignore -100),
//                                                                       "ce" -> nn.CrossEntropyLoss(reduction="mean")
//   train_loop                   core/trainer/segmentation_trainer.py:114-131
//
//   K1 seg_loss_pixel   one thread per label pixel: interpolate the nc logits from the four source rows (fp32, the forward
//                       kernel's arithmetic), log-softmax, loss value -> block partial sums, d loss / d logit (times
//                       grad_scale / normaliser) -> dlogits (B, nc, OH, OW) fp32
//   K2 resize_grad_rows adjoint of the bilinear resize as a gather (deterministic): one thread per (image, class, source pixel)
//                       collects the label pixels that interpolate from it -> dpred (B, ih*iw, ld) fp16
//   K3 finalize         loss = sum / normaliser
#include "cvx_common.h"
#include "../../include/cvx_engine.h"

namespace {

constexpr int SEG_MAX_NC = 4096;  // sanity bound only: the kernels loop over the classes

__device__ __forceinline__ void bilinear_src(int d, float scale, int in_size, int* i0, int* i1, float* lam) {
  float s = ((float)d + 0.5f) * scale - 0.5f;
  if (s < 0.f) s = 0.f;
  int a = (int)s;
  if (a > in_size - 1) a = in_size - 1;
  *i0 = a;
  *i1 = a + (a < in_size - 1 ? 1 : 0);
  *lam = s - (float)a;
}

// mode 0: focal (alpha, gamma), mean over ALL pixels (ignored ones count in the denominator: focal_loss.mean());
// mode 1: cross-entropy, mean over the non-ignored pixels (nn.CrossEntropyLoss(reduction="mean")): normalised by K3 / K2's scale
__global__ __launch_bounds__(256) void seg_loss_pixel_kernel(const float* rows, int ld, int B, int nc, int ih, int iw, int OH, int OW,
                                                             const long long* target, int mode, float alpha, float gamma, long long ignore_index,
                                                             float* dlogits, double* part /* per block: loss sum, valid pixels */, int* bad) {
  __shared__ double s_sum[4], s_cnt[4];
  const long long n = (long long)B * OH * OW;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  double my_loss = 0.0, my_cnt = 0.0;
  if (i < n) {
    const int ox = (int)(i % OW);
    long long t = i / OW;
    const int oy = (int)(t % OH);
    const int b = (int)(t / OH);
    int y0, y1, x0, x1;
    float ly, lx;
    bilinear_src(oy, (float)ih / (float)OH, ih, &y0, &y1, &ly);
    bilinear_src(ox, (float)iw / (float)OW, iw, &x0, &x1, &lx);
    const float* base = rows + (long long)b * ih * iw * ld;
    const float* r00 = base + ((long long)y0 * iw + x0) * ld;
    const float* r01 = base + ((long long)y0 * iw + x1) * ld;
    const float* r10 = base + ((long long)y1 * iw + x0) * ld;
    const float* r11 = base + ((long long)y1 * iw + x1) * ld;
    auto logit = [&](int c) {
      const float top = r00[c] * (1.f - lx) + r01[c] * lx, bot = r10[c] * (1.f - lx) + r11[c] * lx;
      return top * (1.f - ly) + bot * ly;
    };
    const long long tg = target[i];
    const bool ignored = tg == ignore_index;
    const bool valid = !ignored && tg >= 0 && tg < nc;
    if (!ignored && !valid) atomicOr(bad, 1);  // torch raises a device assert here
    // pass 1: running maximum and rescaled sum of exponentials (no per-thread array: the class count is a run-time value)
    // rows with ld % 8 == 0 (the engine's: 16-byte aligned, padded to 8 columns) are read eight classes at a time with 16-byte loads
    // issued together; a class loop of scalar loads waits for memory 4 * nc times per pass
    const bool vec = (ld & 7) == 0 && (reinterpret_cast<uintptr_t>(rows) & 15) == 0;
    auto logits8 = [&](int c0, float* z) {
      float a[8], bq[8], cq[8], d[8];
      *reinterpret_cast<float4*>(a) = *reinterpret_cast<const float4*>(r00 + c0);
      *reinterpret_cast<float4*>(a + 4) = *reinterpret_cast<const float4*>(r00 + c0 + 4);
      *reinterpret_cast<float4*>(bq) = *reinterpret_cast<const float4*>(r01 + c0);
      *reinterpret_cast<float4*>(bq + 4) = *reinterpret_cast<const float4*>(r01 + c0 + 4);
      *reinterpret_cast<float4*>(cq) = *reinterpret_cast<const float4*>(r10 + c0);
      *reinterpret_cast<float4*>(cq + 4) = *reinterpret_cast<const float4*>(r10 + c0 + 4);
      *reinterpret_cast<float4*>(d) = *reinterpret_cast<const float4*>(r11 + c0);
      *reinterpret_cast<float4*>(d + 4) = *reinterpret_cast<const float4*>(r11 + c0 + 4);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float top = a[k] * (1.f - lx) + bq[k] * lx, bot = cq[k] * (1.f - lx) + d[k] * lx;
        z[k] = top * (1.f - ly) + bot * ly;
      }
    };
    float zmax = -INFINITY, se = 0.f, zt = 0.f;
    auto online = [&](int c, float z) {
      if (c == (int)tg) zt = z;
      if (z > zmax) {
        se = se * expf(zmax - z) + 1.f;
        zmax = z;
      } else {
        se += expf(z - zmax);
      }
    };
    if (vec) {
      for (int c0 = 0; c0 < nc; c0 += 8) {
        float z[8];
        logits8(c0, z);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (c0 + k < nc) online(c0 + k, z[k]);
      }
    } else {
      for (int c = 0; c < nc; ++c) online(c, logit(c));
    }
    const float lse = zmax + logf(se);
    float coef = 0.f;  // d loss / d ce
    if (valid) {
      const float ce = lse - zt;
      if (mode == 0) {
        const float pt = expf(-ce);
        const float om = fmaxf(1.f - pt, 0.f);
        const float w = powf(om, gamma);
        my_loss = (double)(alpha * w * ce);
        // d/dce [alpha (1 - pt)^gamma ce], d pt / d ce = -pt
        coef = alpha * (w + (gamma > 0.f ? gamma * powf(om, gamma - 1.f) * pt * ce : 0.f));
      } else {
        my_loss = (double)ce;
        coef = 1.f;
      }
      my_cnt = 1.0;
    }
    float* dl = dlogits + (long long)b * nc * OH * OW + (long long)oy * OW + ox;
    auto emit = [&](int c, float z) {
      const float p = expf(z - lse);
      dl[(long long)c * OH * OW] = valid ? coef * (p - (c == (int)tg ? 1.f : 0.f)) : 0.f;
    };
    if (vec) {
      for (int c0 = 0; c0 < nc; c0 += 8) {
        float z[8];
        logits8(c0, z);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (c0 + k < nc) emit(c0 + k, z[k]);
      }
    } else {
      for (int c = 0; c < nc; ++c) emit(c, logit(c));
    }
  }
  // block sums (fixed order inside the block; one double atomic per block)
  for (int o = 32; o > 0; o >>= 1) {
    my_loss += __shfl_xor(my_loss, o);
    my_cnt += __shfl_xor(my_cnt, o);
  }
  if ((threadIdx.x & 63) == 0) {
    s_sum[threadIdx.x >> 6] = my_loss;
    s_cnt[threadIdx.x >> 6] = my_cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // one pair per workgroup, summed in index order by seg_loss_finalize_kernel (deterministic, no same-address atomics)
    part[2 * (long long)blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
    part[2 * (long long)blockIdx.x + 1] = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
  }
}

__device__ __forceinline__ void bilinear_dst_range(int s, float scale, int out_size, int* lo, int* hi) {
  const float a = ((float)s - 0.5f) / scale - 0.5f, b = ((float)s + 1.5f) / scale - 0.5f;
  int l = (int)floorf(a) - 1, h = (int)ceilf(b) + 1;
  *lo = l < 0 ? 0 : l;
  *hi = h > out_size - 1 ? out_size - 1 : h;
}

// dpred[b][y*iw + x][c] = scale * sum over label pixels of w(oy, y) * w(ox, x) * g[b][c][oy][ox]; columns nc..ld-1 are zeroed.
// norm_mode 0: scale = grad_scale / norm_const; 1: scale = grad_scale / acc[1] (valid pixels counted by the pixel kernel).
// A workgroup owns 32 consecutive pixels of the flattened (b, y, x) order and all ld columns: thread = (pixel lane, class lane of 8), so a
// wave reads two class planes along x (contiguous), and the 32 x ld results leave through LDS as whole rows (one contiguous run of
// 32 * ld halves) -- a thread per (class, pixel) would write single halves 2 * ld bytes apart from workgroups on different XCDs.
constexpr int RG_PIX = 32;
__global__ __launch_bounds__(256) void resize_grad_rows_kernel(const float* g, int B, int nc, int ih, int iw, int OH, int OW, float grad_scale,
                                                               int norm_mode, double norm_const, const double* acc, half_t* dpred, int ld) {
  extern __shared__ __attribute__((aligned(16))) half_t tile[];  // [RG_PIX][ld]
  const long long npix = (long long)B * ih * iw;
  const long long p0 = (long long)blockIdx.x * RG_PIX;
  const int pl = threadIdx.x & (RG_PIX - 1), cl = threadIdx.x >> 5;
  const long long pidx = p0 + pl;
  const bool live = pidx < npix;
  const int x = live ? (int)(pidx % iw) : 0;
  const long long t = live ? pidx / iw : 0;
  const int y = (int)(t % ih);
  const int b = (int)(t / ih);
  const float sh = (float)ih / (float)OH, sw = (float)iw / (float)OW;
  int oy0, oy1, ox0, ox1;
  bilinear_dst_range(y, sh, OH, &oy0, &oy1);
  bilinear_dst_range(x, sw, OW, &ox0, &ox1);
  const double denom = norm_mode == 0 ? norm_const : (acc[1] > 0.0 ? acc[1] : 1.0);
  const float scale = (float)((double)grad_scale / denom);
  constexpr int XC = 12;  // label columns per chunk (the whole range at DeepLab's 4x upsampling)
  for (int c = cl; c < ld; c += 8) {
    float accv = 0.f;
    if (live && c < nc) {
      const float* gp = g + ((long long)b * nc + c) * OH * OW;
      for (int oxc = ox0; oxc <= ox1; oxc += XC) {  // the column weights of a chunk once, then every label row against them
        float wx[XC];
        int xo[XC];
#pragma unroll
        for (int j = 0; j < XC; ++j) {
          int x0, x1;
          float lx;
          bilinear_src(oxc + j, sw, iw, &x0, &x1, &lx);
          wx[j] = oxc + j <= ox1 ? (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f) : 0.f;
          xo[j] = oxc + j <= ox1 ? oxc + j : ox1;  // clamped: the loads below are unconditional (issued together), the weight is 0 there
        }
        for (int oy = oy0; oy <= oy1; ++oy) {
          int y0, y1;
          float ly;
          bilinear_src(oy, sh, ih, &y0, &y1, &ly);
          const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
          if (wy == 0.f) continue;
          const float* gr = gp + (long long)oy * OW;
          float gv[XC];
#pragma unroll
          for (int j = 0; j < XC; ++j) gv[j] = gr[xo[j]];
          float row = 0.f;
#pragma unroll
          for (int j = 0; j < XC; ++j) row = fmaf(wx[j], gv[j], row);
          accv = fmaf(wy, row, accv);
        }
      }
    }
    tile[pl * ld + c] = (half_t)(accv * scale);
  }
  __syncthreads();
  const long long nlive = npix - p0 < RG_PIX ? npix - p0 : RG_PIX;
  const int nhalf = (int)nlive * ld;
  half_t* dst = dpred + p0 * ld;
  if ((ld & 7) == 0 && (reinterpret_cast<uintptr_t>(dpred) & 15) == 0) {
    for (int i = threadIdx.x * 8; i < nhalf; i += 256 * 8) *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(tile + i);
  } else {
    for (int i = threadIdx.x; i < nhalf; i += 256) dst[i] = tile[i];
  }
}

__global__ __launch_bounds__(256) void seg_loss_finalize_kernel(const double* part, int nblocks, double* acc, int norm_mode, double norm_const,
                                                                float* loss) {
  __shared__ double s0[256], s1[256];
  double a = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) {
    a += part[2 * (long long)i];
    c += part[2 * (long long)i + 1];
  }
  s0[threadIdx.x] = a;
  s1[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      s0[threadIdx.x] += s0[threadIdx.x + o];
      s1[threadIdx.x] += s1[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    acc[0] = s0[0];
    acc[1] = s1[0];
    const double denom = norm_mode == 0 ? norm_const : s1[0];  // CE over zero valid pixels: 0 / 0 = nan, as torch
    loss[0] = (float)(s0[0] / denom);
  }
}

}  // namespace

static long long seg_pixel_blocks(long long npix) { return cvx_cdiv(npix, 256); }
extern "C" int64_t cvx_seg_loss_workspace_bytes(int32_t batch, int32_t nc, int32_t oh, int32_t ow) {
  // 64 B of totals, the per-workgroup (loss, count) pairs, the NCHW logit gradients
  return 64 + seg_pixel_blocks((long long)batch * oh * ow) * 16 + (int64_t)batch * nc * oh * ow * 4;
}

extern "C" int cvx_seg_loss(const float* rows_f32, int32_t ld, int32_t batch, int32_t nc, int32_t ih, int32_t iw, int32_t oh, int32_t ow,
                            const int64_t* target, int32_t mode, float alpha, float gamma, int64_t ignore_index, float loss_scale,
                            float* loss_out, void* dpred_f16, int32_t* bad_target, void* workspace, void* hip_stream) {
  CVX_CHECK(rows_f32 && target && loss_out && dpred_f16 && bad_target && workspace, "null arguments");
  CVX_CHECK(batch > 0 && nc > 0 && nc <= SEG_MAX_NC && nc <= ld && ih > 0 && iw > 0 && oh > 0 && ow > 0, "bad sizes");
  CVX_CHECK(mode == 0 || mode == 1, "mode: 0 focal, 1 cross-entropy");
  CVX_CHECK(loss_scale > 0.f, "loss_scale must be positive");
  hipStream_t st = (hipStream_t)hip_stream;
  const long long npix = (long long)batch * oh * ow;
  const long long nblk = seg_pixel_blocks(npix);
  double* acc = (double*)workspace;
  double* part = (double*)((char*)workspace + 64);
  float* dlogits = (float*)((char*)workspace + 64 + nblk * 16);
  CVX_CHECK(nblk < (1LL << 31) && ld <= 4096, "too many pixels / columns");
  CVX_HIP(hipMemsetAsync(bad_target, 0, 4, st));
  hipLaunchKernelGGL(seg_loss_pixel_kernel, dim3((unsigned)nblk), dim3(256), 0, st, rows_f32, ld, batch, nc, ih, iw, oh, ow,
                     (const long long*)target, mode, alpha, gamma, (long long)ignore_index, dlogits, part, bad_target);
  hipLaunchKernelGGL(seg_loss_finalize_kernel, dim3(1), dim3(256), 0, st, part, (int)nblk, acc, mode, (double)npix, loss_out);
  const long long nlow = (long long)batch * ih * iw;
  hipLaunchKernelGGL(resize_grad_rows_kernel, dim3((unsigned)cvx_cdiv(nlow, RG_PIX)), dim3(256), (size_t)RG_PIX * ld * 2, st, dlogits, batch, nc, ih, iw, oh,
                     ow, loss_scale, mode, (double)npix, acc, (half_t*)dpred_f16, ld);
  CVX_HIP(hipGetLastError());
  return 0;
}

extern "C" int cvx_resize_bilinear_nchw_grad_to_rows(const float* grad_nchw, int32_t batch, int32_t nc, int32_t ih, int32_t iw, int32_t oh,
                                                     int32_t ow, float scale, void* dpred_f16, int32_t ld, void* hip_stream) {
  CVX_CHECK(grad_nchw && dpred_f16 && batch > 0 && nc > 0 && nc <= ld && ih > 0 && iw > 0 && oh > 0 && ow > 0, "bad arguments");
  CVX_CHECK(ld <= 4096, "too many columns");
  const long long nlow = (long long)batch * ih * iw;
  hipLaunchKernelGGL(resize_grad_rows_kernel, dim3((unsigned)cvx_cdiv(nlow, RG_PIX)), dim3(256), (size_t)RG_PIX * ld * 2, (hipStream_t)hip_stream, grad_nchw,
                     batch, nc, ih, iw, oh, ow, scale, 0, 1.0, nullptr, (half_t*)dpred_f16, ld);
  CVX_HIP(hipGetLastError());
  return 0;
}
