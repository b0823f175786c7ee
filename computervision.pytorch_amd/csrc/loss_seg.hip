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
                                                             float* dlogits, double* acc /* [0] loss sum, [1] valid pixels */, int* bad) {
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
    float zmax = -INFINITY, se = 0.f, zt = 0.f;
    for (int c = 0; c < nc; ++c) {
      const float z = logit(c);
      if (c == (int)tg) zt = z;
      if (z > zmax) {
        se = se * expf(zmax - z) + 1.f;
        zmax = z;
      } else {
        se += expf(z - zmax);
      }
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
    for (int c = 0; c < nc; ++c) {
      const float p = expf(logit(c) - lse);
      dl[(long long)c * OH * OW] = valid ? coef * (p - (c == (int)tg ? 1.f : 0.f)) : 0.f;
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
  if (threadIdx.x == 0) {
    atomicAdd(&acc[0], (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]));
    atomicAdd(&acc[1], (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]));
  }
}

__device__ __forceinline__ void bilinear_dst_range(int s, float scale, int out_size, int* lo, int* hi) {
  const float a = ((float)s - 0.5f) / scale - 0.5f, b = ((float)s + 1.5f) / scale - 0.5f;
  int l = (int)floorf(a) - 1, h = (int)ceilf(b) + 1;
  *lo = l < 0 ? 0 : l;
  *hi = h > out_size - 1 ? out_size - 1 : h;
}

// dpred[b][y*iw + x][c] = scale * sum over label pixels of w(oy, y) * w(ox, x) * g[b][c][oy][ox]; columns nc..ld-1 are zeroed.
// norm_mode 0: scale = grad_scale / norm_const; 1: scale = grad_scale / acc[1] (valid pixels counted by K1)
__global__ __launch_bounds__(256) void resize_grad_rows_kernel(const float* g, int B, int nc, int ih, int iw, int OH, int OW, float grad_scale,
                                                               int norm_mode, double norm_const, const double* acc, half_t* dpred, int ld) {
  const long long n = (long long)B * ld * ih * iw;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % iw);
  long long t = i / iw;
  const int y = (int)(t % ih);
  t /= ih;
  const int c = (int)(t % ld);
  const int b = (int)(t / ld);
  half_t* dst = dpred + ((long long)b * ih * iw + (long long)y * iw + x) * ld + c;
  if (c >= nc) {
    *dst = (half_t)0.f;
    return;
  }
  const float sh = (float)ih / (float)OH, sw = (float)iw / (float)OW;
  int oy0, oy1, ox0, ox1;
  bilinear_dst_range(y, sh, OH, &oy0, &oy1);
  bilinear_dst_range(x, sw, OW, &ox0, &ox1);
  const float* gp = g + ((long long)b * nc + c) * OH * OW;
  float accv = 0.f;
  for (int oxc = ox0; oxc <= ox1; oxc += 16) {  // the column weights of up to 16 label columns once, then every label row against them
    float wx[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      int x0, x1;
      float lx;
      bilinear_src(oxc + j, sw, iw, &x0, &x1, &lx);
      wx[j] = oxc + j <= ox1 ? (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f) : 0.f;
    }
    for (int oy = oy0; oy <= oy1; ++oy) {
      int y0, y1;
      float ly;
      bilinear_src(oy, sh, ih, &y0, &y1, &ly);
      const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
      if (wy == 0.f) continue;
      const float* gr = gp + (long long)oy * OW + oxc;
      float row = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (wx[j] != 0.f) row = fmaf(wx[j], gr[j], row);
      accv = fmaf(wy, row, accv);
    }
  }
  const double denom = norm_mode == 0 ? norm_const : (acc[1] > 0.0 ? acc[1] : 1.0);
  *dst = (half_t)(accv * (float)((double)grad_scale / denom));
}

__global__ void seg_loss_finalize_kernel(const double* acc, int norm_mode, double norm_const, float* loss) {
  const double denom = norm_mode == 0 ? norm_const : acc[1];  // CE over zero valid pixels: 0 / 0 = nan, as torch
  loss[0] = (float)(acc[0] / denom);
}

}  // namespace

extern "C" int64_t cvx_seg_loss_workspace_bytes(int32_t batch, int32_t nc, int32_t oh, int32_t ow) {
  return (int64_t)batch * nc * oh * ow * 4 + 64;
}

extern "C" int cvx_seg_loss(const float* rows_f32, int32_t ld, int32_t batch, int32_t nc, int32_t ih, int32_t iw, int32_t oh, int32_t ow,
                            const int64_t* target, int32_t mode, float alpha, float gamma, int64_t ignore_index, float loss_scale,
                            float* loss_out, void* dpred_f16, int32_t* bad_target, void* workspace, void* hip_stream) {
  CVX_CHECK(rows_f32 && target && loss_out && dpred_f16 && bad_target && workspace, "null arguments");
  CVX_CHECK(batch > 0 && nc > 0 && nc <= SEG_MAX_NC && nc <= ld && ih > 0 && iw > 0 && oh > 0 && ow > 0, "bad sizes");
  CVX_CHECK(mode == 0 || mode == 1, "mode: 0 focal, 1 cross-entropy");
  CVX_CHECK(loss_scale > 0.f, "loss_scale must be positive");
  hipStream_t st = (hipStream_t)hip_stream;
  double* acc = (double*)workspace;
  float* dlogits = (float*)((char*)workspace + 64);
  CVX_HIP(hipMemsetAsync(acc, 0, 64, st));
  CVX_HIP(hipMemsetAsync(bad_target, 0, 4, st));
  const long long npix = (long long)batch * oh * ow;
  hipLaunchKernelGGL(seg_loss_pixel_kernel, dim3((unsigned)cvx_cdiv(npix, 256)), dim3(256), 0, st, rows_f32, ld, batch, nc, ih, iw, oh, ow,
                     (const long long*)target, mode, alpha, gamma, (long long)ignore_index, dlogits, acc, bad_target);
  const long long nrow = (long long)batch * ld * ih * iw;
  hipLaunchKernelGGL(resize_grad_rows_kernel, dim3((unsigned)cvx_cdiv(nrow, 256)), dim3(256), 0, st, dlogits, batch, nc, ih, iw, oh, ow, loss_scale,
                     mode, (double)npix, acc, (half_t*)dpred_f16, ld);
  hipLaunchKernelGGL(seg_loss_finalize_kernel, dim3(1), dim3(1), 0, st, acc, mode, (double)npix, loss_out);
  CVX_HIP(hipGetLastError());
  return 0;
}

extern "C" int cvx_resize_bilinear_nchw_grad_to_rows(const float* grad_nchw, int32_t batch, int32_t nc, int32_t ih, int32_t iw, int32_t oh,
                                                     int32_t ow, float scale, void* dpred_f16, int32_t ld, void* hip_stream) {
  CVX_CHECK(grad_nchw && dpred_f16 && batch > 0 && nc > 0 && nc <= ld && ih > 0 && iw > 0 && oh > 0 && ow > 0, "bad arguments");
  const long long nrow = (long long)batch * ld * ih * iw;
  hipLaunchKernelGGL(resize_grad_rows_kernel, dim3((unsigned)cvx_cdiv(nrow, 256)), dim3(256), 0, (hipStream_t)hip_stream, grad_nchw, batch, nc, ih,
                     iw, oh, ow, scale, 0, 1.0, nullptr, (half_t*)dpred_f16, ld);
  CVX_HIP(hipGetLastError());
  return 0;
}
