// CenterNet's CombinedLoss on gfx950: penalty-reduced focal loss on the heat map + two masked L1 terms gathered at the object centres,
// forward value AND the gradient w.r.t. the head output rows in one pass chain (no autograd tape).
//
// Reference semantics (file:line under the reference tree):
//   CombinedLoss.__call__   core/loss/centernet_loss.py:46-67   heatmap = clamp(sigmoid(pred[..., :nc]), 1e-4, 1 - 1e-4);
//                                                               "reg" = pred[..., nc:nc+2], "wh" = pred[..., -2:]  (the model emits
//                                                               [heatmap | wh head | reg head], centernet_model.py:368-378: the loss's
//                                                               names are swapped against the heads' -- reproduced as is);
//                                                               total = hm_w * focal + off_w * L1(reg) + wh_w * L1(wh)
//   FocalLoss               :5-26     pos = (t == 1), neg = (t < 1); -(sum log(p)(1-p)^2 pos + sum log(1-p) p^2 (1-t)^4 neg) / num_pos
//                                     (num_pos == 0: -neg sum)
//   RegL1Loss               :29-43    pred gathered at `indices` (row = y * w + x); sum |pred*mask - true*mask| / (sum(mask) * 2 + 1e-4)
//
//   K1 cn_reduce     per heat-map element: p, the two focal sums and num_pos (fixed-point block sums, order-independent)
//   K2 cn_l1         one workgroup per image: the gathered L1 sums and mask count; scatter of sign * mask into an fp32 plane
//                    (B, A, 4), sequentially over the objects of the image (objects may share a centre)
//   K3 cn_grad       per row: d loss / d logits from the sums of K1 / K2 -> dpred fp16 (loss-scaled), loss value by thread 0
#include <algorithm>
#include "cvx_common.h"
#include "../../include/cvx_engine.h"

namespace {

struct CnSums {  // device accumulator block (doubles; one atomic add per workgroup and value)
  double pos_loss, neg_loss, num_pos, l1_a, l1_b, mask_sum;
};

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(256) void cn_reduce_kernel(const float* rows, int ld, long long npix, int nc, const float* heat_true, CnSums* acc) {
  __shared__ double sm[3][4];
  const long long n = npix * nc;
  double pl = 0, nl = 0, np = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long pix = i / nc;
    const int c = (int)(i - pix * nc);
    const float t = heat_true[i];
    float p = 1.f / (1.f + expf(-rows[pix * ld + c]));
    p = fminf(fmaxf(p, 1e-4f), 1.f - 1e-4f);
    if (t == 1.f) {
      pl += (double)(logf(p) * (1.f - p) * (1.f - p));
      np += 1.0;
    } else if (t < 1.f) {
      const float w = (1.f - t) * (1.f - t);
      nl += (double)(logf(1.f - p) * p * p * w * w);
    }
  }
  pl = wave_sum(pl);
  nl = wave_sum(nl);
  np = wave_sum(np);
  if ((threadIdx.x & 63) == 0) {
    sm[0][threadIdx.x >> 6] = pl;
    sm[1][threadIdx.x >> 6] = nl;
    sm[2][threadIdx.x >> 6] = np;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&acc->pos_loss, (sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]));
    atomicAdd(&acc->neg_loss, (sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3]));
    atomicAdd(&acc->num_pos, (sm[2][0] + sm[2][1]) + (sm[2][2] + sm[2][3]));
  }
}

// cols_a: the output's columns nc, nc+1 ("reg" of the loss), cols_b: its last two ("wh" of the loss); plane: (B, A, 4) fp32, zero on entry
__global__ __launch_bounds__(64) void cn_l1_kernel(const float* rows, int ld, int A, int col_a, int col_b, const float* true_a, const float* true_b,
                                                   const float* mask, const long long* indices, int K, float* plane, CnSums* acc, int* bad) {
  const int b = blockIdx.x;
  double la = 0, lb = 0, ms = 0;
  if (threadIdx.x < 4) {  // one thread per output coordinate walks the image's objects in order: deterministic with shared centres
    const int j = threadIdx.x;
    for (int k = 0; k < K; ++k) {
      const float m = mask[(long long)b * K + k];
      const long long idx = indices[(long long)b * K + k];
      if (idx < 0 || idx >= A) {
        if (m != 0.f) atomicOr(bad, 1);
        continue;
      }
      const float pred = rows[((long long)b * A + idx) * ld + (j < 2 ? col_a + j : col_b + j - 2)];
      const float tv = (j < 2 ? true_a : true_b)[((long long)b * K + k) * 2 + (j & 1)];
      const float d = pred * m - tv * m;
      (j < 2 ? la : lb) += (double)fabsf(d);
      ms += (double)m;                                   // mask.unsqueeze(2).expand_as(pred): counted once per coordinate
      const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      plane[((long long)b * A + idx) * 4 + j] += sg * m;
    }
  }
  la = wave_sum(la);
  lb = wave_sum(lb);
  ms = wave_sum(ms);
  if (threadIdx.x == 0) {
    atomicAdd(&acc->l1_a, la);
    atomicAdd(&acc->l1_b, lb);
    atomicAdd(&acc->mask_sum, ms * 0.5);                 // each L1 term's own denominator: sum over its two coordinates = ms / 2
  }
}

__global__ __launch_bounds__(256) void cn_grad_kernel(const float* rows, int ld, long long npix, int nc, int col_a, int col_b, const float* heat_true,
                                                      const float* plane, const CnSums* acc, float hm_w, float a_w, float b_w, float loss_scale,
                                                      half_t* dpred, float* loss_out) {
  const double num_pos = acc->num_pos;
  const float hm_scale = (float)((double)hm_w * (double)loss_scale / (num_pos > 0.0 ? num_pos : 1.0));
  const float l1_den = (float)(acc->mask_sum + 1e-4);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const double hm = num_pos > 0.0 ? -(acc->pos_loss + acc->neg_loss) / num_pos : -acc->neg_loss;
    loss_out[0] = (float)((double)hm_w * hm + (double)a_w * acc->l1_a / (double)l1_den + (double)b_w * acc->l1_b / (double)l1_den);
    loss_out[1] = (float)hm;
    loss_out[2] = (float)(acc->l1_a / (double)l1_den);
    loss_out[3] = (float)(acc->l1_b / (double)l1_den);
  }
  const long long n = npix * ld;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long pix = i / ld;
    const int c = (int)(i - pix * ld);
    float g = 0.f;
    if (c < nc) {
      const float t = heat_true[pix * nc + c];
      const float ps = 1.f / (1.f + expf(-rows[i]));
      if (ps >= 1e-4f && ps <= 1.f - 1e-4f) {          // torch.clamp passes the gradient inside the range (boundaries included)
        float dp = 0.f;                                 // d(pos + neg sums) / dp
        if (t == 1.f)
          dp = (1.f - ps) * (1.f - ps) / ps - 2.f * (1.f - ps) * logf(ps);
        else if (t < 1.f) {
          const float w = (1.f - t) * (1.f - t);
          dp = (-ps * ps / (1.f - ps) + 2.f * ps * logf(1.f - ps)) * w * w;
        }
        g = -dp * ps * (1.f - ps) * hm_scale;
      }
    } else if (c >= col_a && c < col_a + 2) {
      g = plane[pix * 4 + (c - col_a)] * a_w * loss_scale / l1_den;
    } else if (c >= col_b && c < col_b + 2) {
      g = plane[pix * 4 + 2 + (c - col_b)] * b_w * loss_scale / l1_den;
    }
    dpred[i] = (half_t)g;
  }
}

}  // namespace

extern "C" int64_t cvx_centernet_loss_workspace_bytes(int32_t batch, int32_t anchors) { return 256 + (int64_t)batch * anchors * 4 * 4; }

extern "C" int cvx_centernet_loss(const float* rows_f32, int32_t ld, int32_t batch, int32_t anchors, int32_t nc, int32_t col_a, int32_t col_b,
                                  const float* heat_true, const float* true_a, const float* true_b, const float* mask, const int64_t* indices,
                                  int32_t max_objects, float hm_weight, float a_weight, float b_weight, float loss_scale, float* loss_items,
                                  void* dpred_f16, int32_t* bad_index, void* workspace, void* hip_stream) {
  CVX_CHECK(rows_f32 && heat_true && true_a && true_b && mask && indices && loss_items && dpred_f16 && bad_index && workspace, "null arguments");
  CVX_CHECK(batch > 0 && anchors > 0 && nc > 0 && nc <= ld && max_objects >= 0 && col_a >= nc && col_a + 2 <= ld && col_b >= nc && col_b + 2 <= ld,
            "bad sizes / columns");
  CVX_CHECK(loss_scale > 0.f, "loss_scale must be positive");
  hipStream_t st = (hipStream_t)hip_stream;
  CnSums* acc = (CnSums*)workspace;
  float* plane = (float*)((char*)workspace + 256);
  const long long npix = (long long)batch * anchors;
  CVX_HIP(hipMemsetAsync(workspace, 0, 256 + (size_t)npix * 16, st));
  CVX_HIP(hipMemsetAsync(bad_index, 0, 4, st));
  // one workgroup per CU: every workgroup ends in three double atomics on the same three addresses, and those serialise (2048 of them: 80 us)
  const int blocks = (int)std::min<long long>(256, cvx_cdiv(npix * nc, 256));
  hipLaunchKernelGGL(cn_reduce_kernel, dim3(blocks), dim3(256), 0, st, rows_f32, ld, npix, nc, heat_true, acc);
  hipLaunchKernelGGL(cn_l1_kernel, dim3(batch), dim3(64), 0, st, rows_f32, ld, anchors, col_a, col_b, true_a, true_b, mask, (const long long*)indices,
                     max_objects, plane, acc, bad_index);
  const int gblocks = (int)std::min<long long>(4096, cvx_cdiv(npix * ld, 256));
  hipLaunchKernelGGL(cn_grad_kernel, dim3(gblocks), dim3(256), 0, st, rows_f32, ld, npix, nc, col_a, col_b, heat_true, plane, acc, hm_weight, a_weight,
                     b_weight, loss_scale, (half_t*)dpred_f16, loss_items);
  CVX_HIP(hipGetLastError());
  return 0;
}

// ---- CenterNet target drawing (CenterNet.generate_targets, core/algorithms/centernet.py:66-112; gaussian_radius / gaussian2D /
// draw_umich_gaussian, core/utils/gaussian.py:5-57): the CPU work of centernet_collate -------------------------------------------
// Per object: corners from (cx, cy, w, h) in float32, scaled to the feature map; integer height / width (truncation); the CornerNet radius
// (three quadratics, float64); the integer centre; a (2r+1)^2 Gaussian with sigma = (2r+1)/6 in float64, values below eps * max dropped,
// merged into the class plane with a maximum -- which commutes, so all objects of the batch are drawn concurrently with an atomic
// maximum on the bit patterns of the non-negative floats.  reg = fractional part of the centre, wh = the integer size, ind = y * W + x.
namespace {

__global__ __launch_bounds__(256) void cn_draw_kernel(const float* labels, const int* counts, int kmax, int H, int W, int nc, float* heat, float* reg,
                                                      float* wh, float* mask, float* ind) {
#pragma clang fp contract(off)  // numpy rounds every product and sum: no fused multiply-adds in the coordinate arithmetic
  __shared__ int s_par[5];  // x, y, radius, class, valid
  const int b = blockIdx.x / kmax, j = blockIdx.x - b * kmax;
  const long long o = (long long)b * kmax + j;
  if (threadIdx.x == 0) {
    s_par[4] = 0;
    if (j < counts[b]) {
      const float* l = labels + o * 5;  // class id, cx, cy, w, h (normalised)
      const float xmin = (l[1] - l[3] / 2) * (float)W, ymin = (l[2] - l[4] / 2) * (float)H;
      const float xmax = (l[1] + l[3] / 2) * (float)W, ymax = (l[2] + l[4] / 2) * (float)H;
      const int h = (int)(ymax - ymin), w = (int)(xmax - xmin);
      const double ov = 0.7, hh = (double)h, ww = (double)w;
      const double b1 = hh + ww, c1 = ww * hh * (1 - ov) / (1 + ov);
      const double r1 = (b1 + sqrt(b1 * b1 - 4 * 1 * c1)) / 2;
      const double b2 = 2 * (hh + ww), c2 = (1 - ov) * ww * hh;
      const double r2 = (b2 + sqrt(b2 * b2 - 4 * 4 * c2)) / 2;
      const double a3 = 4 * ov, b3 = -2 * ov * (hh + ww), c3 = (ov - 1) * ww * hh;
      const double r3 = (b3 + sqrt(b3 * b3 - 4 * a3 * c3)) / 2;
      const double rr = fmin(r1, fmin(r2, r3));
      int radius = (int)rr;
      if (radius < 0) radius = 0;
      const float cx = (xmin + xmax) / 2, cy = (ymin + ymax) / 2;
      const int ix = (int)cx, iy = (int)cy;
      s_par[0] = ix;
      s_par[1] = iy;
      s_par[2] = radius;
      s_par[3] = (int)l[0];
      s_par[4] = 1;
      reg[o * 2] = cx - (float)ix;
      reg[o * 2 + 1] = cy - (float)iy;
      wh[o * 2] = (float)w;
      wh[o * 2 + 1] = (float)h;
      mask[o] = 1.f;
      ind[o] = (float)(iy * W + ix);
    } else {
      reg[o * 2] = reg[o * 2 + 1] = wh[o * 2] = wh[o * 2 + 1] = mask[o] = ind[o] = 0.f;
    }
  }
  __syncthreads();
  if (!s_par[4]) return;
  const int x = s_par[0], y = s_par[1], r = s_par[2], cls = s_par[3];
  if (x < 0 || y < 0 || cls < 0 || cls >= nc) return;
  const int left = min(x, r), right = min(W - x, r + 1), top = min(y, r), bottom = min(H - y, r + 1);
  const int nw = left + right, nh = top + bottom;
  if (nw <= 0 || nh <= 0) return;
  const double sigma = (double)(2 * r + 1) / 6.0;
  for (int t = threadIdx.x; t < nw * nh; t += 256) {
    const int dy = t / nw - top, dx = t - (t / nw) * nw - left;
    double g = exp(-(double)(dx * dx + dy * dy) / (2 * sigma * sigma));
    if (g < 2.220446049250313e-16) g = 0.0;  // h[h < eps * h.max()] = 0, h.max() = 1 at the centre
    const float gf = (float)g;
    atomicMax(reinterpret_cast<unsigned*>(heat + (((long long)b * H + (y + dy)) * W + (x + dx)) * nc + cls), __float_as_uint(gf));
  }
}

}  // namespace

extern "C" int cvx_centernet_draw_targets(const float* labels, const int32_t* counts, int32_t batch, int32_t max_boxes, int32_t fh, int32_t fw, int32_t nc,
                                          float* heatmap, float* reg, float* wh, float* reg_mask, float* indices, void* hip_stream) {
  CVX_CHECK(labels && counts && heatmap && reg && wh && reg_mask && indices && batch > 0 && max_boxes > 0 && fh > 0 && fw > 0 && nc > 0, "bad arguments");
  hipStream_t st = (hipStream_t)hip_stream;
  CVX_HIP(hipMemsetAsync(heatmap, 0, (size_t)batch * fh * fw * nc * 4, st));
  hipLaunchKernelGGL(cn_draw_kernel, dim3(batch * max_boxes), dim3(256), 0, st, labels, counts, max_boxes, fh, fw, nc, heatmap, reg, wh, reg_mask, indices);
  CVX_HIP(hipGetLastError());
  return 0;
}
