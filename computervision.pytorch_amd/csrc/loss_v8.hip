// YOLOv8 detection loss on gfx950: task-aligned assigner + BCE + CIoU + DFL, forward value AND the
// gradient w.r.t. the head output in one pass chain (no autograd tape).
//
// Reference semantics (file:line under the reference tree):
//   Loss.__call__                core/algorithms/yolo_v8.py:75-124
//   TaskAlignedAssigner          core/utils/bboxes.py:275-470  (+ :231-272)
//   bbox_iou (CIoU)              core/utils/ultralytics_iou.py:64-117
//   BboxLoss / _df_loss          core/loss/ultralytics_loss.py:25-57
//
// Work is organised per TARGET (N rows) instead of the reference's zero-padded (B, Gmax) grid: padded
// rows never win anything (their mask is 0 and their overlaps are 0), so the results are identical.
// Targets must be grouped by image (stable order), which yolo8_collate (core/data/collate.py:17-29)
// produces; the Python wrapper sorts otherwise.
//
//   K0 tgt_prep     per target: cxcywh(norm) -> xyxy px, validity, per-image [start,end)
//   K1 pred_prep    per anchor: softmax(16) expectation -> ltrb -> xyxy (grid units)
//   K2 tal_metric   per target (one block): in-gt test, CIoU, metric = sqrt(score)*iou^6, top-10
//   K3 tal_resolve  per anchor: multi-claim resolution by max overlap -> gt index
//   K4 tal_posmax   per target: max metric / max overlap over its final positives
//   K5 tal_norm     per anchor: normalised target score; partial sums of target_scores_sum
//   K6 loss_grad    16 lanes per anchor: BCE, CIoU, DFL values + d/dpred (fp16, loss-scaled)
//   K7 finalize     deterministic reduction of the partial sums -> loss_items[3]
#include <cstring>
#include "cvx_common.h"
#include "../../include/cvx_engine.h"

namespace {

constexpr int REG = 16;
constexpr int TOPK = 10;
constexpr int MAXLV = 4;

struct Levels {
  int n;
  int a_off[MAXLV + 1];
  int w[MAXLV];
  float stride[MAXLV];
};

__device__ __forceinline__ void anchor_of(const Levels& L, int a, float* ax, float* ay, float* stride) {
  int lv = 0;
#pragma unroll
  for (int i = 1; i < MAXLV; ++i)
    if (i < L.n && a >= L.a_off[i]) lv = i;
  int r = a - L.a_off[lv];
  int y = r / L.w[lv], x = r - y * L.w[lv];
  *ax = x + 0.5f;
  *ay = y + 0.5f;
  *stride = L.stride[lv];
}

// CIoU value (ultralytics_iou.py:88-113), box1 = a, box2 = b, xyxy
__device__ __forceinline__ float ciou_val(float ax1, float ay1, float ax2, float ay2, float bx1, float by1, float bx2, float by2) {
  const float eps = 1e-7f;
  float w1 = ax2 - ax1, h1 = ay2 - ay1 + eps, w2 = bx2 - bx1, h2 = by2 - by1 + eps;
  float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f), ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
  float inter = iw * ih;
  float uni = w1 * h1 + w2 * h2 - inter + eps;
  float iou = inter / uni;
  float cw = fmaxf(ax2, bx2) - fminf(ax1, bx1), ch = fmaxf(ay2, by2) - fminf(ay1, by1);
  float c2 = cw * cw + ch * ch + eps;
  float sx = bx1 + bx2 - ax1 - ax2, sy = by1 + by2 - ay1 - ay2;
  float rho2 = (sx * sx + sy * sy) * 0.25f;
  float dat = atanf(w2 / h2) - atanf(w1 / h1);
  float v = 0.40528473456935108578f * dat * dat;  // 4/pi^2
  float alpha = v / (v - iou + (1.f + eps));
  return iou - (rho2 / c2 + v * alpha);
}

// CIoU value and gradient w.r.t. box a (alpha treated as a constant, ultralytics_iou.py:111-112)
__device__ __forceinline__ float ciou_grad(float ax1, float ay1, float ax2, float ay2, float bx1, float by1, float bx2, float by2, float g[4]) {
  const float eps = 1e-7f;
  float w1 = ax2 - ax1, h1 = ay2 - ay1 + eps, w2 = bx2 - bx1, h2 = by2 - by1 + eps;
  float iw_raw = fminf(ax2, bx2) - fmaxf(ax1, bx1), ih_raw = fminf(ay2, by2) - fmaxf(ay1, by1);
  float iw = fmaxf(iw_raw, 0.f), ih = fmaxf(ih_raw, 0.f);
  float inter = iw * ih;
  float uni = w1 * h1 + w2 * h2 - inter + eps;
  float iou = inter / uni;
  float cw = fmaxf(ax2, bx2) - fminf(ax1, bx1), ch = fmaxf(ay2, by2) - fminf(ay1, by1);
  float c2 = cw * cw + ch * ch + eps;
  float sx = bx1 + bx2 - ax1 - ax2, sy = by1 + by2 - ay1 - ay2;
  float rho2 = (sx * sx + sy * sy) * 0.25f;
  float dat = atanf(w2 / h2) - atanf(w1 / h1);
  const float k4pi2 = 0.40528473456935108578f;
  float v = k4pi2 * dat * dat;
  float alpha = v / (v - iou + (1.f + eps));
  // order: x1, y1, x2, y2
  float diw[4] = {(iw_raw > 0.f && ax1 > bx1) ? -1.f : 0.f, 0.f, (iw_raw > 0.f && ax2 < bx2) ? 1.f : 0.f, 0.f};
  float dih[4] = {0.f, (ih_raw > 0.f && ay1 > by1) ? -1.f : 0.f, 0.f, (ih_raw > 0.f && ay2 < by2) ? 1.f : 0.f};
  float dw1[4] = {-1.f, 0.f, 1.f, 0.f}, dh1[4] = {0.f, -1.f, 0.f, 1.f};
  float dcw[4] = {ax1 < bx1 ? -1.f : 0.f, 0.f, ax2 > bx2 ? 1.f : 0.f, 0.f};
  float dch[4] = {0.f, ay1 < by1 ? -1.f : 0.f, 0.f, ay2 > by2 ? 1.f : 0.f};
  float drho[4] = {-0.5f * sx, -0.5f * sy, -0.5f * sx, -0.5f * sy};
  float inv_u2 = 1.f / (uni * uni), inv_c22 = 1.f / (c2 * c2), inv_hw = 1.f / (h1 * h1 + w1 * w1);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float dinter = diw[i] * ih + iw * dih[i];
    float duni = dw1[i] * h1 + w1 * dh1[i] - dinter;
    float diou = (dinter * uni - inter * duni) * inv_u2;
    float dc2 = 2.f * cw * dcw[i] + 2.f * ch * dch[i];
    float dpen = (drho[i] * c2 - rho2 * dc2) * inv_c22;
    float datan1 = (h1 * dw1[i] - w1 * dh1[i]) * inv_hw;
    float dv = 2.f * k4pi2 * dat * (-datan1);
    g[i] = diou - dpen - alpha * dv;
  }
  return iou - (rho2 / c2 + v * alpha);
}

// ---- K0 -------------------------------------------------------------------------------------------
__global__ void tgt_prep_kernel(const float* tg, int N, int B, float img_w, float img_h, float* gtbox, int* gtlabel, int* gtb, int* gtvalid,
                                int* img_start, int* img_end) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* t = tg + (long long)n * 6;
  int b = (int)t[0];
  float cx = t[2] * img_w, cy = t[3] * img_h, w = t[4] * img_w, h = t[5] * img_h;
  float x1 = cx - w / 2, y1 = cy - h / 2, x2 = cx + w / 2, y2 = cy + h / 2;
  gtbox[n * 4 + 0] = x1;
  gtbox[n * 4 + 1] = y1;
  gtbox[n * 4 + 2] = x2;
  gtbox[n * 4 + 3] = y2;
  gtlabel[n] = (int)t[1];
  gtb[n] = b;
  gtvalid[n] = (x1 + y1 + x2 + y2) > 0.f;
  if (b >= 0 && b < B) {
    if (n == 0 || (int)tg[(long long)(n - 1) * 6] != b) img_start[b] = n;
    if (n == N - 1 || (int)tg[(long long)(n + 1) * 6] != b) img_end[b] = n + 1;
  }
}

// ---- K1 -------------------------------------------------------------------------------------------
__global__ void pred_prep_kernel(const float* pred, int B, int A, int no, Levels L, float* pbox) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (b*A + a)*4 + side
  if (i >= (long long)B * A * 4) return;
  int side = (int)(i & 3);
  long long ba = i >> 2;
  int a = (int)(ba % A);
  const float* l = pred + ba * no + side * REG;
  float v[REG];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < REG; k += 4) {
    f4 q = *reinterpret_cast<const f4*>(l + k);
    v[k] = q[0];
    v[k + 1] = q[1];
    v[k + 2] = q[2];
    v[k + 3] = q[3];
  }
#pragma unroll
  for (int k = 0; k < REG; ++k) mx = fmaxf(mx, v[k]);
  float s = 0.f, e = 0.f;
#pragma unroll
  for (int k = 0; k < REG; ++k) {
    float p = __expf(v[k] - mx);
    s += p;
    e += p * k;
  }
  float d = e / s;
  float ax, ay, st;
  anchor_of(L, a, &ax, &ay, &st);
  float c = (side & 1) ? ay : ax;
  pbox[i] = side < 2 ? c - d : c + d;  // x1 = ax - l, y1 = ay - t, x2 = ax + r, y2 = ay + b
}

// ---- K2: one block per target -----------------------------------------------------------------------
// (one target per block, TAL_TPB threads: the block's life is a chain of short scans over the 8400 anchors -- 96 blocks of 256 threads took 53 us)
constexpr int TAL_TPB = 1024;
__global__ __launch_bounds__(TAL_TPB) void tal_metric_kernel(const float* pred, int A, int no, int nc, Levels L, const float* pbox, const float* gtbox,
                                                         const int* gtlabel, const int* gtb, const int* gtvalid, float* ov, float* metric,
                                                         uint8_t* mask) {
  extern __shared__ float smet[];  // A floats
  __shared__ float rv[TAL_TPB / 64];
  __shared__ int ri[TAL_TPB / 64];
  const int n = blockIdx.x;
  const int b = gtb[n];
  const bool valid = gtvalid[n] != 0;
  const float gx1 = gtbox[n * 4], gy1 = gtbox[n * 4 + 1], gx2 = gtbox[n * 4 + 2], gy2 = gtbox[n * 4 + 3];
  int label = gtlabel[n];
  label = min(max(label, 0), nc - 1);
  for (int a = threadIdx.x; a < A; a += TAL_TPB) {
    float ax, ay, st;
    anchor_of(L, a, &ax, &ay, &st);
    float px = ax * st, py = ay * st;
    float dmin = fminf(fminf(px - gx1, py - gy1), fminf(gx2 - px, gy2 - py));
    float o = 0.f, m = 0.f;
    if (valid && dmin > 1e-9f) {
      const float* pb = pbox + ((long long)b * A + a) * 4;
      o = fmaxf(ciou_val(gx1, gy1, gx2, gy2, pb[0] * st, pb[1] * st, pb[2] * st, pb[3] * st), 0.f);
      float sc = cvx_sigmoid(pred[((long long)b * A + a) * no + 4 * REG + label]);
      float o2 = o * o;
      m = sqrtf(sc) * (o2 * o2 * o2);
    }
    ov[(long long)n * A + a] = o;
    metric[(long long)n * A + a] = m;
    mask[(long long)n * A + a] = 0;
    smet[a] = m;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = 0; r < TOPK; ++r) {
    float best = 0.f;
    int bi = 0x7fffffff;
    for (int a = threadIdx.x; a < A; a += TAL_TPB) {
      float m = smet[a];
      if (m > best) {  // strided ascending scan: first (lowest) index wins ties inside a thread
        best = m;
        bi = a;
      }
    }
    for (int o = 1; o < 64; o <<= 1) {
      float ob = __shfl_xor(best, o);
      int oi = __shfl_xor(bi, o);
      if (ob > best || (ob == best && oi < bi)) {
        best = ob;
        bi = oi;
      }
    }
    if (lane == 0) {
      rv[wave] = best;
      ri[wave] = bi;
    }
    __syncthreads();
    float fb = rv[0];
    int fi = ri[0];
#pragma unroll
    for (int w = 1; w < TAL_TPB / 64; ++w)
      if (rv[w] > fb || (rv[w] == fb && ri[w] < fi)) {
        fb = rv[w];
        fi = ri[w];
      }
    __syncthreads();
    if (!(fb > 0.f)) break;  // block-uniform
    if (threadIdx.x == 0) {
      smet[fi] = -1.f;
      mask[(long long)n * A + fi] = 1;
    }
    __syncthreads();
  }
}

// ---- K3 -------------------------------------------------------------------------------------------
__global__ void tal_resolve_kernel(int B, int A, const int* img_start, const int* img_end, const float* ov, const uint8_t* mask, int* gt_idx) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * A) return;
  int b = (int)(i / A), a = (int)(i - (long long)b * A);
  int s = img_start[b], e = img_end[b];
  int cnt = 0, first = -1, arg = s;
  float best = -1.f;
  for (int n = s; n < e; ++n) {
    if (mask[(long long)n * A + a]) {
      if (cnt == 0) first = n;
      ++cnt;
    }
    float o = ov[(long long)n * A + a];
    if (o > best) {
      best = o;
      arg = n;
    }
  }
  gt_idx[i] = cnt == 0 ? -1 : (cnt == 1 ? first : arg);
}

// ---- K4: one block per target -----------------------------------------------------------------------
__global__ __launch_bounds__(TAL_TPB) void tal_posmax_kernel(int A, const int* gtb, const int* gt_idx, const float* ov, const float* metric,
                                                         float* pos_align, float* pos_ov) {
  __shared__ float sa[TAL_TPB / 64], so[TAL_TPB / 64];
  const int n = blockIdx.x;
  const int b = gtb[n];
  float ma = 0.f, mo = 0.f;
  for (int a = threadIdx.x; a < A; a += TAL_TPB) {
    if (gt_idx[(long long)b * A + a] == n) {
      ma = fmaxf(ma, metric[(long long)n * A + a]);
      mo = fmaxf(mo, ov[(long long)n * A + a]);
    }
  }
  ma = cvx_wave_max64(ma);
  mo = cvx_wave_max64(mo);
  if ((threadIdx.x & 63) == 0) {
    sa[threadIdx.x >> 6] = ma;
    so[threadIdx.x >> 6] = mo;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float fa = sa[0], fo = so[0];
#pragma unroll
    for (int w = 1; w < TAL_TPB / 64; ++w) {
      fa = fmaxf(fa, sa[w]);
      fo = fmaxf(fo, so[w]);
    }
    pos_align[n] = fa;
    pos_ov[n] = fo;
  }
}

// ---- K5 -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tal_norm_kernel(long long BA, int A, const int* gt_idx, const float* metric, const float* pos_align,
                                                       const float* pos_ov, float* norm, float* part) {
  __shared__ float sw[4];
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  float v = 0.f;
  if (i < BA) {
    int g = gt_idx[i];
    if (g >= 0) {
      int a = (int)(i % A);
      v = metric[(long long)g * A + a] * pos_ov[g] / (pos_align[g] + 1e-9f);
    }
    norm[i] = v;
  }
  v = cvx_wave_sum64(v);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

__device__ __forceinline__ float block_sum_partials(const float* part, int np) {
  // every block computes the same deterministic total
  __shared__ float sw[4];
  float v = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) v += part[i];
  v = cvx_wave_sum64(v);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = (sw[0] + sw[1]) + (sw[2] + sw[3]);
  __syncthreads();
  return t;
}

// ---- K6: 16 lanes per anchor, 16 anchors per 256-thread block ---------------------------------------
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* pred, int B, int A, int no, int nc, Levels L, const float* pbox,
                                                        const float* gtbox, const int* gtlabel, const int* gt_idx, const float* norm,
                                                        const float* tss_part, int n_tss_part, float gain_box, float gain_cls, float gain_dfl,
                                                        float loss_scale, half_t* dpred, float* loss_part) {
  __shared__ float sl[4][3];
  const float tss = fmaxf(block_sum_partials(tss_part, n_tss_part), 1.f);
  const int lane = threadIdx.x & 63, s = lane & 15;
  const long long BA = (long long)B * A;
  const long long i = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = i < BA;
  const long long ii = live ? i : 0;
  const int a = (int)(ii % A);
  const float* p = pred + ii * no;
  half_t* dp = dpred + ii * no;
  const int g = live ? gt_idx[ii] : -1;
  const bool fg = g >= 0;
  const float w = fg ? norm[ii] : 0.f;
  const float kscale = (float)B / tss * loss_scale;
  float l_cls = 0.f, l_box = 0.f, l_dfl = 0.f;

  // ---- class BCE with logits (yolo_v8.py:113) ----
  int label = fg ? min(max(gtlabel[g], 0), nc - 1) : -1;
  const float kc = gain_cls * kscale;
  if ((nc & 3) == 0) {
    // four classes per lane (16-byte loads, 8-byte stores) and ONE exponential per logit: with e = exp(-|x|),
    // softplus(-|x|) = log1p(e) and sigmoid(x) = (x >= 0 ? 1 : e) / (1 + e)
    for (int c4 = s * 4; c4 < nc; c4 += 64) {
      if (live) {
        const f4 x4 = *reinterpret_cast<const f4*>(p + 4 * REG + c4);
        h4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float x = x4[k];
          const float t = (c4 + k == label) ? w : 0.f;
          const float e = __expf(-fabsf(x));
          // log1p(e): short series where 1 + e would lose e's low bits, plain log otherwise (|error| < 1e-7 relative)
          const float l1p = e < 1e-3f ? e * (1.f - e * (0.5f - e * (1.f / 3.f))) : __logf(1.f + e);
          l_cls += fmaxf(x, 0.f) - x * t + l1p;
          const float sg = (x >= 0.f ? 1.f : e) * __builtin_amdgcn_rcpf(1.f + e);
          o[k] = (half_t)((sg - t) * kc);
        }
        *reinterpret_cast<h4*>(dp + 4 * REG + c4) = o;
      }
    }
  } else {
    for (int c = s; c < nc; c += 16) {
      if (live) {
        float x = p[4 * REG + c];
        float t = (c == label) ? w : 0.f;
        l_cls += fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));
        dp[4 * REG + c] = (half_t)((cvx_sigmoid(x) - t) * kc);
      }
    }
  }

  // ---- box (CIoU) + DFL on the 64 distribution logits: lane s -> side s>>2, bins 4*(s&3).. ----
  const int side = s >> 2, q = s & 3;
  f4 lg = live ? *reinterpret_cast<const f4*>(p + side * REG + q * 4) : f4{0.f, 0.f, 0.f, 0.f};
  float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
  mx = fmaxf(mx, __shfl_xor(mx, 1));
  mx = fmaxf(mx, __shfl_xor(mx, 2));
  float ex[4], se = 0.f, sb = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ex[k] = __expf(lg[k] - mx);
    se += ex[k];
    sb += ex[k] * (float)(q * 4 + k);
  }
  se += __shfl_xor(se, 1);
  se += __shfl_xor(se, 2);
  sb += __shfl_xor(sb, 1);
  sb += __shfl_xor(sb, 2);
  const float inv_se = 1.f / se;
  const float E = sb * inv_se;  // expected distance of this side
  const int base = lane & ~15;
  const float dl = __shfl(E, base + 0), dt = __shfl(E, base + 4), dr = __shfl(E, base + 8), db = __shfl(E, base + 12);
  float gout[4] = {0.f, 0.f, 0.f, 0.f};
  if (fg) {
    float ax, ay, st;
    anchor_of(L, a, &ax, &ay, &st);
    const float inv_st = 1.f / st;
    const float tx1 = gtbox[g * 4] * inv_st, ty1 = gtbox[g * 4 + 1] * inv_st, tx2 = gtbox[g * 4 + 2] * inv_st, ty2 = gtbox[g * 4 + 3] * inv_st;
    float gb[4];
    float ci = ciou_grad(ax - dl, ay - dt, ax + dr, ay + db, tx1, ty1, tx2, ty2, gb);
    if (s == 0) l_box = (1.f - ci) * w;
    // d(1-ciou)/d(dist of my side): x1 = ax - l, y1 = ay - t, x2 = ax + r, y2 = ay + b
    const float dE = side == 0 ? gb[0] : side == 1 ? gb[1] : side == 2 ? -gb[2] : -gb[3];
    const float kb = gain_box * kscale * w;
    // DFL target of my side (bbox2dist, bboxes.py:225-228; clamp to [0, reg_max-1-0.01])
    float tdist = side == 0 ? ax - tx1 : side == 1 ? ay - ty1 : side == 2 ? tx2 - ax : ty2 - ay;
    tdist = fminf(fmaxf(tdist, 0.f), (float)(REG - 1) - 0.01f);
    const int tl = (int)tdist;
    const float wl = (float)(tl + 1) - tdist, wr = 1.f - wl;
    const float kd = gain_dfl * kscale * w * 0.25f;
    const float lse = mx + __logf(se);
    float ce = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int bin = q * 4 + k;
      const float pk = ex[k] * inv_se;
      const float ind = (bin == tl ? wl : 0.f) + (bin == tl + 1 ? wr : 0.f);
      ce += ind * (lse - lg[k]);
      gout[k] = kb * dE * pk * ((float)bin - E) + kd * (pk - ind);
    }
    ce += __shfl_xor(ce, 1);
    ce += __shfl_xor(ce, 2);  // CE_left*wl + CE_right*wr of this side
    if (q == 0) l_dfl = ce * 0.25f * w;
  }
  if (live) {
    h4 o = {(half_t)gout[0], (half_t)gout[1], (half_t)gout[2], (half_t)gout[3]};
    *reinterpret_cast<h4*>(dp + side * REG + q * 4) = o;
  }

  l_box = cvx_wave_sum64(l_box);
  l_cls = cvx_wave_sum64(l_cls);
  l_dfl = cvx_wave_sum64(l_dfl);
  if (lane == 0) {
    sl[threadIdx.x >> 6][0] = l_box;
    sl[threadIdx.x >> 6][1] = l_cls;
    sl[threadIdx.x >> 6][2] = l_dfl;
  }
  __syncthreads();
  if (threadIdx.x < 3) loss_part[(long long)blockIdx.x * 3 + threadIdx.x] = (sl[0][threadIdx.x] + sl[1][threadIdx.x]) + (sl[2][threadIdx.x] + sl[3][threadIdx.x]);
}

// ---- K7 -------------------------------------------------------------------------------------------
// 1024 threads (round 5: with 256 the one workgroup walked 16,800 x 3 partials in 66 dependent trips, 22 us of the main chain for three
// numbers); the target-score total is summed exactly as loss_grad_kernel's blocks sum it (the first 256 threads, same order)
__global__ __launch_bounds__(1024) void loss_finalize_kernel(const float* loss_part, int nblocks, const float* tss_part, int n_tss_part,
                                                             float gain_box, float gain_cls, float gain_dfl, float* items, float* aux) {
  __shared__ double sw[16][3];
  __shared__ float st[4];
  {
    float v = 0.f;
    if (threadIdx.x < 256) {
      for (int i = threadIdx.x; i < n_tss_part; i += 256) v += tss_part[i];
      v = cvx_wave_sum64(v);
      if ((threadIdx.x & 63) == 0) st[threadIdx.x >> 6] = v;
    }
  }
  double acc[3] = {0, 0, 0};
  for (int i = threadIdx.x; i < nblocks; i += 1024)
    for (int k = 0; k < 3; ++k) acc[k] += (double)loss_part[(long long)i * 3 + k];
  for (int k = 0; k < 3; ++k) {
    for (int o = 1; o < 64; o <<= 1) acc[k] += __shfl_xor(acc[k], o);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float tss_raw = (st[0] + st[1]) + (st[2] + st[3]);
    const float tss = fmaxf(tss_raw, 1.f);
    const float gains[3] = {gain_box, gain_cls, gain_dfl};
    for (int k = 0; k < 3; ++k) {
      double t = 0;
      for (int w = 0; w < 16; ++w) t += sw[w][k];
      items[k] = (float)(t / (double)tss) * gains[k];
    }
    if (aux) aux[0] = tss_raw;
  }
}

// zero-target fast path: dpred for the class logits only, box/dfl zero
struct Ws {
  float *gtbox, *pbox, *ov, *metric, *norm, *pos_align, *pos_ov, *tss_part, *loss_part, *aux;
  int *gtlabel, *gtb, *gtvalid, *img_start, *img_end, *gt_idx;
  uint8_t* mask;
};

long long align_up(long long x) { return (x + 255) & ~255LL; }

long long carve(char* base, int B, int A, int N, Ws* w) {
  long long off = 0;
  auto take = [&](long long bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes);
    return p;
  };
  const long long BA = (long long)B * A;
  const int Nn = N > 0 ? N : 1;
  Ws t;
  t.gtbox = (float*)take(Nn * 16LL);
  t.gtlabel = (int*)take(Nn * 4LL);
  t.gtb = (int*)take(Nn * 4LL);
  t.gtvalid = (int*)take(Nn * 4LL);
  t.img_start = (int*)take(B * 4LL);
  t.img_end = (int*)take(B * 4LL);
  t.pbox = (float*)take(BA * 16);
  t.ov = (float*)take((long long)Nn * A * 4);
  t.metric = (float*)take((long long)Nn * A * 4);
  t.mask = (uint8_t*)take((long long)Nn * A);
  t.gt_idx = (int*)take(BA * 4);
  t.norm = (float*)take(BA * 4);
  t.pos_align = (float*)take(Nn * 4LL);
  t.pos_ov = (float*)take(Nn * 4LL);
  t.tss_part = (float*)take(((BA + 255) / 256) * 4);
  t.loss_part = (float*)take(((BA + 15) / 16) * 12);
  t.aux = (float*)take(64);
  if (w) *w = t;
  return off;
}

}  // namespace

extern "C" int64_t cvx_loss_v8_workspace_bytes(int32_t batch, int32_t anchors, int32_t nc, int32_t max_targets) {
  (void)nc;
  return carve(nullptr, batch, anchors, max_targets, nullptr) + 256;
}

extern "C" int cvx_loss_v8(const float* pred, int32_t B, int32_t A, int32_t nc, const float* targets, int32_t N, int32_t max_targets,
                           const int32_t* level_hw, const float* strides, int32_t n_levels, float gain_box, float gain_cls, float gain_dfl,
                           float loss_scale, float* loss_items, void* dpred_f16, void* workspace, int64_t workspace_bytes, void* hip_stream) {
  return cvx_loss_v8_strided(pred, nc + 4 * REG, B, A, nc, targets, N, max_targets, level_hw, strides, n_levels, gain_box, gain_cls, gain_dfl,
                             loss_scale, loss_items, dpred_f16, workspace, workspace_bytes, hip_stream);
}

extern "C" int cvx_loss_v8_strided(const float* pred, int32_t pred_ld, int32_t B, int32_t A, int32_t nc, const float* targets, int32_t N,
                                   int32_t max_targets, const int32_t* level_hw, const float* strides, int32_t n_levels, float gain_box,
                                   float gain_cls, float gain_dfl, float loss_scale, float* loss_items, void* dpred_f16, void* workspace,
                                   int64_t workspace_bytes, void* hip_stream) {
  CVX_CHECK(pred && loss_items && dpred_f16 && workspace && level_hw && strides, "null arguments");
  CVX_CHECK(pred_ld >= nc + 4 * REG && pred_ld % 4 == 0, "pred_ld must cover 64 + nc values and be a multiple of 4");
  CVX_CHECK(n_levels >= 1 && n_levels <= MAXLV, "1..4 levels");
  CVX_CHECK(N >= 0 && N <= max_targets && (N == 0 || targets), "targets");
  CVX_CHECK(workspace_bytes >= cvx_loss_v8_workspace_bytes(B, A, nc, max_targets), "workspace too small");
  CVX_CHECK(((uintptr_t)pred % 16) == 0 && ((uintptr_t)dpred_f16 % 8) == 0, "alignment");
  hipStream_t st = (hipStream_t)hip_stream;
  const int no = pred_ld;  // row pitch of pred and dpred; columns beyond 64 + nc are neither read nor written
  Levels L;
  memset(&L, 0, sizeof(L));
  L.n = n_levels;
  int off = 0;
  for (int i = 0; i < n_levels; ++i) {
    L.a_off[i] = off;
    L.w[i] = level_hw[2 * i + 1];
    L.stride[i] = strides[i];
    off += level_hw[2 * i] * level_hw[2 * i + 1];
  }
  L.a_off[n_levels] = off;
  CVX_CHECK(off == A, "level sizes do not add up to the anchor count");
  const float img_h = level_hw[0] * strides[0], img_w = level_hw[1] * strides[0];  // yolo_v8.py:87
  char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  Ws w;
  carve(base, B, A, max_targets, &w);
  const long long BA = (long long)B * A;
  const int n_tss = (int)((BA + 255) / 256);
  const int n_lb = (int)((BA + 15) / 16);
  half_t* dpred = (half_t*)dpred_f16;

  {  // img_start | img_end are carved back to back: one fill
    char *a = (char*)w.img_start, *b = (char*)w.img_end;
    if (b > a && b - a <= 4096) {
      CVX_HIP(hipMemsetAsync(a, 0, (size_t)(b - a) + (size_t)B * 4, st));
    } else {
      CVX_HIP(hipMemsetAsync(w.img_start, 0, B * 4, st));
      CVX_HIP(hipMemsetAsync(w.img_end, 0, B * 4, st));
    }
  }
  if (N > 0) {
    hipLaunchKernelGGL(tgt_prep_kernel, dim3(cvx_cdiv(N, 128)), dim3(128), 0, st, targets, N, B, img_w, img_h, w.gtbox, w.gtlabel, w.gtb,
                       w.gtvalid, w.img_start, w.img_end);
    hipLaunchKernelGGL(pred_prep_kernel, dim3(cvx_cdiv(BA * 4, 256)), dim3(256), 0, st, pred, B, A, no, L, w.pbox);
    hipLaunchKernelGGL(tal_metric_kernel, dim3(N), dim3(TAL_TPB), (size_t)A * 4, st, pred, A, no, nc, L, w.pbox, w.gtbox, w.gtlabel, w.gtb,
                       w.gtvalid, w.ov, w.metric, w.mask);
  }
  hipLaunchKernelGGL(tal_resolve_kernel, dim3(cvx_cdiv(BA, 256)), dim3(256), 0, st, B, A, w.img_start, w.img_end, w.ov, w.mask, w.gt_idx);
  if (N > 0)
    hipLaunchKernelGGL(tal_posmax_kernel, dim3(N), dim3(TAL_TPB), 0, st, A, w.gtb, w.gt_idx, w.ov, w.metric, w.pos_align, w.pos_ov);
  hipLaunchKernelGGL(tal_norm_kernel, dim3(n_tss), dim3(256), 0, st, BA, A, w.gt_idx, w.metric, w.pos_align, w.pos_ov, w.norm, w.tss_part);
  hipLaunchKernelGGL(loss_grad_kernel, dim3(n_lb), dim3(256), 0, st, pred, B, A, no, nc, L, w.pbox, w.gtbox, w.gtlabel, w.gt_idx, w.norm,
                     w.tss_part, n_tss, gain_box, gain_cls, gain_dfl, loss_scale, dpred, w.loss_part);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, st, w.loss_part, n_lb, w.tss_part, n_tss, gain_box, gain_cls, gain_dfl,
                     loss_items, w.aux);
  CVX_HIP(hipGetLastError());
  return 0;
}

// Inspection of the last cvx_loss_v8* call that used `workspace` (same batch / anchors / max_targets): per anchor the
// index of the assigned target row (-1 = background) and its normalised target score
// (TaskAlignedAssigner outputs target_gt_idx / fg_mask / target_scores, core/utils/bboxes.py:330-345).
extern "C" int cvx_loss_v8_assignment(const void* workspace, int32_t B, int32_t A, int32_t max_targets, int32_t* gt_index, float* norm_score,
                                      void* hip_stream) {
  CVX_CHECK(workspace && gt_index && norm_score && B > 0 && A > 0, "bad arguments");
  char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  Ws w;
  carve(base, B, A, max_targets, &w);
  hipStream_t st = (hipStream_t)hip_stream;
  CVX_HIP(hipMemcpyAsync(gt_index, w.gt_idx, (size_t)B * A * 4, hipMemcpyDeviceToDevice, st));
  CVX_HIP(hipMemcpyAsync(norm_score, w.norm, (size_t)B * A * 4, hipMemcpyDeviceToDevice, st));
  return 0;
}

// ---- yolo8_collate dict -> target rows ----------------------------------------------------------------------
// batch_idx (N), cls (N), bboxes (N,4) -> rows [batch_idx, cls, cx, cy, w, h] grouped by image, the relative order inside
// an image kept (what Loss.preprocess does row by row, core/algorithms/yolo_v8.py:51-65).  Rank = number of rows that
// must precede: O(N^2) compares, N is a few hundred at most.
namespace {
__global__ __launch_bounds__(256) void pack_targets_kernel(const float* bi, const float* cls, const float* box, int N, float* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float b = bi[i];
  int rank = 0;
  for (int j = 0; j < N; ++j) {
    const float bj = bi[j];
    rank += (bj < b || (bj == b && j < i)) ? 1 : 0;
  }
  float* o = out + (long long)rank * 6;
  o[0] = b;
  o[1] = cls[i];
  o[2] = box[i * 4 + 0];
  o[3] = box[i * 4 + 1];
  o[4] = box[i * 4 + 2];
  o[5] = box[i * 4 + 3];
}
}  // namespace

extern "C" int cvx_pack_targets(const float* batch_idx, const float* cls, const float* bboxes, int32_t n, float* rows, void* hip_stream) {
  if (n <= 0) return 0;
  CVX_CHECK(batch_idx && cls && bboxes && rows, "null arguments");
  hipLaunchKernelGGL(pack_targets_kernel, dim3(cvx_cdiv(n, 256)), dim3(256), 0, (hipStream_t)hip_stream, batch_idx, cls, bboxes, n, rows);
  CVX_HIP(hipGetLastError());
  return 0;
}
