// YOLOv7's loss on gfx950: candidate generation, SimOTA assignment per image, CIoU / objectness / class terms -- forward value AND the
// gradient w.r.t. the head rows in one pass chain (no autograd tape, no host round trip, no dynamic shapes on the host).
//
// Reference semantics (core/loss/yolo7_loss.py):
//   find_3_positive   :340-398   per level, for each of the five cell offsets in turn: every (anchor, target) pair whose size ratio to the
//                                anchor is < 4 and whose centre lies in the half cell facing that neighbour -> ordered candidate list
//   build_targets     :129-338   per image over its candidates of all levels: IoU and class cost matrices, dynamic k = int(sum of the top-20
//                                IoUs) >= 1 per ground truth, the k cheapest candidates per ground truth, candidates claimed by several
//                                ground truths go to the cheapest one
//   __call__          :38-127    per level: CIoU loss (mean over the matched entries), objectness BCE against the clamped IoU over ALL cells
//                                (x balance 0.4 / 1 / 4), class BCE (mean); total = 0.05 box + obj_ratio obj + cls_ratio cls
// Duplicates (one cell matched through two candidates) are kept as in the reference: both count in the means, and the objectness target
// of the cell is the IoU of the LAST entry in list order (index_put on the CPU) -- realised with an atomic maximum on (order, value) keys.
//
//   K1 y7_cand     one workgroup per level: ordered compaction of the 5 x 3 x N candidate slots
//   K2 y7_assign   one workgroup per image: its candidates in order, cost matrices in scratch, one wave per ground truth for the dynamic k
//                  and the k-cheapest selection, conflict resolution, ordered matched lists per level
//   K3 y7_entry    per matched entry: CIoU (value + analytic gradient), class BCE, objectness-target key, gradients (atomic adds)
//   K4 y7_obj      per cell: objectness BCE and gradient
//   K5 y7_final    gradient buffer -> fp16 x loss scale, loss items
#include <algorithm>
#include <cstring>
#include "cvx_common.h"
#include "../../include/cvx_engine.h"

namespace {

constexpr int Y7_GMAX = 64;             // ground truths per image
constexpr int Y7_CMAX = 45 * Y7_GMAX;   // candidates per image: 5 offsets x 3 anchors x 3 levels per ground truth

struct Y7Level {
  int h, w, a_off;
  float anc[3][2];   // anchors in grid units (float32 of anchor_px / stride)
  float stride, balance;
};
struct Y7Geom {
  Y7Level lv[3];
  int B, A, ld, nc, N;
};
struct Y7State {
  int cand_count[3], n_matched[3], bad;
  double box_sum[3], cls_sum[3], obj_sum[3];
};

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float bce_logits(float x, float t) { return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ const float* cell(const float* rows, const Y7Geom& G, int lv, int b, int a, int gj, int gi) {
  return rows + ((long long)b * G.A + G.lv[lv].a_off + gj * G.lv[lv].w + gi) * G.ld + a * (5 + G.nc);
}

// CIoU of box a (x1, y1, x2, y2) against box b and its gradient w.r.t. a's corners, alpha detached (core/utils/iou.py:184-218)
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

// ordered compaction helper: position of this thread's flagged element among the block's flagged elements (block of NT threads)
template <int NT>
__device__ __forceinline__ int block_rank(bool flag, int* s_cnt, int* total) {
  const unsigned long long bal = __ballot(flag);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) s_cnt[wave] = __popcll(bal);
  __syncthreads();
  int before = 0, tot = 0;
  for (int q = 0; q < NT / 64; ++q) {
    if (q < wave) before += s_cnt[q];
    tot += s_cnt[q];
  }
  __syncthreads();
  *total = tot;
  return before + __popcll(bal & ((1ull << lane) - 1ull));
}

// ---- K1 -------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void y7_cand_kernel(const float* targets, Y7Geom G, int* cand /* [3][cap][5] */, int cap, Y7State* st) {
  __shared__ int s_cnt[16];
  const int lv = blockIdx.x;
  const Y7Level L = G.lv[lv];
  int* out = cand + (long long)lv * cap * 5;
  int base_out = 0;
  const int slots = 15 * G.N;
  for (int s0 = 0; s0 < slots; s0 += 1024) {
    const int s = s0 + threadIdx.x;
    bool flag = false;
    int e[5] = {0, 0, 0, 0, 0};
    if (s < slots) {
      const int oi = s / (3 * G.N), rem = s - oi * 3 * G.N, a = rem / G.N, n = rem - a * G.N;
      const float* t = targets + (long long)n * 6;
      const float gx = t[2] * (float)L.w, gy = t[3] * (float)L.h, gw = t[4] * (float)L.w, gh = t[5] * (float)L.h;
      const float rw = gw / L.anc[a][0], rh = gh / L.anc[a][1];
      const bool ok = fmaxf(fmaxf(rw, 1.f / rw), fmaxf(rh, 1.f / rh)) < 4.f;
      const float ix = (float)L.w - gx, iy = (float)L.h - gy;
      bool cond = true;
      float ox = 0.f, oy = 0.f;
      if (oi == 1) cond = fmodf(gx, 1.f) < 0.5f && gx > 1.f, ox = 0.5f;
      if (oi == 2) cond = fmodf(gy, 1.f) < 0.5f && gy > 1.f, oy = 0.5f;
      if (oi == 3) cond = fmodf(ix, 1.f) < 0.5f && ix > 1.f, ox = -0.5f;
      if (oi == 4) cond = fmodf(iy, 1.f) < 0.5f && iy > 1.f, oy = -0.5f;
      flag = ok && cond;
      e[0] = (int)t[0];
      e[1] = a;
      e[2] = min(max((int)(gy - oy), 0), L.h - 1);
      e[3] = min(max((int)(gx - ox), 0), L.w - 1);
      e[4] = n;
    }
    int tot;
    const int pos = base_out + block_rank<1024>(flag, s_cnt, &tot);
    if (flag && pos < cap)
      for (int q = 0; q < 5; ++q) out[(long long)pos * 5 + q] = e[q];
    base_out += tot;
  }
  if (threadIdx.x == 0) st->cand_count[lv] = min(base_out, cap);
}

// ---- K2 -------------------------------------------------------------------------------------------------------------------------------
struct Y7Scratch {  // per image
  int* ent;            // [CMAX][4]: level, anchor, gj, gi
  float* box;          // [CMAX][4] predicted corners in pixels
  float* cost;         // [GMAX][CMAX]
  float* iou;          // [GMAX][CMAX]
  unsigned char* match;  // [GMAX][CMAX]
  int* gt_of;          // [CMAX]
};

__global__ __launch_bounds__(256) void y7_assign_kernel(const float* rows, const float* targets, Y7Geom G, float img_size, const int* cand, int cap,
                                                        int* ent_all, float* box_all, float* cost_all, float* iou_all, unsigned char* match_all,
                                                        int* gtof_all, int* matched /* [3][B][CMAX][4] */, int* mcount /* [B][3] */, Y7State* st) {
  __shared__ int s_cnt[4], s_glist[Y7_GMAX], s_G, s_C;
  const int b = blockIdx.x;
  int* ent = ent_all + (long long)b * Y7_CMAX * 4;
  float* box = box_all + (long long)b * Y7_CMAX * 4;
  float* cost = cost_all + (long long)b * Y7_GMAX * Y7_CMAX;
  float* iou = iou_all + (long long)b * Y7_GMAX * Y7_CMAX;
  unsigned char* match = match_all + (long long)b * Y7_GMAX * Y7_CMAX;
  int* gt_of = gtof_all + (long long)b * Y7_CMAX;
  if (threadIdx.x == 0) {
    int g = 0;
    for (int n = 0; n < G.N; ++n)
      if ((int)targets[(long long)n * 6] == b) {
        if (g < Y7_GMAX)
          s_glist[g++] = n;
        else
          atomicOr(&st->bad, 1);
      }
    s_G = g;
  }
  if (threadIdx.x < 3) mcount[b * 3 + threadIdx.x] = 0;
  __syncthreads();
  const int NG = s_G;
  if (NG == 0) return;
  // (b) this image's candidates, level by level, in list order
  int C = 0;
  for (int lv = 0; lv < 3; ++lv) {
    const int cnt = st->cand_count[lv];
    const int* cl = cand + (long long)lv * cap * 5;
    const Y7Level L = G.lv[lv];
    for (int c0 = 0; c0 < cnt; c0 += 256) {
      const int c = c0 + threadIdx.x;
      const bool flag = c < cnt && cl[(long long)c * 5] == b;
      int tot;
      const int pos = C + block_rank<256>(flag, s_cnt, &tot);
      if (flag) {
        if (pos < Y7_CMAX) {
          const int a = cl[(long long)c * 5 + 1], gj = cl[(long long)c * 5 + 2], gi = cl[(long long)c * 5 + 3];
          ent[pos * 4] = lv;
          ent[pos * 4 + 1] = a;
          ent[pos * 4 + 2] = gj;
          ent[pos * 4 + 3] = gi;
          const float* v = cell(rows, G, lv, b, a, gj, gi);
          const float px = (sigm(v[0]) * 2.f - 0.5f + (float)gi) * L.stride, py = (sigm(v[1]) * 2.f - 0.5f + (float)gj) * L.stride;
          const float sw = sigm(v[2]) * 2.f, sh = sigm(v[3]) * 2.f;
          const float pw = sw * sw * L.anc[a][0] * L.stride, ph = sh * sh * L.anc[a][1] * L.stride;
          box[pos * 4] = px - pw / 2;
          box[pos * 4 + 1] = py - ph / 2;
          box[pos * 4 + 2] = px + pw / 2;
          box[pos * 4 + 3] = py + ph / 2;
        } else {
          atomicOr(&st->bad, 2);
        }
      }
      C += tot;
    }
  }
  C = min(C, Y7_CMAX);
  if (C == 0) return;
  __syncthreads();
  // (c) cost matrices
  for (int idx = threadIdx.x; idx < NG * C; idx += 256) {
    const int g = idx / C, c = idx - g * C;
    const float* t = targets + (long long)s_glist[g] * 6;
    const float tx = t[2] * img_size, ty = t[3] * img_size, tw = t[4] * img_size, th = t[5] * img_size;
    const float tx1 = tx - tw / 2, ty1 = ty - th / 2, tx2 = tx + tw / 2, ty2 = ty + th / 2;
    const float* bx = box + c * 4;
    const float iw = fmaxf(fminf(tx2, bx[2]) - fmaxf(tx1, bx[0]), 0.f), ih = fmaxf(fminf(ty2, bx[3]) - fmaxf(ty1, bx[1]), 0.f);
    const float inter = iw * ih;
    const float io = inter / ((tx2 - tx1) * (ty2 - ty1) + (bx[2] - bx[0]) * (bx[3] - bx[1]) - inter);
    const float* v = cell(rows, G, ent[c * 4], b, ent[c * 4 + 1], ent[c * 4 + 2], ent[c * 4 + 3]);
    const float so = sigm(v[4]);
    const int tc = (int)t[1];
    float cl = 0.f;
    for (int k = 0; k < G.nc; ++k) {
      const float y = sqrtf(sigm(v[5 + k]) * so);
      cl += bce_logits(logf(y / (1.f - y)), k == tc ? 1.f : 0.f);
    }
    iou[g * Y7_CMAX + c] = io;
    cost[g * Y7_CMAX + c] = cl + 3.f * (-logf(io + 1e-8f));
    match[g * Y7_CMAX + c] = 0;
  }
  __syncthreads();
  // (d, e) one wave per ground truth: dynamic k, then the k cheapest candidates
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int g = wave; g < NG; g += 4) {
    unsigned char* mg = match + g * Y7_CMAX;
    const float* ig = iou + g * Y7_CMAX;
    const float* cg = cost + g * Y7_CMAX;
    float sum = 0.f;
    const int rounds = min(20, C);
    for (int r = 0; r < rounds; ++r) {
      float bv = -INFINITY;
      int bi = 0x7fffffff;
      for (int c = lane; c < C; c += 64)
        if (!mg[c] && (ig[c] > bv || (ig[c] == bv && c < bi))) bv = ig[c], bi = c;
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
      }
      if (bi == 0x7fffffff) break;
      sum += bv;
      if (lane == 0) mg[bi] = 1;
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();
    }
    for (int c = lane; c < C; c += 64) mg[c] = 0;
    __threadfence_block();
    int k = (int)sum;
    if (k < 1) k = 1;
    for (int r = 0; r < k && r < C; ++r) {
      float bv = INFINITY;
      int bi = 0x7fffffff;
      for (int c = lane; c < C; c += 64)
        if (!mg[c] && (cg[c] < bv || (cg[c] == bv && c < bi))) bv = cg[c], bi = c;
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov < bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
      }
      if (bi == 0x7fffffff) break;
      if (lane == 0) mg[bi] = 1;
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();
    }
  }
  __syncthreads();
  // (f) a candidate claimed by several ground truths goes to the cheapest one
  for (int c = threadIdx.x; c < C; c += 256) {
    int cnt = 0, gsel = -1;
    for (int g = 0; g < NG; ++g)
      if (match[g * Y7_CMAX + c]) ++cnt, gsel = gsel < 0 ? g : gsel;
    if (cnt > 1) {
      float bv = INFINITY;
      for (int g = 0; g < NG; ++g)
        if (cost[g * Y7_CMAX + c] < bv) bv = cost[g * Y7_CMAX + c], gsel = g;
    }
    gt_of[c] = gsel;
  }
  __syncthreads();
  // (g) ordered matched lists per level
  for (int lv = 0; lv < 3; ++lv) {
    int M = 0;
    int* ml = matched + (((long long)lv * G.B + b) * Y7_CMAX) * 4;
    for (int c0 = 0; c0 < C; c0 += 256) {
      const int c = c0 + threadIdx.x;
      const bool flag = c < C && ent[c * 4] == lv && gt_of[c] >= 0;
      int tot;
      const int pos = M + block_rank<256>(flag, s_cnt, &tot);
      if (flag) {
        ml[pos * 4] = ent[c * 4 + 1];
        ml[pos * 4 + 1] = ent[c * 4 + 2];
        ml[pos * 4 + 2] = ent[c * 4 + 3];
        ml[pos * 4 + 3] = s_glist[gt_of[c]];
      }
      M += tot;
    }
    if (threadIdx.x == 0) {
      mcount[b * 3 + lv] = M;
      atomicAdd(&st->n_matched[lv], M);
    }
  }
}

// ---- K3 -------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void y7_entry_kernel(const float* rows, const float* targets, Y7Geom G, const int* matched, const int* mcount,
                                                       float cp, float cn, float box_w, float cls_w, unsigned long long* tobj, float* gbuf, Y7State* st) {
  __shared__ double s_sum[2][4];
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per_lv = (long long)G.B * Y7_CMAX;
  const int lv = blockIdx.y;
  double lbox = 0, lcls = 0;
  if (idx < per_lv) {
    const int b = (int)(idx / Y7_CMAX), slot = (int)(idx - (long long)b * Y7_CMAX);
    if (slot < mcount[b * 3 + lv]) {
      const Y7Level L = G.lv[lv];
      const int* m = matched + (((long long)lv * G.B + b) * Y7_CMAX + slot) * 4;
      const int a = m[0], gj = m[1], gi = m[2], n = m[3];
      const float* v = cell(rows, G, lv, b, a, gj, gi);
      float* gv = gbuf + (v - rows);
      const float* t = targets + (long long)n * 6;
      const float s0 = sigm(v[0]), s1 = sigm(v[1]), s2 = sigm(v[2]), s3 = sigm(v[3]);
      const float x = s0 * 2.f - 0.5f, y = s1 * 2.f - 0.5f, w = (s2 * 2.f) * (s2 * 2.f) * L.anc[a][0], h = (s3 * 2.f) * (s3 * 2.f) * L.anc[a][1];
      const float tx = t[2] * (float)L.w - (float)gi, ty = t[3] * (float)L.h - (float)gj, tw = t[4] * (float)L.w, th = t[5] * (float)L.h;
      float g4[4];
      const float ci = ciou_grad(x - w / 2, y - h / 2, x + w / 2, y + h / 2, tx - tw / 2, ty - th / 2, tx + tw / 2, ty + th / 2, g4);
      lbox = (double)(1.f - ci);
      const int nm = st->n_matched[lv];
      const float kb = -box_w / (float)nm;                       // d(1 - ciou).mean() * 0.05
      atomicAdd(&gv[0], kb * (g4[0] + g4[2]) * 2.f * s0 * (1.f - s0));
      atomicAdd(&gv[1], kb * (g4[1] + g4[3]) * 2.f * s1 * (1.f - s1));
      atomicAdd(&gv[2], kb * (g4[2] - g4[0]) * 0.5f * 8.f * s2 * s2 * (1.f - s2) * L.anc[a][0]);
      atomicAdd(&gv[3], kb * (g4[3] - g4[1]) * 0.5f * 8.f * s3 * s3 * (1.f - s3) * L.anc[a][1]);
      const int tc = (int)t[1];
      const float kc = cls_w / ((float)nm * (float)G.nc);
      for (int k = 0; k < G.nc; ++k) {
        const float tt = k == tc ? cp : cn;
        lcls += (double)bce_logits(v[5 + k], tt);
        atomicAdd(&gv[5 + k], kc * (sigm(v[5 + k]) - tt));
      }
      // objectness target of the cell: the LAST entry in list order wins -> maximum over (order + 1, value) keys
      const unsigned long long key = ((unsigned long long)(idx + 1) << 32) | (unsigned long long)__float_as_uint(fmaxf(ci, 0.f));
      long long cellidx = 0;
      for (int q = 0; q < lv; ++q) cellidx += (long long)G.B * 3 * G.lv[q].h * G.lv[q].w;
      cellidx += (((long long)b * 3 + a) * L.h + gj) * L.w + gi;
      atomicMax(&tobj[cellidx], key);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    lbox += __shfl_xor(lbox, o);
    lcls += __shfl_xor(lcls, o);
  }
  if ((threadIdx.x & 63) == 0) {
    s_sum[0][threadIdx.x >> 6] = lbox;
    s_sum[1][threadIdx.x >> 6] = lcls;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double sb = (s_sum[0][0] + s_sum[0][1]) + (s_sum[0][2] + s_sum[0][3]), sc = (s_sum[1][0] + s_sum[1][1]) + (s_sum[1][2] + s_sum[1][3]);
    if (sb != 0.0) atomicAdd(&st->box_sum[lv], sb);
    if (sc != 0.0) atomicAdd(&st->cls_sum[lv], sc);
  }
}

// ---- K4 -------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void y7_obj_kernel(const float* rows, Y7Geom G, const unsigned long long* tobj, float obj_w, float* gbuf, Y7State* st) {
  __shared__ double s_sum[4];
  const int lv = blockIdx.y;
  const Y7Level L = G.lv[lv];
  const long long ncell = (long long)G.B * 3 * L.h * L.w;
  double lo = 0;
  long long base = 0;
  for (int q = 0; q < lv; ++q) base += (long long)G.B * 3 * G.lv[q].h * G.lv[q].w;
  // grid-stride over the level's cells: the launch is capped at 128 workgroups per level, because every workgroup ends in a double atomic on
  // the level's one sum and those serialise (a workgroup per 256 cells: 7200 atomics, 90 us)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < ncell; i += (long long)gridDim.x * 256) {
    const unsigned long long key = tobj[base + i];
    const float tv = key ? __uint_as_float((unsigned)(key & 0xFFFFFFFFull)) : 0.f;
    const int gi = (int)(i % L.w);
    long long t = i / L.w;
    const int gj = (int)(t % L.h);
    t /= L.h;
    const int a = (int)(t % 3), b = (int)(t / 3);
    const float* v = cell(rows, G, lv, b, a, gj, gi);
    lo += (double)bce_logits(v[4], tv);
    gbuf[(v - rows) + 4] = obj_w * L.balance / (float)ncell * (sigm(v[4]) - tv);
  }
  for (int o = 32; o > 0; o >>= 1) lo += __shfl_xor(lo, o);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = lo;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&st->obj_sum[lv], (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]));
}

// ---- K5 -------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void y7_final_kernel(const float* gbuf, long long n, float loss_scale, half_t* dpred, Y7Geom G, const Y7State* st,
                                                       float box_w, float obj_w, float cls_w, float* items, int* bad) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dpred[i] = (half_t)(gbuf[i] * loss_scale);
  if (i == 0) {
    double box = 0, obj = 0, cls = 0;
    for (int lv = 0; lv < 3; ++lv) {
      const double nm = (double)st->n_matched[lv];
      if (nm > 0) {
        box += st->box_sum[lv] / nm;
        cls += st->cls_sum[lv] / (nm * G.nc);
      }
      obj += st->obj_sum[lv] / ((double)G.B * 3 * G.lv[lv].h * G.lv[lv].w) * (double)G.lv[lv].balance;
    }
    items[1] = (float)(box * box_w);
    items[2] = (float)(obj * obj_w);
    items[3] = (float)(cls * cls_w);
    items[0] = items[1] + items[2] + items[3];
    *bad = st->bad;
  }
}

size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

struct Y7Layout {
  size_t state, cand, ent, box, cost, iou, match, gtof, matched, mcount, tobj, gbuf, total;
};
Y7Layout y7_layout(int B, int A, int ld, int N, long long cells) {
  Y7Layout l;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t r = o;
    o += al(bytes);
    return r;
  };
  l.state = take(sizeof(Y7State));
  l.cand = take((size_t)3 * std::max(15 * N, 1) * 5 * 4);
  l.ent = take((size_t)B * Y7_CMAX * 4 * 4);
  l.box = take((size_t)B * Y7_CMAX * 4 * 4);
  l.cost = take((size_t)B * Y7_GMAX * Y7_CMAX * 4);
  l.iou = take((size_t)B * Y7_GMAX * Y7_CMAX * 4);
  l.match = take((size_t)B * Y7_GMAX * Y7_CMAX);
  l.gtof = take((size_t)B * Y7_CMAX * 4);
  l.matched = take((size_t)3 * B * Y7_CMAX * 4 * 4);
  l.mcount = take((size_t)B * 3 * 4);
  l.tobj = take((size_t)cells * 8);
  l.gbuf = take((size_t)B * A * ld * 4);
  l.total = o;
  return l;
}

}  // namespace

extern "C" int64_t cvx_yolo7_loss_workspace_bytes(int32_t batch, int32_t anchors, int32_t ld, int32_t n_targets) {
  return (int64_t)y7_layout(batch, anchors, ld, n_targets, (long long)batch * 3 * anchors).total;
}

extern "C" int cvx_yolo7_loss(const float* rows_f32, int32_t ld, int32_t batch, int32_t nc, const int32_t* level_hw, const float* anchors_px,
                              const float* strides, const float* targets, int32_t n_targets, float img_size, float box_ratio, float obj_ratio,
                              float cls_ratio, float label_smoothing, float loss_scale, float* loss_items, void* dpred_f16, int32_t* bad,
                              void* workspace, void* hip_stream) {
  CVX_CHECK(rows_f32 && level_hw && anchors_px && strides && loss_items && dpred_f16 && bad && workspace && (targets || n_targets == 0), "null arguments");
  CVX_CHECK(batch > 0 && nc > 0 && 3 * (5 + nc) <= ld && n_targets >= 0 && loss_scale > 0.f, "bad sizes");
  hipStream_t st = (hipStream_t)hip_stream;
  Y7Geom G;
  memset(&G, 0, sizeof(G));
  static const float balance[3] = {0.4f, 1.0f, 4.0f};
  int A = 0;
  long long cells = 0;
  for (int lv = 0; lv < 3; ++lv) {
    G.lv[lv].h = level_hw[2 * lv];
    G.lv[lv].w = level_hw[2 * lv + 1];
    G.lv[lv].a_off = A;
    A += G.lv[lv].h * G.lv[lv].w;
    G.lv[lv].stride = strides[lv];
    G.lv[lv].balance = balance[lv];
    for (int a = 0; a < 3; ++a)
      for (int q = 0; q < 2; ++q) G.lv[lv].anc[a][q] = (float)((double)anchors_px[(lv * 3 + a) * 2 + q] / (double)strides[lv]);
  }
  cells = (long long)batch * 3 * A;
  G.B = batch;
  G.A = A;
  G.ld = ld;
  G.nc = nc;
  G.N = n_targets;
  const Y7Layout l = y7_layout(batch, A, ld, n_targets, cells);
  char* w = (char*)workspace;
  Y7State* state = (Y7State*)(w + l.state);
  CVX_HIP(hipMemsetAsync(w + l.state, 0, al(sizeof(Y7State)), st));
  CVX_HIP(hipMemsetAsync(w + l.mcount, 0, (size_t)batch * 3 * 4, st));
  CVX_HIP(hipMemsetAsync(w + l.tobj, 0, (size_t)cells * 8, st));
  CVX_HIP(hipMemsetAsync(w + l.gbuf, 0, (size_t)batch * A * ld * 4, st));
  const int cap = std::max(15 * n_targets, 1);
  if (n_targets > 0) {
    hipLaunchKernelGGL(y7_cand_kernel, dim3(3), dim3(1024), 0, st, targets, G, (int*)(w + l.cand), cap, state);
    hipLaunchKernelGGL(y7_assign_kernel, dim3(batch), dim3(256), 0, st, rows_f32, targets, G, img_size, (const int*)(w + l.cand), cap, (int*)(w + l.ent),
                       (float*)(w + l.box), (float*)(w + l.cost), (float*)(w + l.iou), (unsigned char*)(w + l.match), (int*)(w + l.gtof),
                       (int*)(w + l.matched), (int*)(w + l.mcount), state);
    const float cp = 1.f - 0.5f * label_smoothing, cn = 0.5f * label_smoothing;
    hipLaunchKernelGGL(y7_entry_kernel, dim3((unsigned)cvx_cdiv((long long)batch * Y7_CMAX, 256), 3), dim3(256), 0, st, rows_f32, targets, G,
                       (const int*)(w + l.matched), (const int*)(w + l.mcount), cp, cn, box_ratio, cls_ratio, (unsigned long long*)(w + l.tobj),
                       (float*)(w + l.gbuf), state);
  }
  long long maxcell = 0;
  for (int lv = 0; lv < 3; ++lv) maxcell = std::max(maxcell, (long long)batch * 3 * G.lv[lv].h * G.lv[lv].w);
  hipLaunchKernelGGL(y7_obj_kernel, dim3((unsigned)std::min<long long>(128, cvx_cdiv(maxcell, 256)), 3), dim3(256), 0, st, rows_f32, G, (const unsigned long long*)(w + l.tobj),
                     obj_ratio, (float*)(w + l.gbuf), state);
  const long long n = (long long)batch * A * ld;
  hipLaunchKernelGGL(y7_final_kernel, dim3((unsigned)cvx_cdiv(n, 256)), dim3(256), 0, st, (const float*)(w + l.gbuf), n, loss_scale, (half_t*)dpred_f16, G,
                     state, box_ratio, obj_ratio, cls_ratio, loss_items, bad);
  CVX_HIP(hipGetLastError());
  return 0;
}
