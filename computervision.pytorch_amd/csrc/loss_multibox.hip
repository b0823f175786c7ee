// SSD's MultiBoxLossV2 on gfx950: softmax cross-entropy with batch-wide hard-negative mining + smooth-L1 on the positives, forward value
// AND the gradient w.r.t. (loc, conf) in one pass chain (no autograd tape, no sort).
//
// Reference semantics (core/loss/multi_box_loss.py:77-192):
//   y_pred = cat(loc, softmax(conf));  conf_loss = -sum_c y_true_c * log(clamp(p_c, 1e-7));  loc_loss = sum smooth_l1(true - pred)
//   pos = y_true[..., -1];  num_pos per image;  num_neg = min(ratio * num_pos, A - num_pos);  k = sum(num_neg) (100 if no image has any)
//   max_confs = sum_{c >= 1} p_c * (1 - pos), flattened over the WHOLE batch;  indices = topk(max_confs, k);  neg = conf_loss[indices]
//   conf = (sum pos conf + sum neg) / sum(num_pos or 1);  loc = sum pos loc / sum(num_pos or 1);  total = (1 - alpha) conf + alpha loc, alpha 0.5
//
// The top-k is a SELECTION, not a sort: an 8-bit radix select over the order-preserving bit patterns of the non-negative keys finds the
// k-th largest key T in four histogram passes; everything above T is taken, ties at T are taken in flat-index order (torch.topk leaves the
// choice among equal keys unspecified).
//
//   K1 mb_anchor     per anchor: softmax, conf / loc loss, key; per-image positives, positive sums
//   K2 mb_plan       one thread: k and the normaliser from the per-image counts
//   K3 mb_hist / mb_pick (x4)  radix select of the k-th largest key
//   K4 mb_ties       one workgroup: which of the keys equal to T are taken (index order)
//   K5 mb_grad       per anchor: negative sum, gradients w.r.t. loc and conf (times grad_scale), loss items by the last block
#include <algorithm>
#include "cvx_common.h"
#include "../../include/cvx_engine.h"

namespace {

struct MbState {
  double pos_conf, pos_loc;
  double norm;               // sum over images of (num_pos or 1)
  unsigned long long k;      // negatives to take, batch-wide
  unsigned prefix, remaining;  // radix select state: key bits fixed so far, rank still to resolve inside the prefix bucket
  unsigned thr, take_ties;   // final threshold key and how many keys equal to it are taken
  unsigned hist[256];
};

__device__ __forceinline__ double wsum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// A workgroup's 256 anchors own one contiguous run of conf (256 x nc1 floats) and of y_true (256 x ld): with STAGED they are copied to
// LDS with coalesced loads and every thread walks its own row there (odd row strides: no bank conflicts); straight from memory a
// thread's row sits 4 * nc1 bytes from its neighbour's, every load instruction touches 64 cache lines and the address path is the
// bound (0.39 ms for 32 x 8732 anchors).  The staged form needs 256 * (nc1 + ld) * 4 bytes <= 64 KB (SSD's 21 classes: 48 KB).
template <bool STAGED>
__device__ __forceinline__ void mb_stage_rows(const float* conf, const float* y_true, long long i0, long long N, int nc1, int ld, float* sconf, float* sy) {
  if (!STAGED) return;
  const int nvalid = (int)(N - i0 < 256 ? N - i0 : 256);
  const float* c0 = conf + i0 * nc1;
  const float* y0 = y_true + i0 * ld;
  for (int j = threadIdx.x; j < nvalid * nc1; j += 256) sconf[j] = c0[j];
  for (int j = threadIdx.x; j < nvalid * ld; j += 256) sy[j] = y0[j];
  __syncthreads();
}

template <bool STAGED>
__global__ __launch_bounds__(256) void mb_anchor_kernel(const float* loc, const float* conf, const float* y_true, int B, int A, int nc1, unsigned* key,
                                                        float* closs, int* num_pos, double* part /* per block: sum cl*pos, sum ll*pos */) {
  extern __shared__ __attribute__((aligned(16))) float srows[];
  __shared__ double sm[2][4];
  const long long N = (long long)B * A;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int ld = 4 + nc1 + 1;
  mb_stage_rows<STAGED>(conf, y_true, (long long)blockIdx.x * 256, N, nc1, ld, srows, srows + 256 * nc1);
  double pc = 0, pl = 0;
  if (i < N) {
    const float* yt = STAGED ? srows + 256 * nc1 + threadIdx.x * ld : y_true + i * ld;
    const float* z = STAGED ? srows + threadIdx.x * nc1 : conf + i * nc1;
    float zmax = -INFINITY;
    for (int c = 0; c < nc1; ++c) zmax = fmaxf(zmax, z[c]);
    float se = 0.f;
    for (int c = 0; c < nc1; ++c) se += expf(z[c] - zmax);
    float cl = 0.f, fg = 0.f;
    for (int c = 0; c < nc1; ++c) {
      const float p = expf(z[c] - zmax) / se;
      cl -= yt[4 + c] * logf(fmaxf(p, 1e-7f));
      if (c >= 1) fg += p;
    }
    const float pos = yt[ld - 1];
    float ll = 0.f;
    for (int j = 0; j < 4; ++j) {
      const float d = yt[j] - loc[i * 4 + j], a = fabsf(d);
      ll += a < 1.f ? 0.5f * d * d : a - 0.5f;
    }
    closs[i] = cl;
    key[i] = __float_as_uint(fg * (1.f - pos));   // >= 0: the bit pattern orders like the value
    pc = (double)(cl * pos);
    pl = (double)(ll * pos);
    if (pos != 0.f) atomicAdd(&num_pos[i / A], 1);
  }
  pc = wsum(pc);
  pl = wsum(pl);
  if ((threadIdx.x & 63) == 0) {
    sm[0][threadIdx.x >> 6] = pc;
    sm[1][threadIdx.x >> 6] = pl;
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // summed in block order by mb_plan_kernel: deterministic, and no same-address atomics (1000+ of them serialise)
    part[2 * (long long)blockIdx.x] = (sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]);
    part[2 * (long long)blockIdx.x + 1] = (sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3]);
  }
}

// fixed-order sum of n doubles at stride `stride` by one 256-thread workgroup (every thread returns the total)
__device__ double block_sum_256(const double* v, int n, int stride, double* sh) {
  double a = 0;
  for (int i = threadIdx.x; i < n; i += 256) a += v[(long long)i * stride];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  const double t = sh[0];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(256) void mb_plan_kernel(const int* num_pos, int B, int A, float ratio, const double* part, int nblocks, MbState* stt) {
  __shared__ double sh[256];
  const double pc = block_sum_256(part, nblocks, 2, sh), pl = block_sum_256(part + 1, nblocks, 2, sh);
  if (threadIdx.x != 0) return;
  stt->pos_conf = pc;
  stt->pos_loc = pl;
  double k = 0, norm = 0;
  int any = 0;
  for (int b = 0; b < B; ++b) {
    const double np = (double)num_pos[b];
    const double nn = fmin((double)ratio * np, (double)A - np);
    if (nn > 0) any = 1;
    k += nn;
    norm += np != 0.0 ? np : 1.0;
  }
  if (!any) k = 100.0;                                   // negatives_for_hard
  const double cap = (double)B * A;
  stt->k = (unsigned long long)(k < cap ? k : cap);      // int(num_neg_batch)
  stt->norm = norm;
  stt->prefix = 0;
  stt->remaining = (unsigned)stt->k;
}

__global__ __launch_bounds__(256) void mb_hist_kernel(const unsigned* key, long long N, int shift, MbState* stt) {
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const unsigned prefix = stt->prefix;
  const unsigned hi_mask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
    const unsigned kx = key[i];
    if ((kx & hi_mask) == (prefix & hi_mask)) atomicAdd(&h[(kx >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&stt->hist[threadIdx.x], h[threadIdx.x]);
}

__global__ void mb_pick_kernel(int shift, MbState* stt) {  // one thread: the bucket (from the top) that holds the rank still looked for
  unsigned rem = stt->remaining;
  int b = 255;
  if (rem == 0) {  // k == 0: nothing is taken; a threshold above every key
    stt->prefix = 0xFFFFFFFFu;
  } else {
    for (; b > 0; --b) {
      if (stt->hist[b] >= rem) break;
      rem -= stt->hist[b];
    }
    stt->prefix |= (unsigned)b << shift;
    stt->remaining = rem;
  }
  for (int q = 0; q < 256; ++q) stt->hist[q] = 0;
  if (shift == 0) {
    stt->thr = stt->prefix;
    stt->take_ties = stt->remaining;  // of the keys equal to thr, this many (lowest indices first) complete the k
  }
}

// Keys equal to the threshold are taken in flat-index order until k is complete.  Rank of a tie = ties in the workgroups before it
// (count per 256-key block, then one exclusive scan) + ties before it inside its block (ballots, in mb_grad_kernel).
__global__ __launch_bounds__(256) void mb_tie_count_kernel(const unsigned* key, long long N, const MbState* stt, unsigned* cnt) {
  __shared__ unsigned sc[4];
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long bal = __ballot(i < N && key[i] == stt->thr);
  if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = (unsigned)__popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) cnt[blockIdx.x] = (sc[0] + sc[1]) + (sc[2] + sc[3]);
}
__global__ __launch_bounds__(1024) void mb_tie_scan_kernel(const unsigned* cnt, int n, unsigned* base) {  // exclusive scan, one workgroup
  __shared__ unsigned sw[16];
  __shared__ unsigned carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + threadIdx.x;
    const unsigned v = i < n ? cnt[i] : 0u;
    unsigned inc = v;
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) sw[wave] = inc;
    __syncthreads();
    unsigned before = carry;
    for (int q = 0; q < wave; ++q) before += sw[q];
    if (i < n) base[i] = before + inc - v;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned t = 0;
      for (int q = 0; q < 16; ++q) t += sw[q];
      carry += t;
    }
    __syncthreads();
  }
}

template <bool STAGED>
__global__ __launch_bounds__(256) void mb_grad_kernel(const float* loc, const float* conf, const float* y_true, int B, int A, int nc1, const unsigned* key,
                                                      const float* closs, const unsigned* tie_base, const MbState* stt, float alpha, float grad_scale,
                                                      float* dloc, float* dconf, double* part_neg) {
  extern __shared__ __attribute__((aligned(16))) float srows[];
  __shared__ double sm[4];
  __shared__ unsigned sties[4];
  const long long N = (long long)B * A;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int ld = 4 + nc1 + 1;
  mb_stage_rows<STAGED>(conf, y_true, (long long)blockIdx.x * 256, N, nc1, ld, srows, srows + 256 * nc1);
  const float inv = (float)((double)grad_scale / stt->norm);
  const unsigned thr = stt->thr;
  double neg = 0;
  const unsigned kx = i < N ? key[i] : 0u;
  const bool tie = i < N && kx == thr;
  const unsigned long long bal = __ballot(tie);
  if ((threadIdx.x & 63) == 0) sties[threadIdx.x >> 6] = (unsigned)__popcll(bal);
  __syncthreads();
  unsigned rank = tie_base[blockIdx.x] + (unsigned)__popcll(bal & ((1ull << (threadIdx.x & 63)) - 1ull));
  for (int q = 0; q < (int)(threadIdx.x >> 6); ++q) rank += sties[q];
  if (i < N) {
    const float* yt = STAGED ? srows + 256 * nc1 + threadIdx.x * ld : y_true + i * ld;
    const float pos = yt[ld - 1];
    const bool taken = stt->k > 0 && (kx > thr || (tie && rank < stt->take_ties));
    if (taken) neg = (double)closs[i];
    const float wc = (pos + (taken ? 1.f : 0.f)) * (1.f - alpha) * inv;   // weight of this anchor's cross-entropy term
    const float* z = STAGED ? srows + threadIdx.x * nc1 : conf + i * nc1;
    float* dz = STAGED ? srows + threadIdx.x * nc1 : dconf + i * nc1;  // staged: in place over the thread's own logits (z[j] is read before dz[j] is written)
    float zmax = -INFINITY;
    for (int c = 0; c < nc1; ++c) zmax = fmaxf(zmax, z[c]);
    float se = 0.f;
    for (int c = 0; c < nc1; ++c) se += expf(z[c] - zmax);
    // d/dz_j of -sum_c y_c log(clamp(p_c)) = sum_c y_c [p_c >= 1e-7] (p_j - delta_jc)
    float ysum = 0.f;
    for (int c = 0; c < nc1; ++c) {
      const float p = expf(z[c] - zmax) / se;
      if (p >= 1e-7f) ysum += yt[4 + c];
    }
    for (int j = 0; j < nc1; ++j) {
      const float p = expf(z[j] - zmax) / se;
      const float yj = p >= 1e-7f ? yt[4 + j] : 0.f;
      dz[j] = wc != 0.f ? wc * (p * ysum - yj) : 0.f;
    }
    const float wl = pos * alpha * inv;
    for (int j = 0; j < 4; ++j) {
      const float d = loc[i * 4 + j] - yt[j];              // d smooth_l1(true - pred) / d pred
      dloc[i * 4 + j] = wl != 0.f ? wl * (fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f)) : 0.f;
    }
  }
  neg = wsum(neg);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = neg;
  __syncthreads();
  if (STAGED) {  // the workgroup's gradient rows leave as one contiguous run
    const long long i0 = (long long)blockIdx.x * 256;
    const int nvalid = (int)(N - i0 < 256 ? N - i0 : 256);
    float* d0 = dconf + i0 * nc1;
    for (int j = threadIdx.x; j < nvalid * nc1; j += 256) d0[j] = srows[j];
  }
  if (threadIdx.x == 0) part_neg[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);  // summed in block order by mb_final_kernel
}
__global__ __launch_bounds__(256) void mb_final_kernel(const double* part_neg, int nblocks, const MbState* stt, float alpha, float* loss_items) {
  __shared__ double sh[256];
  const double neg = block_sum_256(part_neg, nblocks, 1, sh);
  if (threadIdx.x != 0) return;
  const double c = (stt->pos_conf + neg) / stt->norm, l = stt->pos_loc / stt->norm;
  loss_items[0] = (float)(c * (1.0 - (double)alpha) + l * (double)alpha);
  loss_items[1] = (float)l;
  loss_items[2] = (float)c;
}

size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" int64_t cvx_multibox_loss_workspace_bytes(int32_t batch, int32_t anchors) {
  const size_t N = (size_t)batch * anchors;
  const size_t nb = (N + 255) / 256;  // state, positives per image, keys, per-anchor loss, per-block (pos_conf, pos_loc, neg) sums, tie counts + bases
  return (int64_t)(al256(sizeof(MbState)) + al256((size_t)batch * 4) + al256(N * 4) * 2 + al256(nb * 24) + al256(nb * 4) * 2);
}

extern "C" int cvx_multibox_loss(const float* loc, const float* conf, const float* y_true, int32_t batch, int32_t anchors, int32_t nc1, float neg_pos_ratio,
                                 float alpha, float grad_scale, float* loss_items, float* dloc, float* dconf, void* workspace, void* hip_stream) {
  CVX_CHECK(loc && conf && y_true && loss_items && dloc && dconf && workspace, "null arguments");
  CVX_CHECK(batch > 0 && anchors > 0 && nc1 >= 2 && (long long)batch * anchors < (1LL << 31), "bad sizes");
  hipStream_t st = (hipStream_t)hip_stream;
  const long long N = (long long)batch * anchors;
  char* w = (char*)workspace;
  MbState* stt = (MbState*)w;
  w += al256(sizeof(MbState));
  int* num_pos = (int*)w;
  w += al256((size_t)batch * 4);
  unsigned* key = (unsigned*)w;
  w += al256((size_t)N * 4);
  float* closs = (float*)w;
  w += al256((size_t)N * 4);
  const unsigned nb = (unsigned)cvx_cdiv(N, 256);
  double* part = (double*)w;            // [nb][2] positives' sums, then [nb] negatives' sums
  double* part_neg = part + 2 * (size_t)nb;
  w += al256((size_t)nb * 24);
  unsigned* tie_cnt = (unsigned*)w;
  w += al256((size_t)nb * 4);
  unsigned* tie_base = (unsigned*)w;
  CVX_HIP(hipMemsetAsync(workspace, 0, al256(sizeof(MbState)) + al256((size_t)batch * 4), st));
  const size_t stage_bytes = (size_t)256 * (nc1 + 4 + nc1 + 1) * 4;
  const bool staged = stage_bytes <= 64 * 1024;
  if (staged)
    hipLaunchKernelGGL(mb_anchor_kernel<true>, dim3(nb), dim3(256), stage_bytes, st, loc, conf, y_true, batch, anchors, nc1, key, closs, num_pos, part);
  else
    hipLaunchKernelGGL(mb_anchor_kernel<false>, dim3(nb), dim3(256), 0, st, loc, conf, y_true, batch, anchors, nc1, key, closs, num_pos, part);
  hipLaunchKernelGGL(mb_plan_kernel, dim3(1), dim3(256), 0, st, num_pos, batch, anchors, neg_pos_ratio, part, (int)nb, stt);
  const int hb = (int)std::min<long long>(512, nb);
  for (int shift = 24; shift >= 0; shift -= 8) {
    hipLaunchKernelGGL(mb_hist_kernel, dim3(hb), dim3(256), 0, st, key, N, shift, stt);
    hipLaunchKernelGGL(mb_pick_kernel, dim3(1), dim3(1), 0, st, shift, stt);
  }
  hipLaunchKernelGGL(mb_tie_count_kernel, dim3(nb), dim3(256), 0, st, key, N, stt, tie_cnt);
  hipLaunchKernelGGL(mb_tie_scan_kernel, dim3(1), dim3(1024), 0, st, tie_cnt, (int)nb, tie_base);
  if (staged)
    hipLaunchKernelGGL(mb_grad_kernel<true>, dim3(nb), dim3(256), stage_bytes, st, loc, conf, y_true, batch, anchors, nc1, key, closs, tie_base, stt, alpha,
                       grad_scale, dloc, dconf, part_neg);
  else
    hipLaunchKernelGGL(mb_grad_kernel<false>, dim3(nb), dim3(256), 0, st, loc, conf, y_true, batch, anchors, nc1, key, closs, tie_base, stt, alpha,
                       grad_scale, dloc, dconf, part_neg);
  hipLaunchKernelGGL(mb_final_kernel, dim3(1), dim3(256), 0, st, part_neg, (int)nb, stt, alpha, loss_items);
  CVX_HIP(hipGetLastError());
  return 0;
}

// ---- SSD target encoding (Ssd.generate_targets + _encode_box, core/algorithms/ssd.py:327-480): the CPU work of ssd_collate -------------
// Per ground-truth box: IoU with every prior (float64 on float32-valued operands, as the reference's numpy code ends up doing); priors above the overlap
// threshold are "assigned" -- or the single best prior when none is.  Per prior: among the ground truths it is assigned to, the one with the
// largest IoU (first on ties, numpy argmax) -- positive iff that IoU > 0; its box encoded against the prior in float64, the row cast to
// float32.
namespace {

__device__ __forceinline__ double prior_iou(const float* a, double bx0, double by0, double bx1, double by1) {
#pragma clang fp contract(off)  // numpy rounds every product and sum: no fused multiply-adds
  const double w = fmax(fmin((double)a[2], bx1) - fmax((double)a[0], bx0), 0.0), h = fmax(fmin((double)a[3], by1) - fmax((double)a[1], by0), 0.0);
  const double inter = w * h;
  const double area_true = (bx1 - bx0) * (by1 - by0), area_gt = (double)((a[2] - a[0]) * (a[3] - a[1]));  // float32 product of the float32 anchors (:423)
  return inter / (area_true + area_gt - inter);
}
// (cx, cy, w, h) -> corners in float32 (xywh_to_xyxy on the float32 label), then float64: the reference carries the corners in a float64
// array next to the one-hot labels (ssd.py:343), so IoU and encoding run in float64 on float32-valued operands
__device__ __forceinline__ void label_box(const float* l, double* x0, double* y0, double* x1, double* y1) {
#pragma clang fp contract(off)
  *x0 = (double)(l[1] - l[3] / 2);
  *y0 = (double)(l[2] - l[4] / 2);
  *x1 = (double)(l[1] + l[3] / 2);
  *y1 = (double)(l[2] + l[4] / 2);
}

// one workgroup per (image, ground truth): is any prior above the threshold?  else the first prior of maximal IoU is forced
__global__ __launch_bounds__(256) void ssd_force_kernel(const float* labels, const int* counts, int nmax, const float* priors, int A, float thr, int* force) {
  __shared__ double s_v[256];
  __shared__ int s_i[256], s_any;
  const int b = blockIdx.x / nmax, n = blockIdx.x - b * nmax;
  if (threadIdx.x == 0) s_any = 0;
  __syncthreads();
  if (n >= counts[b]) {
    if (threadIdx.x == 0) force[blockIdx.x] = -1;
    return;
  }
  double x0, y0, x1, y1;
  label_box(labels + ((long long)b * nmax + n) * 5, &x0, &y0, &x1, &y1);
  double best = -INFINITY;
  int bi = 0x7fffffff, any = 0;
  for (int a = threadIdx.x; a < A; a += 256) {
    const double iou = prior_iou(priors + (long long)a * 4, x0, y0, x1, y1);
    if (iou > (double)thr) any = 1;
    if (iou > best) {          // strict: the first index of the maximum inside this thread's stride
      best = iou;
      bi = a;
    }
  }
  if (any) s_any = 1;
  s_v[threadIdx.x] = best;
  s_i[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const double v = s_v[threadIdx.x + s];
      const int i = s_i[threadIdx.x + s];
      if (v > s_v[threadIdx.x] || (v == s_v[threadIdx.x] && i < s_i[threadIdx.x])) {
        s_v[threadIdx.x] = v;
        s_i[threadIdx.x] = i;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) force[blockIdx.x] = s_any ? -1 : s_i[0];
}

__global__ __launch_bounds__(256) void ssd_encode_kernel(const float* labels, const int* counts, int nmax, const float* priors, int A, int nc1, float thr,
                                                         float var_xy, float var_wh, const int* force, int B, float* y_true) {
#pragma clang fp contract(off)
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * A) return;
  const int b = (int)(i / A), a = (int)(i - (long long)b * A);
  const float* pr = priors + (long long)a * 4;
  double best = 0.0;
  int bn = -1;
  const int cnt = counts[b];
  for (int n = 0; n < cnt; ++n) {
    double x0, y0, x1, y1;
    label_box(labels + ((long long)b * nmax + n) * 5, &x0, &y0, &x1, &y1);
    const double iou = prior_iou(pr, x0, y0, x1, y1);
    const double stored = (iou > (double)thr || force[b * nmax + n] == a) ? iou : 0.0;
    if (bn < 0 ? stored > 0.0 : stored > best) {   // numpy argmax over the ground truths: the first maximum; positive only when > 0
      best = stored;
      bn = n;
    }
  }
  const int ld = 4 + nc1 + 1;
  float* row = y_true + i * ld;
  for (int c = 0; c < ld; ++c) row[c] = 0.f;
  if (bn < 0) {
    row[4] = 1.f;  // background
    return;
  }
  const float* l = labels + ((long long)b * nmax + bn) * 5;
  double x0, y0, x1, y1;
  label_box(l, &x0, &y0, &x1, &y1);
  const double bcx = 0.5 * (x0 + x1), bcy = 0.5 * (y0 + y1), bw = x1 - x0, bh = y1 - y0;
  // the prior's centre and size are float32 operations on the float32 anchors (assigned_anchors, :455-458), promoted afterwards
  const double acx = (double)((pr[0] + pr[2]) * 0.5f), acy = (double)((pr[1] + pr[3]) * 0.5f), aw = (double)(pr[2] - pr[0]), ah = (double)(pr[3] - pr[1]);
  row[0] = (float)((bcx - acx) / aw / (double)var_xy);
  row[1] = (float)((bcy - acy) / ah / (double)var_xy);
  row[2] = (float)(log(bw / aw) / (double)var_wh);
  row[3] = (float)(log(bh / ah) / (double)var_wh);
  const int cls = (int)l[0] + 1;  // label[:, 1] += 1: column 0 of the one-hot is the background
  if (cls >= 0 && cls < nc1) row[4 + cls] = 1.f;
  row[ld - 1] = 1.f;
}

}  // namespace

extern "C" int cvx_ssd_encode_targets(const float* labels, const int32_t* counts, int32_t batch, int32_t max_boxes, const float* priors, int32_t anchors,
                                      int32_t nc1, float overlap_threshold, float variance_xy, float variance_wh, float* y_true, int32_t* workspace,
                                      void* hip_stream) {
  CVX_CHECK(labels && counts && priors && y_true && workspace && batch > 0 && max_boxes > 0 && anchors > 0 && nc1 >= 2, "bad arguments");
  hipStream_t st = (hipStream_t)hip_stream;
  hipLaunchKernelGGL(ssd_force_kernel, dim3(batch * max_boxes), dim3(256), 0, st, labels, counts, max_boxes, priors, anchors, overlap_threshold, workspace);
  hipLaunchKernelGGL(ssd_encode_kernel, dim3((unsigned)cvx_cdiv((long long)batch * anchors, 256)), dim3(256), 0, st, labels, counts, max_boxes, priors,
                     anchors, nc1, overlap_threshold, variance_xy, variance_wh, workspace, batch, y_true);
  CVX_HIP(hipGetLastError());
  return 0;
}
