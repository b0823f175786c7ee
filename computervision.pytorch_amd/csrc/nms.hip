// Eval tail on gfx950: DFL decode + sigmoid, then class-aware greedy NMS.
//
//   cvx_decode : pred (B,A,no) -> y (B,4+nc,A)                  core/models/yolov8/modules.py:434-446
//   cvx_nms    : y -> kept rows + anchor indices                core/utils/ultralytics_ops.py:131-264
//
// NMS semantics are those of oracle/nms_ref.py (torchvision 0.14.1 batched_nms restated): candidates with
// best-class score > conf, ordered by (score desc, anchor index asc), greedy suppression when
// inter/(area_i+area_j-inter) > iou_thres, evaluated in fp32 with the same operation order (explicit
// round-to-nearest intrinsics: no FMA contraction), first max_det survivors.  Both strategies of the library:
// VANILLA (boxes interact inside a class only) and OFFSET (_batched_nms_coordinate_trick: every box shifted by
// cls * (max coordinate + 1) in fp32, class-agnostic pass over the shifted, re-rounded boxes), plus the library's own
// switch between them by candidate count.
//
// One workgroup per image: the (score,index) keys are sorted with an in-LDS bitonic network
// (<= 16384 candidates = 128 KB of the CU's 160 KB LDS), the suppression matrix is built as 64-bit
// row masks in HBM by all 16 waves, and one wave walks it.
#include <cstring>
#include "cvx_common.h"
#include "../../include/cvx_engine.h"

namespace {

constexpr int REG = 16;
constexpr int MAXLV = 4;
constexpr int NMS_CAP = 16384;  // candidates per image the LDS sort holds
constexpr int NMS_THREADS = 1024;

struct Levels {
  int n;
  int a_off[MAXLV + 1];
  int w[MAXLV];
  float stride[MAXLV];
};

// One thread per anchor walking its own row of 64 + nc floats touches 64 cache lines per load instruction (215 us for 32 x 8400 x 144:
// a tenth of what the bytes cost).  A workgroup takes DEC_AW consecutive anchors instead: their rows are one contiguous run of `pred`
// (16-byte loads, lane = consecutive address) parked in LDS at an odd pitch; then thread = (anchor, box side) does the DFL softmax of
// its 16 bins out of LDS (lane = anchor: conflict-free), 64 threads assemble the boxes, and the class scores leave with the anchor as
// the fast index -- `y` is attribute-major, so a wave writes 256 contiguous bytes per class.  Same arithmetic, in the same order, as
// the per-row kernel below (kept for class counts whose tile would not fit 64 KiB of LDS): the outputs are bit-identical.
constexpr int DEC_AW = 64;

__global__ __launch_bounds__(256) void decode_kernel(const float* pred, int B, int A, int ld, int nc, Levels L, float* y) {
  extern __shared__ __attribute__((aligned(16))) float dtile[];  // [DEC_AW][TS] | d[4][DEC_AW]
  const int ncol = 4 * REG + nc, TS = ncol | 1;
  float* sd = dtile + DEC_AW * TS;
  const long long total = (long long)B * A;
  const long long i0 = (long long)blockIdx.x * DEC_AW;
  const int rows = (int)(total - i0 < DEC_AW ? total - i0 : DEC_AW);
  const int tid = threadIdx.x;
  const float* src = pred + i0 * ld;
  if (((ncol | ld) & 3) == 0 && (reinterpret_cast<uintptr_t>(pred) & 15) == 0) {
    const int q4 = ncol >> 2;
    for (int e = tid; e < rows * q4; e += 256) {
      const int r = e / q4, c4 = e - r * q4;
      const float4 v = *reinterpret_cast<const float4*>(src + (long long)r * ld + 4 * c4);
      float* t = dtile + r * TS + 4 * c4;
      t[0] = v.x;
      t[1] = v.y;
      t[2] = v.z;
      t[3] = v.w;
    }
  } else {
    for (int e = tid; e < rows * ncol; e += 256) {
      const int r = e / ncol, c = e - r * ncol;
      dtile[r * TS + c] = src[(long long)r * ld + c];
    }
  }
  __syncthreads();
  const int al = tid & (DEC_AW - 1), grp = tid >> 6;  // anchor of the tile, wave
  {
    const float* t = dtile + al * TS + grp * REG;  // wave = box side
    float v[REG], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < REG; ++k) {
      v[k] = t[k];
      mx = fmaxf(mx, v[k]);
    }
    float s = 0.f, e = 0.f;
#pragma unroll
    for (int k = 0; k < REG; ++k) {
      float q = __expf(v[k] - mx);
      s += q;
      e += q * k;
    }
    sd[grp * DEC_AW + al] = e / s;
  }
  __syncthreads();
  if (al >= rows) return;
  const long long i = i0 + al;
  const int b = (int)(i / A), a = (int)(i - (long long)b * A);
  float* o = y + (long long)b * (4 + nc) * A + a;
  if (grp == 0) {
    int lv = 0;
    for (int k = 1; k < L.n; ++k)
      if (a >= L.a_off[k]) lv = k;
    int r = a - L.a_off[lv];
    int yy = r / L.w[lv], xx = r - yy * L.w[lv];
    float ax = xx + 0.5f, ay = yy + 0.5f, st = L.stride[lv];
    float x1 = ax - sd[al], y1 = ay - sd[DEC_AW + al], x2 = ax + sd[2 * DEC_AW + al], y2 = ay + sd[3 * DEC_AW + al];
    o[0] = (x1 + x2) * 0.5f * st;
    o[(long long)A] = (y1 + y2) * 0.5f * st;
    o[2LL * A] = (x2 - x1) * st;
    o[3LL * A] = (y2 - y1) * st;
  }
  const float* t = dtile + al * TS + 4 * REG;
  for (int c = grp; c < nc; c += 4) o[(long long)(4 + c) * A] = cvx_sigmoid(t[c]);
}

__global__ void decode_rows_kernel(const float* pred, int B, int A, int no, int nc, Levels L, float* y) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // b*A + a
  if (i >= (long long)B * A) return;
  int b = (int)(i / A), a = (int)(i - (long long)b * A);
  const float* p = pred + i * no;
  float d[4];
#pragma unroll
  for (int side = 0; side < 4; ++side) {
    float v[REG], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < REG; ++k) {
      v[k] = p[side * REG + k];
      mx = fmaxf(mx, v[k]);
    }
    float s = 0.f, e = 0.f;
#pragma unroll
    for (int k = 0; k < REG; ++k) {
      float q = __expf(v[k] - mx);
      s += q;
      e += q * k;
    }
    d[side] = e / s;
  }
  int lv = 0;
  for (int k = 1; k < L.n; ++k)
    if (a >= L.a_off[k]) lv = k;
  int r = a - L.a_off[lv];
  int yy = r / L.w[lv], xx = r - yy * L.w[lv];
  float ax = xx + 0.5f, ay = yy + 0.5f, st = L.stride[lv];
  float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
  float* o = y + (long long)b * (4 + nc) * A + a;
  o[0] = (x1 + x2) * 0.5f * st;
  o[(long long)A] = (y1 + y2) * 0.5f * st;
  o[2LL * A] = (x2 - x1) * st;
  o[3LL * A] = (y2 - y1) * st;
  for (int c = 0; c < nc; ++c) o[(long long)(4 + c) * A] = cvx_sigmoid(p[4 * REG + c]);
}

// fp32 IoU exactly as torchvision's nms_kernel / oracle.nms_ref.greedy_nms_per_class
__device__ __forceinline__ bool iou_gt(const float4& a, float area_a, const float4& b, float area_b, float thr) {
  float w = fmaxf(0.f, __fsub_rn(fminf(a.z, b.z), fmaxf(a.x, b.x)));
  float h = fmaxf(0.f, __fsub_rn(fminf(a.w, b.w), fmaxf(a.y, b.y)));
  float inter = __fmul_rn(w, h);
  float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_a, area_b), inter));
  return ovr > thr;
}

struct NmsWs {
  float4* box;              // [B][cap] sorted boxes (xyxy)
  float4* boxs;             // [B][cap] the boxes the suppression test sees (class-shifted in OFFSET mode)
  float* score;             // [B][cap]
  int* cls;                 // [B][cap]
  int* aidx;                // [B][cap]
  unsigned long long* mat;  // [B][cap][cap/64]
};

__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(const float* y, int A, int nc, float conf_thres, float iou_thres, int max_det, int cap,
                                                          int variant_flags, NmsWs ws, float* out_rows, int* out_index, int* counts) {
  const int variant = variant_flags & 0xff;
  const bool xyxy = (variant_flags & CVX_NMS_BOXES_XYXY) != 0;
  extern __shared__ unsigned long long keys[];  // cap2 entries (power of two >= candidates)
  __shared__ int s_n;
  __shared__ unsigned long long s_removed[NMS_CAP / 64];
  __shared__ int s_nkeep;
  __shared__ float s_wmax[NMS_THREADS / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* yb = y + (long long)b * (4 + nc) * A;
  if (tid == 0) {
    s_n = 0;
    s_nkeep = 0;
  }
  __syncthreads();
  // ---- 1. candidates: best class score > conf (ultralytics_ops.py:190,225-226) ----
  for (int a0 = 0; a0 < A; a0 += NMS_THREADS) {
    int a = a0 + tid;
    float best = -1.f;
    if (a < A) {
      for (int c = 0; c < nc; ++c) best = fmaxf(best, yb[(long long)(4 + c) * A + a]);
    }
    if (a < A && best > conf_thres) {
      int slot = atomicAdd(&s_n, 1);
      if (slot < cap) keys[slot] = ((unsigned long long)(0xFFFFFFFFu - __float_as_uint(best)) << 32) | (unsigned)a;
    }
  }
  __syncthreads();
  if (s_n > cap) {  // more candidates than the in-LDS sort holds (only possible beyond 16384 anchors): signalled, never truncated silently
    if (tid == 0) counts[b] = -1;
    return;
  }
  const int n = min(s_n, cap);
  int cap2 = 1;
  while (cap2 < n) cap2 <<= 1;
  for (int i = n + tid; i < cap2; i += NMS_THREADS) keys[i] = ~0ull;
  __syncthreads();
  // ---- 2. bitonic sort ascending on (inverted score, anchor) = score desc, anchor asc ----
  for (int k = 2; k <= cap2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < cap2; i += NMS_THREADS) {
        int l = i ^ j;
        if (l > i) {
          unsigned long long x = keys[i], z = keys[l];
          bool up = (i & k) == 0;
          if ((x > z) == up) {
            keys[i] = z;
            keys[l] = x;
          }
        }
      }
      __syncthreads();
    }
  }
  // ---- 3. gather sorted candidates ----
  // torchvision 0.14.1 batched_nms: vanilla when boxes.numel() = 4 n exceeds 20000 (CUDA tensor) / 4000 (CPU tensor)
  const bool offset_mode = variant == CVX_NMS_OFFSET || (variant == CVX_NMS_TV0141_CUDA && 4 * n <= 20000) ||
                           (variant == CVX_NMS_TV0141_CPU && 4 * n <= 4000);
  float4* box = ws.box + (long long)b * cap;
  float4* boxs = ws.boxs + (long long)b * cap;
  float* score = ws.score + (long long)b * cap;
  int* cls = ws.cls + (long long)b * cap;
  int* aidx = ws.aidx + (long long)b * cap;
  for (int i = tid; i < n; i += NMS_THREADS) {
    int a = (int)(keys[i] & 0xFFFFFFFFu);
    float cx = yb[a], cy = yb[(long long)A + a], w = yb[2LL * A + a], h = yb[3LL * A + a];
    if (xyxy) {  // CVX_NMS_BOXES_XYXY: rows 0..3 already hold corners (SSD's clipped decode): taken as they are
      box[i] = make_float4(cx, cy, w, h);
    } else {
      float hw = __fmul_rn(w, 0.5f), hh = __fmul_rn(h, 0.5f);
      box[i] = make_float4(__fsub_rn(cx, hw), __fsub_rn(cy, hh), __fadd_rn(cx, hw), __fadd_rn(cy, hh));
    }
    float best = -1.f;
    int bc = 0;
    for (int c = 0; c < nc; ++c) {
      float s = yb[(long long)(4 + c) * A + a];
      if (s > best) {
        best = s;
        bc = c;
      }
    }
    score[i] = best;
    cls[i] = bc;
    aidx[i] = a;
  }
  __syncthreads();
  // ---- 3b. the boxes the suppression test sees: shifted by cls * (boxes.max() + 1), fp32, in OFFSET mode ----
  {
    float m = -INFINITY;
    if (offset_mode)
      for (int i = tid; i < n; i += NMS_THREADS) {
        const float4 bx = box[i];
        m = fmaxf(fmaxf(m, fmaxf(bx.x, bx.y)), fmaxf(bx.z, bx.w));
      }
    m = cvx_wave_max64(m);
    if ((tid & 63) == 0) s_wmax[tid >> 6] = m;
    __syncthreads();
    float mx = s_wmax[0];
    for (int w = 1; w < NMS_THREADS / 64; ++w) mx = fmaxf(mx, s_wmax[w]);
    const float unit = __fadd_rn(mx, 1.0f);
    for (int i = tid; i < n; i += NMS_THREADS) {
      float4 bx = box[i];
      if (offset_mode) {
        const float off = __fmul_rn((float)cls[i], unit);
        bx = make_float4(__fadd_rn(bx.x, off), __fadd_rn(bx.y, off), __fadd_rn(bx.z, off), __fadd_rn(bx.w, off));
      }
      boxs[i] = bx;
    }
  }
  __syncthreads();
  // ---- 4. suppression matrix: bit j of row i set when j > i, same class, IoU > thr ----
  const int nw = (n + 63) / 64;
  unsigned long long* mat = ws.mat + (long long)b * cap * (cap / 64);
  for (long long t = tid; t < (long long)n * nw; t += NMS_THREADS) {
    int i = (int)(t / nw), wj = (int)(t - (long long)i * nw);
    unsigned long long bits = 0;
    if (wj * 64 + 63 > i) {
      float4 bi = boxs[i];
      float ai = __fmul_rn(__fsub_rn(bi.z, bi.x), __fsub_rn(bi.w, bi.y));
      int ci = cls[i];
      int j0 = max(wj * 64, i + 1), j1 = min(n, wj * 64 + 64);
      for (int j = j0; j < j1; ++j) {
        if (!offset_mode && cls[j] != ci) continue;
        float4 bj = boxs[j];
        float aj = __fmul_rn(__fsub_rn(bj.z, bj.x), __fsub_rn(bj.w, bj.y));
        if (iou_gt(bi, ai, bj, aj, iou_thres)) bits |= 1ull << (j - wj * 64);
      }
    }
    mat[(long long)i * nw + wj] = bits;
  }
  for (int i = tid; i < nw; i += NMS_THREADS) s_removed[i] = 0;
  __threadfence_block();
  __syncthreads();
  // ---- 5. greedy walk by wave 0 ----
  if (tid < 64) {
    int nkeep = 0;
    for (int i = 0; i < n && nkeep < max_det; ++i) {
      bool dead = (s_removed[i >> 6] >> (i & 63)) & 1ull;  // uniform
      if (dead) continue;
      if (tid == 0) out_index[(long long)b * max_det + nkeep] = i;  // the keep list lives in the output's own index column until step 6
      ++nkeep;
      for (int wj = tid; wj < nw; wj += 64) s_removed[wj] |= mat[(long long)i * nw + wj];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (tid == 0) s_nkeep = nkeep;
  }
  __syncthreads();
  // ---- 6. emit ----
  const int nk = s_nkeep;
  for (int r = tid; r < nk; r += NMS_THREADS) {
    int i = out_index[(long long)b * max_det + r];  // (each thread reads and rewrites only its own slots)
    float4 bx = box[i];
    float* o = out_rows + ((long long)b * max_det + r) * 6;
    o[0] = bx.x;
    o[1] = bx.y;
    o[2] = bx.z;
    o[3] = bx.w;
    o[4] = score[i];
    o[5] = (float)cls[i];
    out_index[(long long)b * max_det + r] = aidx[i];
  }
  if (tid == 0) counts[b] = nk;
}

int cap_for(int A) {
  int c = 64;
  while (c < A && c < NMS_CAP) c <<= 1;
  return c;
}
long long al(long long x) { return (x + 255) & ~255LL; }

}  // namespace

extern "C" int cvx_decode(const float* pred, int32_t B, int32_t A, int32_t nc, const int32_t* level_hw, const float* strides, int32_t n_levels,
                          float* y, void* hip_stream) {
  return cvx_decode_strided(pred, nc + 4 * REG, B, A, nc, level_hw, strides, n_levels, y, hip_stream);
}

extern "C" int cvx_decode_strided(const float* pred, int32_t pred_ld, int32_t B, int32_t A, int32_t nc, const int32_t* level_hw,
                                  const float* strides, int32_t n_levels, float* y, void* hip_stream) {
  CVX_CHECK(pred && y && level_hw && strides && n_levels >= 1 && n_levels <= MAXLV, "bad arguments");
  CVX_CHECK(pred_ld >= nc + 4 * REG, "pred_ld must cover 64 + nc values");
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
  CVX_CHECK(off == A, "level sizes do not add up to the anchor count");
  const int TS = (4 * REG + nc) | 1;
  const size_t lds = (size_t)(DEC_AW * TS + 4 * DEC_AW) * sizeof(float);
  if (lds <= 64 * 1024) {
    hipLaunchKernelGGL(decode_kernel, dim3(cvx_cdiv((long long)B * A, DEC_AW)), dim3(256), lds, (hipStream_t)hip_stream, pred, B, A, pred_ld, nc,
                       L, y);
  } else {
    hipLaunchKernelGGL(decode_rows_kernel, dim3(cvx_cdiv((long long)B * A, 256)), dim3(256), 0, (hipStream_t)hip_stream, pred, B, A, pred_ld,
                       nc, L, y);
  }
  CVX_HIP(hipGetLastError());
  return 0;
}

extern "C" int64_t cvx_nms_workspace_bytes(int32_t B, int32_t A) {
  long long cap = cap_for(A);
  return 2 * al(B * cap * 16) + 3 * al(B * cap * 4) + al((long long)B * cap * (cap / 64) * 8) + 1024;
}

extern "C" int cvx_nms(const float* y, int32_t B, int32_t A, int32_t nc, float conf_thres, float iou_thres, int32_t max_det, float* out_rows,
                       int32_t* out_index, int32_t* counts, void* workspace, int64_t workspace_bytes, void* hip_stream) {
  return cvx_nms_variant(y, B, A, nc, conf_thres, iou_thres, max_det, CVX_NMS_TV0141_CUDA, out_rows, out_index, counts, workspace,
                         workspace_bytes, hip_stream);
}

extern "C" int cvx_nms_variant(const float* y, int32_t B, int32_t A, int32_t nc, float conf_thres, float iou_thres, int32_t max_det,
                               int32_t variant, float* out_rows, int32_t* out_index, int32_t* counts, void* workspace,
                               int64_t workspace_bytes, void* hip_stream) {
  CVX_CHECK(y && out_rows && out_index && counts && workspace, "null arguments");
  CVX_CHECK((variant & 0xff) >= CVX_NMS_TV0141_CUDA && (variant & 0xff) <= CVX_NMS_VANILLA && (variant & ~(0xff | CVX_NMS_BOXES_XYXY)) == 0,
            "unknown batched_nms variant");
  CVX_CHECK(conf_thres >= 0.f && conf_thres <= 1.f && iou_thres >= 0.f && iou_thres <= 1.f, "thresholds must lie in [0,1]");
  CVX_CHECK(max_det >= 1 && max_det <= NMS_CAP, "max_det must lie in [1,16384]");
  CVX_CHECK(workspace_bytes >= cvx_nms_workspace_bytes(B, A), "workspace too small");
  const long long cap = cap_for(A);
  char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  NmsWs ws;
  long long off = 0;
  ws.box = (float4*)(base + off);
  off += al(B * cap * 16);
  ws.boxs = (float4*)(base + off);
  off += al(B * cap * 16);
  ws.score = (float*)(base + off);
  off += al(B * cap * 4);
  ws.cls = (int*)(base + off);
  off += al(B * cap * 4);
  ws.aidx = (int*)(base + off);
  off += al(B * cap * 4);
  ws.mat = (unsigned long long*)(base + off);
  static bool attr_set = false;
  if (!attr_set) {
    CVX_HIP(hipFuncSetAttribute((const void*)nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NMS_CAP * 8));
    attr_set = true;
  }
  hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(NMS_THREADS), (size_t)cap * 8, (hipStream_t)hip_stream, y, A, nc, conf_thres, iou_thres, max_det,
                     (int)cap, (int)variant, ws, out_rows, out_index, counts);
  CVX_HIP(hipGetLastError());
  return 0;
}

// ---- YOLOv7 anchor decode (core/algorithms/yolo_v7.py:246-343) ------------------------------------------------------------
// pred: fp32 rows (B, sum_l H_l*W_l, ld), a row = one pixel of one level (levels in the order of the network's outputs,
// coarsest first), columns a*(5+nc)+k for anchor a = 0..2.  dec: (B, 3*sum_l H_l*W_l, 5+nc) in the reference's order -- level,
// then anchor, then pixel -- with normalised (cx, cy, w, h), objectness and class probabilities.  y (optional): the same
// candidates as (B, 4+nc, 3*sum) channel-major [cx, cy, w, h, obj*cls_k] -- the input format of cvx_nms_variant.
namespace {
struct Y7Levels {
  int n;
  int row0[5];   // first pred row of level l (row0[n] = total rows)
  int w[4], h[4];
  float aw[4][3], ah[4][3];  // anchors scaled to the level's grid (anchor / stride)
};
// A workgroup takes 32 consecutive anchors of the reference's order.  Every anchor's 5 + nc attributes are one contiguous run both in the
// head rows and in `dec`, so a wave moves one anchor at a time (lane = attribute: coalesced in and out) through an LDS tile; `y` is
// attribute-major (contiguous along the anchors), so it is written from the tile with the anchor as the fast index.  One thread per anchor
// walking its own row touched 64 cache lines per load / store instruction.
struct Y7Where {
  int b, l, a, gx, gy;
  long long r;
  const float* p;
};
__device__ __forceinline__ Y7Where y7_locate(long long i, const float* pred, int ld, int attrs, const Y7Levels& L) {
  Y7Where w;
  const long long R = L.row0[L.n];
  w.b = (int)(i / (3 * R));
  w.r = i - (long long)w.b * 3 * R;  // reference anchor index inside the image
  int l = 0;
  while (l + 1 < L.n && w.r >= 3LL * L.row0[l + 1]) ++l;
  const int hw = L.row0[l + 1] - L.row0[l];
  const long long rl = w.r - 3LL * L.row0[l];
  w.l = l;
  w.a = (int)(rl / hw);
  const int pix = (int)(rl - (long long)w.a * hw);
  w.gy = pix / L.w[l];
  w.gx = pix - w.gy * L.w[l];
  w.p = pred + ((long long)w.b * R + L.row0[l] + pix) * ld + w.a * attrs;
  return w;
}
__global__ __launch_bounds__(256) void yolo7_decode_kernel(const float* pred, int ld, int B, int nc, Y7Levels L, float* dec, float* y) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [32][TS]
  const int attrs = 5 + nc, TS = attrs | 1;
  const long long R = L.row0[L.n], total = (long long)B * 3 * R;
  const long long i0 = (long long)blockIdx.x * 32;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  auto sg = [](float v) { return 1.f / (1.f + expf(-v)); };
  for (int q = 0; q < 8; ++q) {
    const int al = wave * 8 + q;
    if (i0 + al >= total) break;
    const Y7Where w = y7_locate(i0 + al, pred, ld, attrs, L);
    for (int k = lane; k < attrs; k += 64) tile[al * TS + k] = sg(w.p[k]);
  }
  __syncthreads();
  if (threadIdx.x < 32 && i0 + threadIdx.x < total) {
    const Y7Where w = y7_locate(i0 + threadIdx.x, pred, ld, attrs, L);
    float* t = tile + threadIdx.x * TS;
    const float sx = t[0], sy = t[1], sw = t[2], sh = t[3];
    const int l = w.l;
    const float cx = (sx * 2.f - 0.5f + (float)w.gx) / (float)L.w[l], cy = (sy * 2.f - 0.5f + (float)w.gy) / (float)L.h[l];
    const float tw = sw * 2.f, th = sh * 2.f;
    t[0] = cx;
    t[1] = cy;
    t[2] = tw * tw * L.aw[l][w.a] / (float)L.w[l];
    t[3] = th * th * L.ah[l][w.a] / (float)L.h[l];
  }
  __syncthreads();
  for (int q = 0; q < 8; ++q) {
    const int al = wave * 8 + q;
    if (i0 + al >= total) break;
    float* d = dec + (i0 + al) * attrs;
    for (int k = lane; k < attrs; k += 64) d[k] = tile[al * TS + k];
  }
  if (y) {
    const int al = threadIdx.x & 31, kk = threadIdx.x >> 5;
    const long long i = i0 + al;
    if (i < total) {
      const long long A3 = 3 * R;
      const int b = (int)(i / A3);
      float* yb = y + (long long)b * (4 + nc) * A3 + (i - (long long)b * A3);
      const float* t = tile + al * TS;
      const float obj = t[4];
      for (int k = kk; k < 4 + nc; k += 8) yb[(long long)k * A3] = k < 4 ? t[k] : obj * t[k + 1];
    }
  }
}
}  // namespace

extern "C" int cvx_yolo7_decode(const float* pred, int32_t pred_ld, int32_t B, int32_t nc, const int32_t* level_hw, const float* anchors_wh,
                                int32_t n_levels, int32_t input_h, int32_t input_w, float* dec, float* y, void* hip_stream) {
  CVX_CHECK(pred && level_hw && anchors_wh && dec && n_levels >= 1 && n_levels <= 4 && B > 0 && nc > 0, "bad arguments");
  CVX_CHECK(pred_ld >= 3 * (5 + nc), "pred_ld must cover 3 * (5 + nc) columns");
  Y7Levels L;
  memset(&L, 0, sizeof(L));
  L.n = n_levels;
  int off = 0;
  for (int l = 0; l < n_levels; ++l) {
    L.row0[l] = off;
    L.h[l] = level_hw[2 * l];
    L.w[l] = level_hw[2 * l + 1];
    off += L.h[l] * L.w[l];
    const float stride_h = (float)input_h / (float)L.h[l], stride_w = (float)input_w / (float)L.w[l];  // yolo_v7.py:259-260
    for (int a = 0; a < 3; ++a) {
      L.aw[l][a] = anchors_wh[(l * 3 + a) * 2] / stride_w;
      L.ah[l][a] = anchors_wh[(l * 3 + a) * 2 + 1] / stride_h;
    }
  }
  L.row0[n_levels] = off;
  const long long n = (long long)B * 3 * off;
  CVX_CHECK(nc <= 400 && (n + 31) / 32 < (1LL << 31), "too many classes / anchors");
  hipLaunchKernelGGL(yolo7_decode_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), (size_t)32 * ((5 + nc) | 1) * 4, (hipStream_t)hip_stream, pred, pred_ld,
                     B, nc, L, dec, y);
  CVX_HIP(hipGetLastError());
  return 0;
}

// ---- SSD decode (core/algorithms/ssd.py:236-325): softmax over the class scores, prior-box regression decode with
// variances (0.1, 0.2), corners clipped to [0, 1].  One thread per prior. ----
namespace {
// STAGED: the workgroup's 256 score rows (one contiguous run of 256 * nc1 floats) pass through LDS -- coalesced in, softmax in place on the
// thread's own row (odd stride: conflict-free), coalesced out; straight from memory a thread's row lies 4 * nc1 bytes from its neighbour's
// and every load / store instruction touches 64 cache lines.
template <bool STAGED>
__global__ __launch_bounds__(256) void ssd_decode_kernel(const float* loc, const float* conf, const float* priors, int B, int A, int nc1, float v0,
                                                         float v1, float* boxes, float* prob, unsigned* class_max) {
  extern __shared__ __attribute__((aligned(16))) float srow[];
  const long long N = (long long)B * A;
  const long long i0 = (long long)blockIdx.x * 256;
  const long long i = i0 + threadIdx.x;
  const int nvalid = (int)(N - i0 < 256 ? N - i0 : 256);
  if (STAGED) {
    const float* c0 = conf + i0 * nc1;
    for (int j = threadIdx.x; j < nvalid * nc1; j += 256) srow[j] = c0[j];
    __syncthreads();
  }
  if (i < N) {
    const int a = (int)(i % A);
    const float* l = loc + i * 4;
    const float4 pr = *reinterpret_cast<const float4*>(priors + (long long)a * 4);
    const float aw = pr.z - pr.x, ah = pr.w - pr.y;
    const float acx = 0.5f * (pr.z + pr.x), acy = 0.5f * (pr.w + pr.y);
    const float cx = l[0] * aw * v0 + acx, cy = l[1] * ah * v0 + acy;
    const float w = expf(l[2] * v1) * aw, h = expf(l[3] * v1) * ah;
    float4 o = make_float4(cx - 0.5f * w, cy - 0.5f * h, cx + 0.5f * w, cy + 0.5f * h);
    o.x = fminf(fmaxf(o.x, 0.f), 1.f);
    o.y = fminf(fmaxf(o.y, 0.f), 1.f);
    o.z = fminf(fmaxf(o.z, 0.f), 1.f);
    o.w = fminf(fmaxf(o.w, 0.f), 1.f);
    *reinterpret_cast<float4*>(boxes + i * 4) = o;
    const float* c = STAGED ? srow + threadIdx.x * nc1 : conf + i * nc1;
    float m = c[0];
    for (int k = 1; k < nc1; ++k) m = fmaxf(m, c[k]);
    float sum = 0.f;
    for (int k = 0; k < nc1; ++k) sum += expf(c[k] - m);
    float* p = STAGED ? srow + threadIdx.x * nc1 : prob + i * nc1;
    for (int k = 0; k < nc1; ++k) p[k] = expf(c[k] - m) / sum;
  }
  if (STAGED) {
    __syncthreads();
    float* p0 = prob + i0 * nc1;
    for (int j = threadIdx.x; j < nvalid * nc1; j += 256) p0[j] = srow[j];
    if (class_max)  // largest probability per class over the batch (probabilities are >= 0: their bit patterns order like the values)
      for (int k = threadIdx.x; k < nc1; k += 256) {
        float m = 0.f;
        for (int r = 0; r < nvalid; ++r) m = fmaxf(m, srow[r * nc1 + k]);
        if (__float_as_uint(m) > class_max[k]) atomicMax(&class_max[k], __float_as_uint(m));
      }
  } else if (class_max) {
    const float* p = prob + i * nc1;
    for (int k = 0; k < nc1; ++k) {
      float m = i < N ? p[k] : 0.f;
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > class_max[k]) atomicMax(&class_max[k], __float_as_uint(m));
    }
  }
}
}  // namespace

extern "C" int cvx_ssd_decode_max(const float* loc, const float* conf, const float* priors, int32_t B, int32_t A, int32_t num_classes_plus_bg,
                                  float variance_xy, float variance_wh, float* boxes, float* prob, float* class_max, void* hip_stream) {
  CVX_CHECK(loc && conf && priors && boxes && prob && B > 0 && A > 0 && num_classes_plus_bg > 1, "bad arguments");
  const long long n = (long long)B * A;
  hipStream_t st = (hipStream_t)hip_stream;
  if (class_max) CVX_HIP(hipMemsetAsync(class_max, 0, (size_t)num_classes_plus_bg * 4, st));
  const size_t stage = (size_t)256 * num_classes_plus_bg * 4;
  if (stage <= 64 * 1024)
    hipLaunchKernelGGL(ssd_decode_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), stage, st, loc, conf, priors, B, A, num_classes_plus_bg,
                       variance_xy, variance_wh, boxes, prob, (unsigned*)class_max);
  else
    hipLaunchKernelGGL(ssd_decode_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, loc, conf, priors, B, A, num_classes_plus_bg,
                       variance_xy, variance_wh, boxes, prob, (unsigned*)class_max);
  CVX_HIP(hipGetLastError());
  return 0;
}
extern "C" int cvx_ssd_decode(const float* loc, const float* conf, const float* priors, int32_t B, int32_t A, int32_t num_classes_plus_bg,
                              float variance_xy, float variance_wh, float* boxes, float* prob, void* hip_stream) {
  return cvx_ssd_decode_max(loc, conf, priors, B, A, num_classes_plus_bg, variance_xy, variance_wh, boxes, prob, nullptr, hip_stream);
}
