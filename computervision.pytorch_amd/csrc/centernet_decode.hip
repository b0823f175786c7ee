// CenterNet heat-map decode on gfx950: sigmoid -> the reference's 3x3 max-pool over (x, class) -> global top-K ->
// gather + box arithmetic -> score mask -> class-agnostic DIoU-NMS.
//
// Reference (file:line under the reference tree): CenterNetA.decode_boxes / _suppress_redundant_centers / _top_k,
// core/algorithms/centernet.py:271-338; diou_nms, core/utils/nms.py:9-31; box_iou / box_diou, core/utils/iou.py:8-64.
// Kept quirks: the reference max-pools the NHWC tensor as if it were NCHW, so the 3x3 window spans (x-1..x+1, c-1..c+1)
// inside one row y (centernet.py:279-280,324); it reads the centre offsets from the "wh" head's two channels and the sizes
// from the "reg" head's (:276-277); the letterbox inverse stays on the host (a handful of scalars per image).
// Tie order of the top-K (torch.topk leaves it undefined): score descending, flat (y, x, c) index ascending, as
// oracle/centernet_ref.py defines it.  All box arithmetic in fp32 with explicit round-to-nearest intrinsics (no FMA
// contraction): identical operation order to the reference's torch ops.
//
//   K1 peak_scores   one thread per (b, y, x, c): window max on the LOGITS (sigmoid is monotone: the arg-max is the same,
//                    and `heat == hmax` is evaluated on the two sigmoid values exactly as the reference does) -> fp32 scores
//   K2 select        one 1024-thread workgroup per image: 3-pass radix select of the K-th largest score (LDS histograms),
//                    collection of the candidates, in-LDS bitonic sort, boxes, mask, greedy DIoU-NMS by one wave
#include <cstring>

#include "../../include/cvx_engine.h"
#include "cvx_common.h"

namespace {

constexpr int SEL_THREADS = 1024;
constexpr int CAND_CAP = 2048;  // top-K candidates incl. score ties the LDS sort holds
constexpr int MAX_K = 256;

__device__ __forceinline__ float sigmoid_ref(float x) {  // 1 / (1 + exp(-x)) in fp32, the reference's torch.sigmoid arithmetic
  return __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-x)));
}

__global__ __launch_bounds__(256) void peak_scores_kernel(const float* pred, int ld, int B, int H, int W, int C, float* scores) {
  const long long n = (long long)B * H * W * C;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int c, x;
  long long pix;  // b*H*W + y*W + x
  if (n <= 0x7fffffffLL) {  // 32-bit index arithmetic: three 64-bit divisions by run-time values cost more than the rest of the thread
    const unsigned iu = (unsigned)i, pu = iu / (unsigned)C;
    c = (int)(iu - pu * (unsigned)C);
    x = (int)(pu % (unsigned)W);
    pix = pu;
  } else {
    c = (int)(i % C);
    pix = i / C;
    x = (int)(pix % W);
  }
  const float* row = pred + pix * ld;
  // the 3 x 3 window over (x, class) with its offsets clamped into the map: a clamped offset lands on an element the window holds anyway,
  // so the maximum is unchanged, and the nine loads are unconditional -- issued together instead of one memory round trip per branch
  const int dxm = x > 0 ? -1 : 0, dxp = x < W - 1 ? 1 : 0, cm = c > 0 ? c - 1 : c, cp = c < C - 1 ? c + 1 : c;
  const float* rm = row + (long long)dxm * ld;
  const float* rp = row + (long long)dxp * ld;
  const float v = row[c];
  const float a0 = rm[cm], a1 = rm[c], a2 = rm[cp], b0 = row[cm], b2 = row[cp], c0 = rp[cm], c1 = rp[c], c2 = rp[cp];
  const float m = fmaxf(fmaxf(fmaxf(fmaxf(a0, a1), fmaxf(a2, b0)), fmaxf(fmaxf(b2, c0), fmaxf(c1, c2))), v);
  const float s = sigmoid_ref(v);
  scores[i] = (m == v || sigmoid_ref(m) == s) ? s : 0.f;
}

struct DecodeOut {
  float* boxes;  // [B][K][4] xyxy in [0, 1]
  float* scores;
  int* classes;
  int* topk_index;  // [B][K] flat (y*W + x)*C + c of every top-K entry (before mask / NMS)
  int* keep;        // [B][K] positions (0..K-1) in the top-K list of the survivors, descending score
  int* counts;      // [B] survivors; -1: more than CAND_CAP candidates tie at the K-th score
};

__device__ __forceinline__ float diou(const float4& a, const float4& b) {  // core/utils/iou.py:8-64
  const float eps = 1e-6f;
  const float a1 = __fmul_rn(__fsub_rn(a.z, a.x), __fsub_rn(a.w, a.y));
  const float a2 = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
  const float iw = fmaxf(__fsub_rn(fminf(a.z, b.z), fmaxf(a.x, b.x)), 0.f);
  const float ih = fmaxf(__fsub_rn(fminf(a.w, b.w), fmaxf(a.y, b.y)), 0.f);
  const float inter = __fmul_rn(iw, ih);
  const float iou = __fdiv_rn(inter, fmaxf(__fsub_rn(__fadd_rn(a1, a2), inter), eps));
  const float cx1 = __fdiv_rn(__fadd_rn(a.x, a.z), 2.f), cy1 = __fdiv_rn(__fadd_rn(a.y, a.w), 2.f);
  const float cx2 = __fdiv_rn(__fadd_rn(b.x, b.z), 2.f), cy2 = __fdiv_rn(__fadd_rn(b.y, b.w), 2.f);
  const float ew = fmaxf(__fsub_rn(fmaxf(a.z, b.z), fminf(a.x, b.x)), 0.f);
  const float eh = fmaxf(__fsub_rn(fmaxf(a.w, b.w), fminf(a.y, b.y)), 0.f);
  const float c_sq = __fadd_rn(__fmul_rn(ew, ew), __fmul_rn(eh, eh));
  const float dx = __fsub_rn(cx1, cx2), dy = __fsub_rn(cy1, cy2);
  const float d_sq = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
  return __fsub_rn(iou, __fdiv_rn(d_sq, fmaxf(c_sq, eps)));
}

// Sorted top-K of n non-negative floats (their bit patterns, `key`) by (value descending, index ascending); zeros never
// qualify.  3-pass radix select of the K-th largest value (LDS histograms), collection of every element >= it, in-LDS
// bitonic sort of 64-bit (inverted value, idx0 + i) keys.  Returns the number of sorted candidates left in cand[] (the
// first min(K, that) are the answer), or -1 when more than CAND_CAP elements tie at the K-th value.  All threads call it.
struct SelShared {
  unsigned hist[2048];
  unsigned long long cand[CAND_CAP];
  unsigned prefix, remaining, ncand;
};

// Histogram increment for a whole wave (every lane calls it; `valid` says whether the lane has a key).  Heat maps are concentrated --
// an untrained head puts every score near 0.5, a trained one most of them near 0 -- so plain LDS atomics pile 64 lanes on one bin and
// serialise; two rounds of "all lanes that share the first lane's bin add once", then the stragglers one by one.
__device__ __forceinline__ void wave_hist_add(unsigned* hist, unsigned bin, bool valid) {
  unsigned long long todo = __ballot(valid);
  const int lane = threadIdx.x & 63;
  for (int round = 0; round < 2 && todo; ++round) {
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned b0 = __shfl(bin, leader);
    const bool mine = valid && bin == b0;
    const unsigned long long same = __ballot(mine);
    if (lane == leader) atomicAdd(&hist[b0], (unsigned)__popcll(same));
    todo &= ~same;
    if (mine) valid = false;
  }
  if (valid) atomicAdd(&hist[bin], 1u);
}

__device__ int topk_sorted(const unsigned* key, long long n, unsigned idx0, int K, SelShared& sh) {
  const int tid = threadIdx.x;
  if (tid == 0) {
    sh.prefix = 0;
    sh.remaining = (unsigned)K;
  }
  const bool vec4 = (reinterpret_cast<uintptr_t>(key) & 15) == 0;
  const int shifts[3] = {21, 10, 0};
  const int widths[3] = {11, 11, 10};
  unsigned known_mask = 0;
  for (int pass = 0; pass < 3; ++pass) {
    for (int i = tid; i < 2048; i += SEL_THREADS) sh.hist[i] = 0;
    __syncthreads();
    const unsigned prefix = sh.prefix;
    const int sft = shifts[pass];
    const unsigned mask = (1u << widths[pass]) - 1;
    // zeros (suppressed non-peaks: 8 of 9 elements) never qualify and the walk below never reads the bin they would fall into.
    // Four keys per lane and trip (one 16-byte load when the slice allows it): a single 4-byte load per trip leaves the pass waiting
    // for memory 80 times over.  Uniform trip count: the wave-level histogram add needs every lane.
    for (long long i0 = 0; i0 < n; i0 += 4 * SEL_THREADS) {
      const long long i = i0 + 4 * tid;
      unsigned k4[4] = {0u, 0u, 0u, 0u};
      if (vec4 && i + 4 <= n) {
        const uint4 q = *reinterpret_cast<const uint4*>(key + i);
        k4[0] = q.x, k4[1] = q.y, k4[2] = q.z, k4[3] = q.w;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (i + u < n) k4[u] = key[i + u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) wave_hist_add(sh.hist, (k4[u] >> sft) & mask, k4[u] != 0u && (k4[u] & known_mask) == prefix);
    }
    __syncthreads();
    {  // the bin that holds the remaining-th largest key, counting from the top bin down: suffix sums of the 2048 bins by all 1024
       // threads (two bins each; Hillis-Steele over the reversed thread order in the candidate array's LDS, free at this point) -- one
       // thread walking the bins costs ~100 us per pass when the scores sit in the lower half of the value range
      static_assert(SEL_THREADS * 2 == 2048, "two histogram bins per thread");
      unsigned* scr = reinterpret_cast<unsigned*>(sh.cand);
      const unsigned c0 = sh.hist[2 * tid], c1 = sh.hist[2 * tid + 1], pair = c0 + c1;
      const int r = SEL_THREADS - 1 - tid;
      scr[r] = pair;
      __syncthreads();
      for (int o = 1; o < SEL_THREADS; o <<= 1) {
        const unsigned v = r >= o ? scr[r - o] : 0u;
        __syncthreads();
        scr[r] += v;
        __syncthreads();
      }
      const unsigned above = scr[r] - pair;         // keys in the bins above this thread's two
      const unsigned rem = sh.remaining;
      __syncthreads();
      const unsigned s1 = above + c1, s0 = s1 + c0;  // keys in bins >= 2 tid + 1, >= 2 tid
      if (above < rem && s1 >= rem) {
        sh.remaining = rem - above;
        sh.prefix = prefix | ((unsigned)(2 * tid + 1) << sft);
      } else if (s1 < rem && (s0 >= rem || tid == 0)) {  // bin 0 also when fewer than `remaining` keys exist at all (the walk's fall-through)
        sh.remaining = rem - s1;
        sh.prefix = prefix | ((unsigned)(2 * tid) << sft);
      }
    }
    __syncthreads();
    known_mask |= mask << sft;
  }
  const unsigned T = sh.prefix;  // K-th largest value (0 when fewer than K elements are non-zero)
  if (tid == 0) sh.ncand = 0;
  __syncthreads();
  for (long long i0 = 0; i0 < n; i0 += 4 * SEL_THREADS) {
    const long long i = i0 + 4 * tid;
    unsigned k4[4] = {0u, 0u, 0u, 0u};
    if (vec4 && i + 4 <= n) {
      const uint4 q = *reinterpret_cast<const uint4*>(key + i);
      k4[0] = q.x, k4[1] = q.y, k4[2] = q.z, k4[3] = q.w;
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i + u < n) k4[u] = key[i + u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned k = k4[u];
      if (k >= T && k != 0) {  // zero = suppressed / never a detection (conf > 0): a top-K that reaches into the zeros is cut short
        const unsigned slot = atomicAdd(&sh.ncand, 1u);
        if (slot < CAND_CAP) sh.cand[slot] = ((unsigned long long)(0xFFFFFFFFu - k) << 32) | (idx0 + (unsigned)(i + u));
      }
    }
  }
  __syncthreads();
  const unsigned ncand = sh.ncand;
  if (ncand > CAND_CAP) return -1;
  unsigned cap2 = 1;
  while (cap2 < ncand) cap2 <<= 1;
  for (unsigned i = ncand + tid; i < cap2; i += SEL_THREADS) sh.cand[i] = ~0ull;
  __syncthreads();
  for (unsigned k = 2; k <= cap2; k <<= 1)
    for (unsigned j = k >> 1; j > 0; j >>= 1) {
      for (unsigned i = tid; i < cap2; i += SEL_THREADS) {
        const unsigned l = i ^ j;
        if (l > i) {
          const unsigned long long x = sh.cand[i], z = sh.cand[l];
          const bool up = (i & k) == 0;
          if ((x > z) == up) {
            sh.cand[i] = z;
            sh.cand[l] = x;
          }
        }
      }
      __syncthreads();
    }
  return (int)ncand;
}

// K2a: grid (B, S).  Slice s of an image's H*W*C scores -> its own sorted top-K, as 64-bit keys in slice_top[b][s][K]
// (~0 = empty, first word ~0 of slot 0 with the second = 0 marks "too many ties").  The global top-K under the strict
// (score, index) order is a subset of the union of the per-slice top-Ks.
__global__ __launch_bounds__(SEL_THREADS) void select_slice_kernel(const float* scores, long long N, int S, int K,
                                                                   unsigned long long* slice_top) {
  __shared__ SelShared sh;
  const int b = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x;
  const long long per = (N + S - 1) / S;
  const long long i0 = (long long)sl * per;
  const long long n = i0 < N ? (N - i0 < per ? N - i0 : per) : 0;
  const unsigned* key = reinterpret_cast<const unsigned*>(scores + (long long)b * N + i0);
  const int nc = topk_sorted(key, n, (unsigned)i0, K, sh);
  unsigned long long* out = slice_top + ((long long)b * S + sl) * K;
  for (int i = tid; i < K; i += SEL_THREADS) out[i] = (nc >= 0 && i < nc) ? sh.cand[i] : ~0ull;
  if (nc < 0 && tid == 0) out[0] = 0xFFFFFFFF00000000ull;  // overflow marker (a real key never has an all-ones value word)
}

// K2b: one workgroup per image: merge the S slice lists (one more bitonic sort of S*K keys), boxes, mask, DIoU-NMS.
__global__ __launch_bounds__(SEL_THREADS) void finish_kernel(const float* pred, int ld, int H, int W, int C, int reg_off, int wh_off,
                                                             const float* scores, const unsigned long long* slice_top, int S, int K,
                                                             float conf, float nms_thr, int use_nms, DecodeOut o) {
  __shared__ unsigned long long cand[CAND_CAP];
  __shared__ float4 s_box[MAX_K];
  __shared__ unsigned char s_alive[MAX_K];
  __shared__ int s_over;
  const int b = blockIdx.x, tid = threadIdx.x;
  const long long N = (long long)H * W * C;
  const float* sc = scores + (long long)b * N;
  const int total = S * K;
  unsigned cap2 = 1;
  while ((int)cap2 < total) cap2 <<= 1;
  if (tid == 0) s_over = 0;
  __syncthreads();
  for (unsigned i = tid; i < cap2; i += SEL_THREADS) {
    const unsigned long long v = (int)i < total ? slice_top[(long long)b * total + i] : ~0ull;
    if (v == 0xFFFFFFFF00000000ull) s_over = 1;
    cand[i] = v;
  }
  __syncthreads();
  if (s_over) {
    if (tid == 0) o.counts[b] = -1;
    return;
  }
  for (unsigned k = 2; k <= cap2; k <<= 1)
    for (unsigned j = k >> 1; j > 0; j >>= 1) {
      for (unsigned i = tid; i < cap2; i += SEL_THREADS) {
        const unsigned l = i ^ j;
        if (l > i) {
          const unsigned long long x = cand[i], z = cand[l];
          const bool up = (i & k) == 0;
          if ((x > z) == up) {
            cand[i] = z;
            cand[l] = x;
          }
        }
      }
      __syncthreads();
    }
  int ncand = 0;  // valid keys come first after the sort
  for (int i = 0; i < K && i < total; ++i)
    if (cand[i] != ~0ull) ++ncand;
  // ---- 3. the K winners: class, pixel, box (centernet.py:282-297) ----
  const int kk = min(K, (int)ncand);
  if (tid < kk) {
    const unsigned idx = (unsigned)(cand[tid] & 0xFFFFFFFFu);
    const int c = (int)(idx % (unsigned)C);
    const unsigned pix = idx / (unsigned)C;
    const int y = (int)(pix / (unsigned)W), x = (int)(pix % (unsigned)W);
    const float* row = pred + ((long long)b * H * W + pix) * ld;
    const float score = sc[idx];
    float cx = __fadd_rn((float)x, row[reg_off]), cy = __fadd_rn((float)y, row[reg_off + 1]);
    float bw = row[wh_off], bh = row[wh_off + 1];
    cx = fminf(fmaxf(__fdiv_rn(cx, (float)W), 0.f), 1.f);
    bw = fminf(fmaxf(__fdiv_rn(bw, (float)W), 0.f), 1.f);
    cy = fminf(fmaxf(__fdiv_rn(cy, (float)H), 0.f), 1.f);
    bh = fminf(fmaxf(__fdiv_rn(bh, (float)H), 0.f), 1.f);
    const float hw2 = __fdiv_rn(bw, 2.f), hh2 = __fdiv_rn(bh, 2.f);
    const float4 bx = make_float4(__fsub_rn(cx, hw2), __fsub_rn(cy, hh2), __fadd_rn(cx, hw2), __fadd_rn(cy, hh2));
    s_box[tid] = bx;
    s_alive[tid] = score >= conf ? 1 : 0;
    float* ob = o.boxes + ((long long)b * K + tid) * 4;
    ob[0] = bx.x;
    ob[1] = bx.y;
    ob[2] = bx.z;
    ob[3] = bx.w;
    o.scores[(long long)b * K + tid] = score;
    o.classes[(long long)b * K + tid] = c;
    o.topk_index[(long long)b * K + tid] = (int)idx;
  }
  if (tid >= kk && tid < K) {  // fewer than K non-zero peaks: the rest of the list is empty
    float* ob = o.boxes + ((long long)b * K + tid) * 4;
    ob[0] = ob[1] = ob[2] = ob[3] = 0.f;
    o.scores[(long long)b * K + tid] = 0.f;
    o.classes[(long long)b * K + tid] = 0;
    o.topk_index[(long long)b * K + tid] = -1;
  }
  __syncthreads();
  // ---- 4. score mask + greedy DIoU-NMS in list order (core/utils/nms.py:9-31), one wave ----
  if (tid < 64) {
    int nkeep = 0;
    for (int i = 0; i < kk; ++i) {
      if (!s_alive[i]) continue;  // wave-uniform (LDS)
      if (tid == 0) o.keep[(long long)b * K + nkeep] = i;
      ++nkeep;
      if (use_nms) {
        const float4 bi = s_box[i];
        for (int j = i + 1 + tid; j < kk; j += 64)
          if (s_alive[j] && !(diou(bi, s_box[j]) <= nms_thr)) s_alive[j] = 0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (tid == 0) o.counts[b] = nkeep;
  }
}

}  // namespace

static int slices_for(int K) {  // slices per image whose K-lists fit the merge sort together
  int S = 16;
  while (S > 1 && S * K > CAND_CAP) S >>= 1;
  return S;
}

extern "C" int64_t cvx_centernet_decode_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t nc) {
  return ((int64_t)B * H * W * nc + 64) * 4 + (int64_t)B * CAND_CAP * 8 + 512;
}

extern "C" int cvx_centernet_decode(const float* pred, int32_t pred_ld, int32_t B, int32_t H, int32_t W, int32_t nc, int32_t reg_col,
                                    int32_t wh_col, int32_t K, float conf, float nms_thr, int32_t use_nms, float* boxes, float* scores,
                                    int32_t* classes, int32_t* topk_index, int32_t* keep, int32_t* counts, void* workspace,
                                    int64_t workspace_bytes, void* hip_stream) {
  CVX_CHECK(pred && boxes && scores && classes && topk_index && keep && counts && workspace, "null arguments");
  CVX_CHECK(B > 0 && H > 0 && W > 0 && nc > 0 && K >= 1 && K <= MAX_K, "bad sizes (1 <= K <= 256)");
  CVX_CHECK((long long)H * W * nc < (1LL << 31) && (long long)H * W * nc >= K, "heat-map size");
  CVX_CHECK(pred_ld >= nc && reg_col >= 0 && reg_col + 2 <= pred_ld && wh_col >= 0 && wh_col + 2 <= pred_ld, "column offsets");
  CVX_CHECK(workspace_bytes >= cvx_centernet_decode_workspace_bytes(B, H, W, nc), "workspace too small");
  hipStream_t st = (hipStream_t)hip_stream;
  float* ws = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  const long long n = (long long)B * H * W * nc;
  hipLaunchKernelGGL(peak_scores_kernel, dim3(cvx_cdiv(n, 256)), dim3(256), 0, st, pred, pred_ld, B, H, W, nc, ws);
  DecodeOut o{boxes, scores, classes, topk_index, keep, counts};
  const int S = slices_for(K);
  unsigned long long* slice_top = reinterpret_cast<unsigned long long*>(ws + ((n + 63) & ~63LL));
  hipLaunchKernelGGL(select_slice_kernel, dim3(B, S), dim3(SEL_THREADS), 0, st, ws, (long long)H * W * nc, S, K, slice_top);
  hipLaunchKernelGGL(finish_kernel, dim3(B), dim3(SEL_THREADS), 0, st, pred, pred_ld, H, W, nc, reg_col, wh_col, ws, slice_top, S, K, conf, nms_thr,
                     use_nms, o);
  CVX_HIP(hipGetLastError());
  return 0;
}
