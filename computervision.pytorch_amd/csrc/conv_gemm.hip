// GEMM-shaped implicit convolution for the big-channel layers (gfx950): K = taps * Cin >= 1024, Cin a multiple of 32 -- VGG's 256..1024-
// channel 3x3 layers, ResNet-101's bottleneck 1x1 / 3x3 convs, the atrous ASPP branches, their data gradients.
//
//   * 512 threads = 8 waves as 4 (pixels) x 2 (channels); macro tile BM x BN = (4 * MT * 32) x (2 * NT * 32): 256 x 256, 256 x 128,
//     128 x 256 or 128 x 128; every wave owns MT x NT tiles of v_mfma_f32_32x32x16_f16 (weights as the A operand: a lane ends up with
//     4 consecutive channels of one pixel);
//   * BOTH operands stream through one LDS-DMA ring of CHUNK = 32 K-values (2 MFMA K-steps), 4 slots, 2 chunks in flight beyond the
//     one being computed (counted vmcnt + one raw s_barrier per chunk):
//       - pixels: the implicit-GEMM gather -- chunk (tap, 32 channels) of pixel p is 64 contiguous bytes of the NHWC view, DMA'd to
//         LDS unit p * 5 + (0..3): an ODD pixel stride (5 units of 16 B), so the 32 lanes of a fragment read fall on 32 different
//         16-byte slots (no bank conflict, tools/lds_conflicts.py); the zero page feeds padding and pad units;
//       - weights: pre-packed once per launch into the ring image order ([chunk][K-step][k-half][BN rows][8]: gemm_pack_kernel), so a
//         chunk is one contiguous block and every DMA piece reads 1 KiB of consecutive bytes;
//   * fragments of the second K-step of a chunk are requested before the MFMAs of the first (inline-asm ds_read: the compiler's own
//     waitcnt insertion would put lgkmcnt(0) in front of every MFMA block), two waves per SIMD cover the rest;
//   * blockIdx -> (pixel tile, channel tile) keeps the channel tiles of one pixel tile on one XCD (they share the gathered pixels in
//     that XCD's L2).
// Roofline: MFMA (these shapes sit above the 315 FLOP/B ridge).  Algorithmic bytes per launch: engine.hip conv_bytes.
#include <algorithm>
#include <cstring>
#include <map>
#include <vector>

#include "conv_tile_common.h"

namespace {
using namespace cvx_tile;

typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int GSLOTS = 4;
constexpr int GW = 8;        // waves
constexpr int CHUNK = 32;    // K-values per ring chunk
constexpr int APS = 5;       // LDS units (16 B) per pixel and chunk: 4 data + 1 pad (odd stride)

template <int MT, int NT>
struct GemmGeom {
  static constexpr int BM = 4 * MT * 32, BN = 2 * NT * 32;
  static constexpr int A_UNITS = BM * APS, B_UNITS = BN * 4;
  static constexpr int A_PIECES = (A_UNITS + 63) / 64, B_PIECES = B_UNITS / 64;
  static constexpr int A_PER_WAVE = (A_PIECES + GW - 1) / GW, B_PER_WAVE = (B_PIECES + GW - 1) / GW;
  static constexpr int PER_WAVE = A_PER_WAVE + B_PER_WAVE;  // DMA instructions per wave per chunk
  static constexpr int SLOT_UNITS = A_PIECES * 64 + B_UNITS;
  static constexpr int LDS_BYTES = GSLOTS * SLOT_UNITS * 16 + 1024 + 4 * BN * 2 * 4;  // ring | dump | statistics scratch
};

__device__ __forceinline__ h8 lds_frag(unsigned a) {
  h8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a));
  return v;
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// [rows][taps*Cin] fp16 (row pitch src_ld, tap block wtap[t] at wtap[t]*Cin) -> [n-block][chunk][K-step][k-half][BN rows][8]
struct GemmPackArgs {
  const half_t* src;
  half_t* dst;
  const ConvTap* taps;  // device table: chunk block `tap` reads weight tap taps[tap].wtap
  int src_ld, rows, Cin, ntaps, BN, nblocks, chunks;
  unsigned long long* dbg;  // cvx_debug_clock_buffer: range check of the source reads (slot 8..)
};
__global__ void gemm_pack_kernel(const GemmPackArgs a) {
  const long long per_block = (long long)a.chunks * 4 * a.BN;  // units per n-block
  const long long total = per_block * a.nblocks;
  const int cpt = a.Cin / CHUNK;  // chunks per tap
  for (long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (long long)gridDim.x * blockDim.x) {
    const int nb = (int)(u / per_block);
    const long long r0 = u - (long long)nb * per_block;
    const int chunk = (int)(r0 / (4 * a.BN));
    const int r1 = (int)(r0 - (long long)chunk * 4 * a.BN);
    const int kh = r1 / a.BN, row = r1 - kh * a.BN;  // kh = K-step * 2 + half
    const int n = nb * a.BN + row;
    const int tap = chunk / cpt, c0 = (chunk - tap * cpt) * CHUNK + kh * 8;
    h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (n < a.rows) {
      const long long off = (long long)n * a.src_ld + a.taps[tap].wtap * a.Cin + c0;
      if (a.dbg && (off < 0 || off + 8 > (long long)a.rows * a.src_ld || tap >= a.ntaps)) {
        if (atomicAdd(a.dbg + 8, 1ull) == 0) {
          a.dbg[9] = (unsigned long long)off;
          a.dbg[10] = (unsigned long long)tap;
          a.dbg[11] = (unsigned long long)a.taps[tap].wtap;
          a.dbg[12] = (unsigned long long)u;
        }
      } else {
        v = *reinterpret_cast<const h8*>(a.src + off);
      }
    }
    *reinterpret_cast<h8*>(a.dst + u * 8) = v;
  }
}

// tuning / debugging aid (cvx_debug_clock_buffer set): every DMA source address is range-checked against the operand it belongs to; a
// violation is recorded in the buffer (slot 0: count, 1: kind, 2: offset, 3: chunk, 4: block) and the access goes to the zero page instead
__device__ __forceinline__ const half_t* dbg_check(const ConvParams& p, const half_t* g, const half_t* base, long long elems, int kind, int chunk) {
  if (!p.clk || g == p.zeros) return g;
  const long long off = g - base;
  if (off >= 0 && off + 8 <= elems) return g;
  if (atomicAdd(p.clk, 1ull) == 0) {
    p.clk[1] = (unsigned long long)kind;
    p.clk[2] = (unsigned long long)off;
    p.clk[3] = (unsigned long long)chunk;
    p.clk[4] = (unsigned long long)blockIdx.x;
    p.clk[5] = (unsigned long long)threadIdx.x;
  }
  return p.zeros;
}

// epilogue body of one kind (EPI): lane holds pixel lr of sub-tile i, channels nbase + j*32 + 8 g + 4 lh + (0..3) in acc[i][j][4 g ..]
template <int MT, int NT, int EPI>
__device__ __forceinline__ void gemm_store(const ConvParams& p, const f16v (&acc)[MT][NT], const long long (&out_off)[MT], const long long (&res_off)[MT],
                                           const bool (&pvalid)[MT], int nbase, int lh) {
  const int act_kind = p.act_kind, res_pre = p.res_pre, accumulate = p.accumulate;
  const half_t* res = p.res;
  const bool guard = p.clk != nullptr;
  const long long out_elems = (long long)p.B * p.out_bstride;
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n0 = nbase + j * 32 + g * 8 + lh * 4;
      if (n0 < p.Cout) {
        f4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI == CVX_EPI_AFFINE_SILU) {
          sc = *reinterpret_cast<const f4*>(p.scale + n0);
          sh = *reinterpret_cast<const f4*>(p.shift + n0);
        } else if constexpr (EPI == CVX_EPI_BIAS_F32) {
          sh = *reinterpret_cast<const f4*>(p.bias + n0);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if (pvalid[i]) {
            const long long off = out_off[i] + n0;
            f4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][4 * g + r] * sc[r] + sh[r];
            if constexpr (EPI == CVX_EPI_BIAS_F32) {
              *reinterpret_cast<f4*>(p.out32 + off) = v;
            } else {
              if constexpr (EPI == CVX_EPI_AFFINE_SILU) {
                f4 rv = {0.f, 0.f, 0.f, 0.f};
                if (res) {
                  const h4 rr = *reinterpret_cast<const h4*>(res + res_off[i] + n0);
#pragma unroll
                  for (int r = 0; r < 4; ++r) rv[r] = (float)rr[r];
                }
                if (res_pre) {
#pragma unroll
                  for (int r = 0; r < 4; ++r) v[r] += rv[r];
                }
                if (act_kind == 0) {
#pragma unroll
                  for (int r = 0; r < 4; ++r) v[r] = cvx_silu(v[r]);
                } else if (act_kind == 1) {
#pragma unroll
                  for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                if (!res_pre) {
#pragma unroll
                  for (int r = 0; r < 4; ++r) v[r] += rv[r];
                }
              }
              const bool bad = guard && (off < 0 || off + 4 > out_elems);
              if (bad) {
                if (atomicAdd(p.clk + 16, 1ull) == 0) {
                  p.clk[17] = (unsigned long long)off;
                  p.clk[18] = (unsigned long long)blockIdx.x;
                }
              } else {
                half_t* dst = p.out16 + off;
                if (accumulate) {
                  const h4 old = *reinterpret_cast<const h4*>(dst);
#pragma unroll
                  for (int r = 0; r < 4; ++r) v[r] += (float)old[r];
                }
                *reinterpret_cast<h4*>(dst) = h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
              }
            }
          }
        }
      }
    }
}

template <int MT, int NT>
__global__ __launch_bounds__(64 * GW) void conv_gemm_kernel(const ConvParams p, const half_t* __restrict__ wpk, int m_tiles, int n_tiles, int nchunks) {
  using G = GemmGeom<MT, NT>;
  constexpr int BM = G::BM, BN = G::BN;
  constexpr int NWAIT = (GSLOTS - 2) * G::PER_WAVE;
  static_assert(NWAIT <= 63, "vmcnt field");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* dump = smem + GSLOTS * G::SLOT_UNITS * 16;
  float* sStat = reinterpret_cast<float*>(dump + 1024);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave & 3, wn = wave >> 2;
  const int lr = lane & 31, lh = lane >> 5;
  // tile of this workgroup: the channel tiles of one pixel tile run on one XCD (blocks b and b + 8 share an XCD)
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int n_tile = seq % n_tiles;
  const int m_tile = (seq / n_tiles) * 8 + xcd;
  if (m_tile >= m_tiles) return;
  if (p.dbg & 4) return;
  const long long M = (long long)p.B * p.OH2 * p.OW2;
  const long long m_base = (long long)m_tile * BM;
  const int cpt = p.Cin / CHUNK;  // chunks per tap

  // ---- per-lane gather assignment: A piece q of this wave covers units piece * 64 + lane -> (pixel, 8-channel group) ----
  const half_t* a_src[G::A_PER_WAVE];
  int a_ih[G::A_PER_WAVE], a_iw[G::A_PER_WAVE], a_ch[G::A_PER_WAVE];
#pragma unroll
  for (int q = 0; q < G::A_PER_WAVE; ++q) {
    const int piece = q * GW + wave;
    const int u = piece * 64 + lane;
    const int pix = u / APS;
    a_ch[q] = u - pix * APS;
    a_src[q] = nullptr;
    a_ih[q] = a_iw[q] = 0;
    const long long m = m_base + pix;
    if (piece < G::A_PIECES && pix < BM && a_ch[q] < 4 && m < M) {
      const unsigned mu = (unsigned)m;
      const unsigned tq = mu / (unsigned)p.OW2;
      const int ow2 = (int)(mu - tq * (unsigned)p.OW2);
      const int b = (int)(tq / (unsigned)p.OH2);
      const int oh2 = (int)(tq - (unsigned)b * (unsigned)p.OH2);
      a_ih[q] = oh2 * p.IS;
      a_iw[q] = ow2 * p.IS;
      a_src[q] = p.in + (long long)b * p.in_bstride + a_ch[q] * 8;
    }
  }
  const half_t* wblk = wpk + (long long)n_tile * nchunks * (4 * BN * 8);

  auto issue = [&](int chunk, int slot) __attribute__((always_inline)) {
    unsigned char* sb = smem + slot * (G::SLOT_UNITS * 16);
    if (chunk < nchunks) {
      const int tap = chunk / cpt, c0 = (chunk - tap * cpt) * CHUNK;
      const ConvTap td = p.taps[tap];
#pragma unroll
      for (int q = 0; q < G::A_PER_WAVE; ++q) {
        const int piece = q * GW + wave;
        const half_t* g = p.zeros;
        const int ih = a_ih[q] + td.dh, iw = a_iw[q] + td.dw;
        if (a_src[q] && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW) g = a_src[q] + ((long long)ih * p.IW + iw) * p.in_ld + c0;
        g = dbg_check(p, g, p.in, (long long)p.B * p.in_bstride, 1, chunk);
        unsigned char* dst = piece < G::A_PIECES ? sb + piece * 1024 : dump;
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)g, (lds_void_ptr)dst, 16, 0, 0);
      }
      const half_t* wsrc = wblk + (long long)chunk * (4 * BN * 8) + lane * 8;
#pragma unroll
      for (int q = 0; q < G::B_PER_WAVE; ++q) {
        const int piece = q * GW + wave;
        const bool real = piece < G::B_PIECES;
        const half_t* gw = dbg_check(p, real ? wsrc + piece * 512 : p.zeros, wpk, (long long)n_tiles * nchunks * (4 * BN * 8), 2, chunk);
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)gw, (lds_void_ptr)(real ? sb + G::A_PIECES * 1024 + piece * 1024 : dump), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < G::PER_WAVE; ++q) __builtin_amdgcn_global_load_lds((gbl_void_ptr)p.zeros, (lds_void_ptr)dump, 16, 0, 0);
    }
  };

  f16v acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int s = 0; s < GSLOTS - 1; ++s) issue(s, s);
  if (p.dbg & 8) {
    wait_vmcnt<0>();
    return;
  }

  // fragment addresses inside a slot (bytes): pixel operand (sub-tile i, K-step ks) and weight operand (tile j, K-step ks)
  const unsigned a_lane = lds_addr(smem) + (unsigned)(((wm * MT * 32 + lr) * APS + lh) * 16);
  const unsigned b_lane = lds_addr(smem) + (unsigned)(G::A_PIECES * 1024 + ((lh * BN) + wn * NT * 32 + lr) * 16);
  for (int c = 0; c < nchunks; ++c) {
    wait_vmcnt<NWAIT>();
    workgroup_barrier();  // chunk c landed for every wave; the slot of chunk c - 1 is free
    issue(c + GSLOTS - 1, (c + GSLOTS - 1) % GSLOTS);
    const unsigned so = (unsigned)((c % GSLOTS) * (G::SLOT_UNITS * 16));
    h8 xa[2][MT], wb[2][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) xa[0][i] = lds_frag(a_lane + so + i * (32 * APS * 16));
#pragma unroll
    for (int j = 0; j < NT; ++j) wb[0][j] = lds_frag(b_lane + so + j * (32 * 16));
#pragma unroll
    for (int i = 0; i < MT; ++i) xa[1][i] = lds_frag(a_lane + so + i * (32 * APS * 16) + 2 * 16);
#pragma unroll
    for (int j = 0; j < NT; ++j) wb[1][j] = lds_frag(b_lane + so + 2 * BN * 16 + j * (32 * 16));
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT + NT) : "memory");  // the first K-step's fragments (LDS returns in order)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[0][j], xa[0][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    wait_lgkm();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[1][j], xa[1][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  wait_vmcnt<0>();  // surplus prefetches (dump pieces) retire before the epilogue's own loads share the counter
  if (p.dbg & 16) return;

  // ---- epilogue: lane holds pixel lr of sub-tile i, channels n_tile*BN + (wn*NT + j)*32 + 8 g + 4 lh + (0..3) in acc[i][j][4 g ..] ----
  long long out_off[MT], res_off[MT];
  bool pvalid[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const long long m = m_base + (wm * MT + i) * 32 + lr;
    pvalid[i] = m < M;
    const unsigned mu = pvalid[i] ? (unsigned)m : 0u;
    const unsigned tq = mu / (unsigned)p.OW2;
    const int ow2 = (int)(mu - tq * (unsigned)p.OW2);
    const int b = (int)(tq / (unsigned)p.OH2);
    const int oh2 = (int)(tq - (unsigned)b * (unsigned)p.OH2);
    const long long pix = (long long)(oh2 * p.OS + p.oph) * p.OWr + (ow2 * p.OS + p.opw);
    out_off[i] = (long long)b * p.out_bstride + pix * p.out_ld;
    res_off[i] = (long long)b * p.res_bstride + pix * p.res_ld;
  }
  const int nbase = n_tile * BN + wn * NT * 32;
  if (p.epi == CVX_EPI_RAW_STATS) {
    // raw fp32 output + per-channel (sum, sum of squares): lanes of one k-half hold 32 pixels of the same 4 channels
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int chl = (wn * NT + j) * 32 + g * 8 + lh * 4;  // channel inside the block's BN range
        const int n0 = n_tile * BN + chl;
        float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if (!pvalid[i]) continue;
          f4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          if (n0 < p.Cout) *reinterpret_cast<f4*>(p.out32 + out_off[i] + n0) = v;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s1[r] += v[r];
            s2[r] += v[r] * v[r];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a = s1[r], b2 = s2[r];
#pragma unroll
          for (int o = 1; o < 32; o <<= 1) {
            a += __shfl_xor(a, o);
            b2 += __shfl_xor(b2, o);
          }
          if (lr == 0) {
            sStat[(wm * BN + chl + r) * 2 + 0] = a;
            sStat[(wm * BN + chl + r) * 2 + 1] = b2;
          }
        }
      }
    __syncthreads();
    for (int t = tid; t < BN * 2; t += 64 * GW) {
      const int ch = t >> 1, which = t & 1;
      const int n = n_tile * BN + ch;
      if (n < p.Cout) {
        const float v = (sStat[(0 * BN + ch) * 2 + which] + sStat[(1 * BN + ch) * 2 + which]) + (sStat[(2 * BN + ch) * 2 + which] + sStat[(3 * BN + ch) * 2 + which]);
        cvx_fix_atomic_add(p.stats, ((long long)(m_tile % p.stats_replicas) * p.Cout + n) * 2 + which, v);
      }
    }
    return;
  }
  // one straight-line body per epilogue kind: with the kinds tested inside the (j, g, i) loops hipcc 7.2 structurised the control flow
  // into a path that left the channel offset of the PLAIN store undefined (a stale pointer was used: the faults of the first bring-up)
  switch (p.epi) {
    case CVX_EPI_AFFINE_SILU: gemm_store<MT, NT, CVX_EPI_AFFINE_SILU>(p, acc, out_off, res_off, pvalid, nbase, lh); break;
    case CVX_EPI_BIAS_F32: gemm_store<MT, NT, CVX_EPI_BIAS_F32>(p, acc, out_off, res_off, pvalid, nbase, lh); break;
    default: gemm_store<MT, NT, CVX_EPI_PLAIN>(p, acc, out_off, res_off, pvalid, nbase, lh); break;
  }
}

// packed weights of a (weight pointer, tap table) pair: re-packed at EVERY launch (the fp16 shadows change with every optimiser step),
// the buffer itself is kept
struct PackSlot {
  half_t* buf = nullptr;
  size_t bytes = 0;
};
std::map<std::pair<const void*, std::pair<const void*, int>>, PackSlot> g_pack;

template <int MT, int NT>
int launch_gemm(const ConvParams& p, hipStream_t stream) {
  using G = GemmGeom<MT, NT>;
  const long long M = (long long)p.B * p.OH2 * p.OW2;
  const int m_tiles = (int)((M + G::BM - 1) / G::BM), n_tiles = (p.Cout + G::BN - 1) / G::BN;
  const int nchunks = p.ntaps * (p.Cin / CHUNK);
  // ---- weights -> ring image order ----
  PackSlot& ps = g_pack[{p.wt, {p.taps, G::BN}}];
  const size_t need = (size_t)n_tiles * nchunks * 4 * G::BN * 16;
  if (ps.bytes < need) {
    if (ps.buf) CVX_HIP(hipFree(ps.buf));
    CVX_HIP(hipMalloc((void**)&ps.buf, need));
    ps.bytes = need;
  }
  GemmPackArgs a;
  memset(&a, 0, sizeof(a));
  a.src = p.wt;
  a.dst = ps.buf;
  a.src_ld = p.wt_ld;
  a.rows = p.Cout;
  a.Cin = p.Cin;
  a.ntaps = p.ntaps;
  a.BN = G::BN;
  a.nblocks = n_tiles;
  a.chunks = nchunks;
  a.taps = p.taps;
  a.dbg = p.clk;
  const long long units = (long long)n_tiles * nchunks * 4 * G::BN;
  static const int dbg = cvx_tune_int("CVX_GEMM_DBG", 0);  // tuning build: 1 no pack, 2 no main kernel, 4 / 8 / 16 main kernel stops after entry / prologue / K loop
  if (!(dbg & 1)) hipLaunchKernelGGL(gemm_pack_kernel, dim3((unsigned)std::min<long long>(1024, (units + 255) / 256)), dim3(256), 0, stream, a);
  if (dbg & 2) return 0;
  ConvParams pd = p;
  pd.dbg = dbg;
  static unsigned long long optin_mask = 0;
  CVX_TRY(cvx_lds_optin((const void*)conv_gemm_kernel<MT, NT>, G::LDS_BYTES, &optin_mask));
  const int grid = ((m_tiles + 7) / 8) * 8 * n_tiles;
  hipLaunchKernelGGL((conv_gemm_kernel<MT, NT>), dim3(grid), dim3(64 * GW), G::LDS_BYTES, stream, pd, (const half_t*)ps.buf, m_tiles, n_tiles, nchunks);
  return 0;
}

}  // namespace

bool cvx_conv_gemm_shape_ok(const ConvParams& p) {
  const long long M = (long long)p.B * p.OH2 * p.OW2;
  return p.zeros && p.nphase <= 1 && p.Cin % CHUNK == 0 && p.ntaps <= CVX_MAX_TAPS && p.Cout % 4 == 0 && M < (1LL << 31);
}

bool cvx_conv_gemm_supported(const ConvParams& p) {
  static const bool off = cvx_tune_set("CVX_NO_GEMM");
  if (off || !cvx_conv_gemm_shape_ok(p)) return false;
  const long long M = (long long)p.B * p.OH2 * p.OW2;
  const long long K = (long long)p.ntaps * p.Cin;
  // measured against the LDS-DMA ring kernel (tools/gemm_probe.py, profiles/r03_gemm_probe.txt): ahead where both dimensions of the
  // weight matrix are large and the pixel count fills two rounds of 256 x 256 tiles; the tuning build lowers the gates for A/B runs
  static const int kmin = cvx_tune_int("CVX_GEMM_KMIN", 4608);
  static const int mmin = cvx_tune_int("CVX_GEMM_MMIN", 32768);
  return K >= kmin && p.Cout >= 256 && M >= mmin;
}

int cvx_conv_gemm_launch(const ConvParams& p, hipStream_t stream) {
  const long long M = (long long)p.B * p.OH2 * p.OW2;
  // macro tile: the largest whose grid still gives every CU a workgroup (256 of them)
  static const int force = cvx_tune_int("CVX_GEMM_TILE", 0);  // 1: 256x256, 2: 256x128, 3: 128x256, 4: 128x128
  auto wgs = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((p.Cout + bn - 1) / bn); };
  int pick = 4;
  if (p.Cout >= 256 && wgs(256, 256) >= 256) pick = 1;
  else if (wgs(256, 128) >= 256) pick = 2;
  else if (p.Cout >= 256 && wgs(128, 256) >= 256) pick = 3;
  if (force) pick = force;
  switch (pick) {
    case 1: CVX_TRY((launch_gemm<2, 4>(p, stream))); break;
    case 2: CVX_TRY((launch_gemm<2, 2>(p, stream))); break;
    case 3: CVX_TRY((launch_gemm<1, 4>(p, stream))); break;
    default: CVX_TRY((launch_gemm<1, 2>(p, stream))); break;
  }
  CVX_HIP(hipGetLastError());
  return 0;
}
