// Weight gradient of the 3x3 / stride-1 / pad-1 layers with 16 ... 144 channels as FEW, FAT workgroups (gfx950) -- round 5.
//
//   dW[co][tap][ci] = sum over pixels  dy[p][co] * x[p + tap][ci]
//
// These launches run on the lowest-priority stream beside the backward pass, and what they cost the step is the CU-time they hold
// (DESIGN.md 5d): the older kernels (conv_wgrad_halo.hip / conv_wgrad.hip) keep 126 ... 512 four-wave workgroups resident at 8 % of the
// MFMA peak.  Here a workgroup is EIGHT waves with a whole CU's LDS, owns a (16 * CO_T output channels) x (all nine taps) x (16 * CI_T input
// channels) block of the result -- 64 x 576 for a 64 -> 64 layer -- and the launch has only as many workgroups as the layer's work needs:
//
//   * chunk = R output rows x Wc columns of one image.  dy (R x (Wc + 2), gutter columns zero) and x ((R + 2) x (Wc + 2), halo) are
//     DMA'd (`buffer_load_dwordx4 ... lds`) into a ring slot as they lie in memory, in a PADDED LINEAR pixel space of pitch Wc + 2: tap
//     (dh, dw) is the constant offset (dh + 1) * (Wc + 2) + dw + 1 between the two images, a K-step is 32 consecutive padded pixels
//     whatever the map width.  Zeros (gutters, rows / columns outside the image, channel padding, the tail of a chunk) are
//     out-of-range DMA lanes: the buffer descriptor's bounds check writes them;
//   * MFMA fragments by the transposing LDS read ds_read_b64_tr_b16 (the reduction index -- pixel -- is the slow index of both images);
//     lane group kg, read j of a K-step take pixels 16 j + 4 kg ... + 3, 32-byte channel slots XOR-swizzled by the pixel on the global
//     side: bank-conflict free for every pitch used;
//   * v_mfma_f32_16x16x32_f16, dy as the A operand.  The waves split (co tiles) x (column tiles = tap x ci tile) [x K-steps for the
//     narrow layers]; a wave's CO_TW + JW fragments feed CO_TW * JW MFMAs per K-step, and the fragment reads of step s + 1 are issued
//     BETWEEN the MFMAs of step s with compile-time LDS offsets (pitches are template constants): no address arithmetic in the loop but
//     one add per fragment and pair of K-steps;
//   * one barrier per chunk, NS - 1 chunks in flight; each pixel split stores its fp32 slab (plain stores, deterministic) for
//     cvx_reduce_slabs.
//
// Roofline: MFMA from 64 channels (288 FLOP per operand byte), HBM below.  Reference: autograd's weight gradient of nn.Conv2d in
// core/models/yolov8/modules.py:19-33 (Conv), :124-135 (Bottleneck), :407-455 (Detect).
#include <algorithm>
#include <cstring>

#include "conv_tile_common.h"

namespace {
using namespace cvx_tile;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int K3_MAXP = 8;  // DMA pieces (1 KiB) per wave and chunk, dy + x
constexpr int K3_KSW = 4;   // K-steps per wave and chunk: a chunk's body is straight-line code

struct K3Plan {
  int W, H, Wc, Wp, ncb;     // map, column-block width, padded pitch Wc + 2, column blocks per row
  int R, Tx;                 // rows per chunk, x pixels of the chunk tile (dy pixels per chunk: the configuration's QD)
  int d_pieces, x_pieces;    // 1-KiB pieces of the two images; piece q of a slot lies at q * 1024 (dy first)
  int slot_bytes, nslot;
  int upi, total_units, units_per_split;  // chunks per image, in all, per pixel split
  int n_co_blk, n_ci_blk;
  int lds_bytes;
  int s2, PP, IW, IH;        // stride 2: the x image is FOUR phase planes of PP padded pixels each (see the kernel); the input map
  unsigned m_wp;             // ceil(2^32 / Wp)
  unsigned long long* clk;   // tuning aid (cvx_debug_clock_buffer): thread 0 of every workgroup stores 100 MHz stamps, 8 slots each
};

template <int CO_T_, int CI_T_, int WC_, int WN_, int WK_>
struct K3Cfg {
  static constexpr int CO_T = CO_T_, CI_T = CI_T_, WC = WC_, WN = WN_, WK = WK_;
  static constexpr int NW = WC * WN * WK;
  static constexpr int CO_TW = CO_T / WC;
  static constexpr int NJ = 9 * CI_T;
  static constexpr int JW = (NJ + WN - 1) / WN;
  static constexpr int QD = 32 * K3_KSW * WK;            // dy pixels (K values) per chunk
  static constexpr int PD = CO_T * 32, PX = CI_T * 32;  // bytes per pixel of the two LDS images
  static constexpr bool d_pow2 = (CO_T & (CO_T - 1)) == 0, x_pow2 = (CI_T & (CI_T - 1)) == 0;
  // slot swizzle f(t) = (t >> sh) & mask: 1 / 2 / 4 / 8 slots per pixel; odd counts need none
  static constexpr int D_MASK = d_pow2 ? CO_T - 1 : 0, X_MASK = x_pow2 ? CI_T - 1 : 0;
  static constexpr int D_SH = CO_T == 2 ? 2 : CO_T == 4 ? 1 : 0;
  static constexpr int X_SH = CI_T == 2 ? 2 : CI_T == 4 ? 1 : 0;
  static_assert(CO_T % WC == 0, "co tiles divide over the co waves");
  static_assert(NW <= 8, "at most eight waves");
};

__device__ __forceinline__ unsigned k3_lds32(const void* p) { return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p; }
template <int OFF>
__device__ __forceinline__ s4 k3_tr(unsigned a) {
  s4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}
__device__ __forceinline__ h8 k3_join(s4 lo, s4 hi) {
  const s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(h8, v);
}
__device__ __forceinline__ void k3_wait_vm(int n) {
#define CVX_WV(N) else if (n == N) wait_vmcnt<N>();
  if (n <= 0) wait_vmcnt<0>();
  CVX_WV(1) CVX_WV(2) CVX_WV(3) CVX_WV(4) CVX_WV(5) CVX_WV(6) CVX_WV(7) CVX_WV(8) CVX_WV(9) CVX_WV(10) CVX_WV(11) CVX_WV(12)
  CVX_WV(13) CVX_WV(14) CVX_WV(15) CVX_WV(16) CVX_WV(17) CVX_WV(18) CVX_WV(19) CVX_WV(20) CVX_WV(21) CVX_WV(22) CVX_WV(23) CVX_WV(24)
  else wait_vmcnt<24>();
#undef CVX_WV
}
__device__ __forceinline__ void k3_clk(const K3Plan& a, int slot) {
  if (a.clk && threadIdx.x == 0) a.clk[(long long)blockIdx.x * 8 + slot] = wall_clock64();
}

// one chunk's DMA geometry (wave-uniform)
struct K3Chunk {
  unsigned d_org, x_org;
  int d_rhi, d_chi, x_rlo, x_rhi, x_clo, x_chi;
  unsigned char* sb;
};

template <class C>
__global__ __launch_bounds__(64 * C::NW) void conv_wgrad_k3_kernel(const WgradParams p, const K3Plan a) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (and has no __amdgpu_buffer_rsrc_t)
  constexpr int CO_TW = C::CO_TW, JW = C::JW, NW = C::NW, PD = C::PD, PX = C::PX;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // workgroups of one pixel split read the same chunks: their ids share id % 8, i.e. one XCD's L2
  const int G = a.n_co_blk * a.n_ci_blk;
  const int qq = blockIdx.x / (8 * G), r8 = blockIdx.x - qq * (8 * G);
  const int bz = qq * 8 + (r8 & 7);
  if (bz >= p.nsplit) return;
  k3_clk(a, 0);
  const int g = r8 >> 3;
  const int cob = g % a.n_co_blk, cib = g / a.n_co_blk;
  const int co0 = cob * (16 * C::CO_T), ci0 = cib * (16 * C::CI_T);
  const int wn = wave % C::WN, wc = (wave / C::WN) % C::WC, wk = wave / (C::WN * C::WC);
  const int Wp = a.Wp, W = a.W;

  const __amdgpu_buffer_rsrc_t rsrc_d = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<half_t*>(p.dy), (short)0, (int)(unsigned)std::min<long long>((long long)p.B * p.dy_bstride * 2, 0xffffffffLL), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<half_t*>(p.x), (short)0, (int)(unsigned)std::min<long long>((long long)p.B * p.x_bstride * 2, 0xffffffffLL), 0x00020000);

  // ---- per-lane DMA table.  Piece q = k * NW + wave of a slot (dy pieces first, then x): byte offset of the lane's 16 bytes relative to the
  // chunk's origin, and row | padded column << 8 for the per-chunk bounds (0xffff: never in range) ----
  const int total_p = a.d_pieces + a.x_pieces;
  const int PWw = (total_p - wave + NW - 1) / NW;  // pieces of this wave per chunk
  unsigned prel[K3_MAXP], ptag[K3_MAXP];
#pragma unroll
  for (int k = 0; k < K3_MAXP; ++k) {
    prel[k] = 0;
    ptag[k] = 0xffffu;
    const int piece = k * NW + wave;
    if (piece >= total_p) continue;
    if (piece < a.d_pieces) {
      constexpr int upp = C::CO_T * 2;  // 16-byte units per pixel
      const int u = piece * 64 + lane;
      const int t = u / upp, v = u - t * upp;
      const int f = (t >> C::D_SH) & C::D_MASK;
      const int ch = co0 + ((((v >> 1) ^ f) << 1) | (v & 1)) * 8;
      const int row = (int)__umulhi((unsigned)t, a.m_wp), colp = t - row * Wp;
      if (ch < p.Cout && row < a.R && colp >= 1 && colp <= a.Wc) {
        prel[k] = (unsigned)(((row * W + colp - 1) * p.dy_ld + ch) * 2);
        ptag[k] = (unsigned)(row | (colp << 8));
      }
    } else {
      constexpr int upp = C::CI_T * 2;
      const int u = (piece - a.d_pieces) * 64 + lane;
      const int t = u / upp, v = u - t * upp;
      const int f = (t >> C::X_SH) & C::X_MASK;
      const int ch = ci0 + ((((v >> 1) ^ f) << 1) | (v & 1)) * 8;
      if (a.s2) {
        // Stride 2: input pixel (2 r + kh - 1, 2 c + kw - 1) of output pixel (r, c).  The x image holds the input's four phase planes
        // (row parity py, column parity px), each in the dy image's padded linear space: plane pixel (rr, colp) = input pixel
        // (2 (y0 - 1 + rr) + py, 2 (c0 - 2 + colp) + px).  Inside a plane a tap is a constant offset again (xB below), and a K-step is 32
        // consecutive plane pixels.  Tags: row / column relative to input pixel (2 y0 - 2, 2 c0 - 4).
        const int pi = t / a.PP, tt = t - pi * a.PP;
        const int rr = (int)__umulhi((unsigned)tt, a.m_wp), colp = tt - rr * Wp;
        const int py = pi >> 1, px = pi & 1;
        if (ch < p.Cin && pi < 4 && rr <= a.R && colp >= 1 && colp <= a.Wc + 1) {
          prel[k] = (unsigned)((((2 * rr - 2 + py) * a.IW + (2 * colp - 4 + px)) * p.x_ld + ch) * 2);
          ptag[k] = (unsigned)((2 * rr + py) | ((2 * colp + px) << 8));
        }
        continue;
      }
      const int tt = t - 1;  // (pixel 0 of the x image is slack: tap (-1, -1) of padded pixel 0)
      const int rr = tt >= 0 ? (int)__umulhi((unsigned)tt, a.m_wp) : 0, colp = tt - rr * Wp;
      if (ch < p.Cin && tt >= 0 && rr < a.R + 2) {
        prel[k] = (unsigned)(((rr * W + colp) * p.x_ld + ch) * 2);
        ptag[k] = (unsigned)(rr | (colp << 8));
      }
    }
  }

  // ---- this split's chunks: unit -> (image, row block, column block) ----
  const int u_begin = std::min(a.total_units, bz * a.units_per_split);
  const int u_end = std::min(a.total_units, u_begin + a.units_per_split);
  const int nch = u_end - u_begin;
  int ib = u_begin / a.upi;
  int irem = u_begin - ib * a.upi;
  int irb = irem / a.ncb, icb = irem - irb * a.ncb;
  int islot = 0;
  // geometry of the next chunk to be requested; advances the cursor
  auto next_chunk = [&]() __attribute__((always_inline)) -> K3Chunk {
    K3Chunk q;
    const int y0 = irb * a.R, c0 = icb * a.Wc;
    q.d_org = (unsigned)(((long long)ib * p.dy_bstride + ((long long)y0 * W + c0) * p.dy_ld) * 2);
    q.x_org = (unsigned)(((long long)ib * p.x_bstride + ((long long)(y0 - 1) * W + c0 - 1) * p.x_ld) * 2);  // (may wrap below zero: valid lanes add it back)
    q.d_rhi = std::min(a.R, a.H - y0);                  // dy rows [0, d_rhi), padded columns [1, d_chi)
    q.d_chi = std::min(a.Wc, W - c0) + 1;
    q.x_rlo = y0 == 0 ? 1 : 0;
    q.x_rhi = std::min(a.R + 2, a.H - y0 + 1);
    q.x_clo = c0 == 0 ? 1 : 0;
    q.x_chi = std::min(a.Wc + 2, W - c0 + 1);
    if (a.s2) {
      q.x_org = (unsigned)(((long long)ib * p.x_bstride + ((long long)(2 * y0) * a.IW + 2 * c0) * p.x_ld) * 2);
      q.x_rlo = std::max(0, 2 - 2 * y0);
      q.x_rhi = std::min(2 * a.R + 2, a.IH - 2 * y0 + 2);
      q.x_clo = std::max(0, 4 - 2 * c0);
      q.x_chi = std::min(2 * a.Wc + 4, a.IW - 2 * c0 + 4);
    }
    if (++icb == a.ncb) {
      icb = 0;
      if (++irb * a.R >= a.H) {
        irb = 0;
        ++ib;
      }
    }
    q.sb = smem + islot * a.slot_bytes;
    islot = islot + 1 == a.nslot ? 0 : islot + 1;
    return q;
  };
  // piece k (compile time) of this wave for chunk q
#define CVX_K3_DMA(q, k)                                                                                                   \
  if ((k) < PWw) {                                                                                                          \
    const int piece = (k) * NW + wave;                                                                                     \
    const int row = (int)(ptag[k] & 0xffu), colp = (int)(ptag[k] >> 8);                                                     \
    if (piece < a.d_pieces) {                                                                                               \
      const unsigned vo = (row < (q).d_rhi && colp < (q).d_chi) ? (q).d_org + prel[k] : 0xffffffffu;                        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_d, (lds_ptr_t)((q).sb + piece * 1024), 16, vo, 0, 0, 0);                \
    } else {                                                                                                                \
      const bool ok = (unsigned)(row - (q).x_rlo) < (unsigned)((q).x_rhi - (q).x_rlo) && (unsigned)(colp - (q).x_clo) < (unsigned)((q).x_chi - (q).x_clo); \
      const unsigned vo = ok ? (q).x_org + prel[k] : 0xffffffffu;                                                           \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)((q).sb + piece * 1024), 16, vo, 0, 0, 0);                \
    }                                                                                                                       \
  }

  // D[m = ci][n = co]: x is the A operand, so a lane ends with FOUR CONSECUTIVE ci of one co -- 16-byte slab stores
  f4 acc[CO_TW][JW];
#pragma unroll
  for (int i = 0; i < CO_TW; ++i)
#pragma unroll
    for (int j = 0; j < JW; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  if (nch > 0) {
    const int npre = std::min(nch, a.nslot - 1);
    for (int s = 0; s < npre; ++s) {
      const K3Chunk q = next_chunk();
#pragma unroll
      for (int k = 0; k < K3_MAXP; ++k) CVX_K3_DMA(q, k)
    }
    k3_clk(a, 1);

    // ---- transposed-read addresses inside a slot.  16-lane group kg = lane >> 4, lane 4 r + c of it: pixel 4 kg + r (+ 16 for the second
    // read) of the K-step, channels 4 c .. 4 c + 3 of the 16-channel tile; slot = tile ^ f(pixel) ----
    const int kg = lane >> 4, rr = (lane & 15) >> 2, cc = lane & 3;
    const unsigned lds0 = k3_lds32(smem);
    const unsigned x_off = (unsigned)a.d_pieces * 1024u;
    constexpr int DSTEP = 32 * PD * C::WK, XSTEP = 32 * PX * C::WK;  // bytes between two K-steps of one wave
    unsigned dA[CO_TW], xB[JW];
    {
      const int pix = 4 * kg + rr;
      const int f = (pix >> C::D_SH) & C::D_MASK;
#pragma unroll
      for (int i = 0; i < CO_TW; ++i) dA[i] = lds0 + (unsigned)(32 * PD * wk + pix * PD + (((wc * CO_TW + i) ^ f) << 5) + cc * 8);
    }
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
      int jt = jj * C::WN + wn;  // column tile -> (tap, ci tile); the surplus ones repeat the last valid tile (computed, not stored)
      jt = jt < C::NJ ? jt : C::NJ - 1;
      const int tap = jt / C::CI_T, cit = jt - tap * C::CI_T;
      int shift = (tap / 3) * Wp + (tap - (tap / 3) * 3);  // (dh + 1) * Wp + dw + 1
      if (a.s2) {  // plane of the tap's parities, one row down for kh >= 1, one column right for kw >= 1
        const int kh = tap / 3, kw = tap - kh * 3;
        shift = ((kh == 1 ? 0 : 2) + (kw == 1 ? 0 : 1)) * a.PP + (kh == 0 ? 0 : Wp) + (kw == 0 ? 0 : 1);
      }
      const int pix = 4 * kg + rr + shift;
      const int f = (pix >> C::X_SH) & C::X_MASK;
      xB[jj] = lds0 + x_off + (unsigned)(32 * PX * wk + pix * PX + ((cit ^ f) << 5) + cc * 8);
    }
    // everything above is needed only after the first wait: keep it above it (conv_tile_kernel.inc.h)
#pragma unroll
    for (int i = 0; i < CO_TW; ++i) asm volatile("" ::"v"(dA[i]));
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) asm volatile("" ::"v"(xB[jj]));

    s4 fal[2][CO_TW], fah[2][CO_TW], fbl[2][JW], fbh[2][JW];
    constexpr int NRD = 2 * (CO_TW + JW), NMF = CO_TW * JW;
    constexpr int RPM = (NRD + NMF - 1) / NMF;  // fragment reads issued behind every MFMA
    constexpr int PPS = K3_MAXP / K3_KSW;       // DMA pieces requested during every K-step
    // read number r of the fragment set of K-step KS (compile time) into set SET
#define CVX_K3_RD(SET, KS, r)                                                                       \
  {                                                                                                 \
    if ((r) < 2 * CO_TW) {                                                                          \
      if (((r) & 1) == 0) fal[SET][(r) >> 1] = k3_tr<(KS) * DSTEP>(dcur[(r) >> 1]);                 \
      else fah[SET][(r) >> 1] = k3_tr<(KS) * DSTEP + 16 * PD>(dcur[(r) >> 1]);                      \
    } else {                                                                                        \
      if (((r) & 1) == 0) fbl[SET][((r) >> 1) - CO_TW] = k3_tr<(KS) * XSTEP>(xcur[((r) >> 1) - CO_TW]);       \
      else fbh[SET][((r) >> 1) - CO_TW] = k3_tr<(KS) * XSTEP + 16 * PX>(xcur[((r) >> 1) - CO_TW]);            \
    }                                                                                               \
  }
    // K-step KS: the MFMAs of set KS & 1; behind them ride the fragment reads of K-step KS + 1 (set 1 - (KS & 1)) and, at the thirds of the
    // block, the next chunk's DMA pieces PPS * KS ... -- nothing but MFMAs, LDS reads and DMA issue in the loop
#define CVX_K3_STEP(KS)                                                                             \
  {                                                                                                 \
    _Pragma("unroll") for (int m = 0; m < NMF; ++m) {                                               \
      const int jj = m / CO_TW, i = m - jj * CO_TW;                                                 \
      if (!(dbg & 2)) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k3_join(fbl[(KS) & 1][jj], fbh[(KS) & 1][jj]), k3_join(fal[(KS) & 1][i], fah[(KS) & 1][i]), acc[i][jj], 0, 0, 0); \
      if ((KS) + 1 < K3_KSW && !(dbg & 4)) {                                                                      \
        _Pragma("unroll") for (int q = 0; q < RPM; ++q)                                             \
          if (m * RPM + q < NRD) CVX_K3_RD(1 - ((KS) & 1), (KS) + 1, m * RPM + q)                    \
      }                                                                                             \
      if (more && !(dbg & 1)) {                                                                     \
        _Pragma("unroll") for (int q = 0; q < PPS; ++q)                                             \
          if (m == ((q + 1) * NMF) / (PPS + 1)) CVX_K3_DMA(nq, PPS * (KS) + q)                       \
      }                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                            \
    }                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                              \
    __builtin_amdgcn_sched_barrier(0);                                                              \
  }
    int issued = npre;
#ifdef CVX_K3_ABL  // compile-time ablation (one-off experiment builds; a run-time test per MFMA costs more than what it removes): results WRONG
    constexpr int dbg = CVX_K3_ABL;
#else
    constexpr int dbg = 0;
#endif
    unsigned so = 0;  // byte offset of the chunk's slot
    for (int c = 0; c < nch; ++c) {
      k3_wait_vm((issued - c - 1) * PWw);  // this wave's pieces of chunk c have landed
      workgroup_barrier();                 // ... everybody's; and every wave is done with chunk c - 1: its slot takes chunk c + NS - 1
      if (c == 0) k3_clk(a, 2);
      const bool more = issued < nch;
      K3Chunk nq;
      memset(&nq, 0, sizeof(nq));
      if (more) {
        nq = next_chunk();
        ++issued;
      }
      // this chunk's fragment addresses
      unsigned dcur[CO_TW], xcur[JW];
#pragma unroll
      for (int i = 0; i < CO_TW; ++i) dcur[i] = dA[i] + so;
#pragma unroll
      for (int jj = 0; jj < JW; ++jj) xcur[jj] = xB[jj] + so;
      // fragments of this wave's first K-step
#pragma unroll
      for (int r = 0; r < NRD; ++r) CVX_K3_RD(0, 0, r)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      CVX_K3_STEP(0)
      CVX_K3_STEP(1)
      CVX_K3_STEP(2)
      CVX_K3_STEP(3)
      so = so + (unsigned)a.slot_bytes == (unsigned)(a.nslot * a.slot_bytes) ? 0u : so + (unsigned)a.slot_bytes;
    }
#undef CVX_K3_RD
#undef CVX_K3_STEP
    k3_clk(a, 3);
  }
#undef CVX_K3_DMA

  // ---- K-step waves fold into wk == 0 through the (now idle) ring ----
  if (C::WK > 1) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    const int wslot = wc * C::WN + wn;
    if (wk > 0) {
#pragma unroll
      for (int i = 0; i < CO_TW; ++i)
#pragma unroll
        for (int jj = 0; jj < JW; ++jj)
          *reinterpret_cast<f4*>(red + ((((wk - 1) * (C::WC * C::WN) + wslot) * CO_TW + i) * JW + jj) * 256 + lane * 4) = acc[i][jj];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int w = 1; w < C::WK; ++w) {
#pragma unroll
        for (int i = 0; i < CO_TW; ++i)
#pragma unroll
          for (int jj = 0; jj < JW; ++jj) {
            const f4 o = *reinterpret_cast<const f4*>(red + ((((w - 1) * (C::WC * C::WN) + wslot) * CO_TW + i) * JW + jj) * 256 + lane * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][jj][r] += o[r];
          }
      }
    }
  }
  if (wk == 0) {
    // ---- the split's slab: lane holds output channel co = .. + (lane & 15), columns ci = .. + 4 * (lane >> 4) + 0 .. 3 ----
    const int Jtot = p.ntaps * p.cin_pad16;
    float* slab = p.slabs + (long long)bz * p.Cout * Jtot;
    const int lg = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
      const int jt = jj * C::WN + wn;
      if (jt >= C::NJ) continue;
      const int tap = jt / C::CI_T, cit = jt - tap * C::CI_T;
      const int ci = ci0 + cit * 16 + fq * 4;
      if (ci >= p.cin_pad16) continue;
      const int j = tap * p.cin_pad16 + ci;
#pragma unroll
      for (int i = 0; i < CO_TW; ++i) {
        const int co = co0 + (wc * CO_TW + i) * 16 + lg;
        if (co < p.Cout) *reinterpret_cast<f4*>(slab + (long long)co * Jtot + j) = acc[i][jj];
      }
    }
  }
  k3_clk(a, 4);
#endif
}

// ---------------------------------------------------------------- host side ----------------------------------------------------------------

//                 CO_T CI_T WC WN WK
typedef K3Cfg<1, 1, 1, 3, 2> CfgA;  // 16 -> 16            : 9 column tiles over 3 waves, 2 K-step groups          (6 waves, 1 x 3 tiles per wave)
typedef K3Cfg<2, 2, 1, 4, 2> CfgB;  // 32 -> 32            : 18 column tiles over 4 waves, 2 K-step groups         (2 x 5)
typedef K3Cfg<4, 4, 2, 4, 1> CfgC;  // 64 -> 64 blocks     : 2 co halves x 36 column tiles over 4 waves            (2 x 9)
typedef K3Cfg<3, 5, 1, 8, 1> CfgD;  // 80 -> 80            : co blocks of 48 (the 5 x 6 tile of the whole layer spills), 45 column tiles over 8 waves (3 x 6)
typedef K3Cfg<3, 4, 1, 8, 1> CfgE;  // 48-channel co blocks x 64-channel ci blocks (64 / 128 / 256 -> 144)          (3 x 5)
typedef K3Cfg<2, 4, 1, 8, 1> CfgF;  // 32-channel co blocks x 64-channel ci blocks (64 -> 32 ...)                   (2 x 5)
typedef K3Cfg<2, 1, 1, 4, 2> CfgG;  // 16 -> 32 (stride 2: model.1)                                                     (2 x 3)
typedef K3Cfg<4, 2, 2, 4, 1> CfgH;  // 32 -> 64 (stride 2: model.3)                                                     (2 x 5)

struct K3Shape {
  int co_t, ci_t, wk, cfg, nw;
};

// which configuration serves a layer (cfg < 0: none)
K3Shape k3_pick(const WgradParams& p) {
  const int ct = (p.Cout + 15) / 16, it = p.cin_pad16 / 16;
  if (p.stride == 2) {  // the large early layers only: four phase planes of a 64-channel x image do not fit a slot
    static const bool s2_off = cvx_tune_set("CVX_K3_NO_S2");
    if (s2_off) return {0, 0, 0, -1, 0};
    if (ct == 2 && it == 1) return {2, 1, 2, 6, 8};
    if (ct == 4 && it == 2) return {4, 2, 1, 7, 8};
    return {0, 0, 0, -1, 0};
  }
  if (ct == 1 && it == 1) return {1, 1, 2, 0, 6};
  if (ct == 2 && it == 2) return {2, 2, 2, 1, 8};
  if (ct == 5 && it == 5) return {3, 5, 1, 3, 8};
  if (ct % 4 == 0 && it % 4 == 0) return {4, 4, 1, 2, 8};
  if (ct % 3 == 0 && it % 4 == 0) return {3, 4, 1, 4, 8};
  if (ct == 2 && it % 4 == 0) return {2, 4, 1, 5, 8};
  return {0, 0, 0, -1, 0};
}

unsigned k3_magic(int d) { return (unsigned)((0x100000000ULL + (unsigned long long)d - 1) / (unsigned long long)d); }

bool k3_shape_ok(const WgradParams& p) {
  if (!(p.std3x3 && p.ntaps == 9)) return false;
  if (!((p.stride == 1 && p.IH == p.OH && p.IW == p.OW) || (p.stride == 2 && p.OH == (p.IH - 1) / 2 + 1 && p.OW == (p.IW - 1) / 2 + 1))) return false;
  if (p.Cin % 8 || p.Cout % 8 || p.x_ld % 8 || p.dy_ld % 8 || p.Cin < 8) return false;
  if ((long long)p.B * p.x_bstride * 2 >= (1LL << 32) || (long long)p.B * p.dy_bstride * 2 >= (1LL << 32)) return false;
  if (p.OW < 4 || p.OH > 250) return false;
  return k3_pick(p).cfg >= 0;
}

// The plan of a launch with `nsplit` pixel splits (nsplit <= 0: the planner's own choice, returned in *nsplit_out).
bool k3_plan(const WgradParams& p, int nsplit, K3Plan* out, K3Shape* shape, int* nsplit_out, bool wide = false) {
  if (!k3_shape_ok(p)) return false;
  const K3Shape sh = k3_pick(p);
  K3Plan a;
  memset(&a, 0, sizeof(a));
  static const int force_r = cvx_tune_int("CVX_K3_R", 0), force_ns = cvx_tune_int("CVX_K3_NSLOT", 0), force_wc = cvx_tune_int("CVX_K3_WC", 0);
  static const int lds_kb = cvx_tune_int("CVX_K3_LDS_KB", 152);
  const int budget = lds_kb * 1024;
  const int PD = sh.co_t * 32, PX = sh.ci_t * 32, NW = sh.nw;
  const bool s2 = p.stride == 2;
  a.s2 = s2 ? 1 : 0;
  a.IW = p.IW;
  a.IH = p.IH;
  const int QD = 32 * K3_KSW * sh.wk;  // K3Cfg::QD
  a.W = p.OW;
  a.H = p.OH;
  a.n_co_blk = cvx_cdiv(p.Cout, 16 * sh.co_t);
  a.n_ci_blk = cvx_cdiv(p.cin_pad16, 16 * sh.ci_t);
  bool found = false;
  double best = 0;
  int bR = 0, bWc = 0, bNS = 0;
  auto geom = [&](int R, int Wc) {
    a.R = R;
    a.Wc = Wc;
    a.Wp = Wc + 2;
    a.ncb = cvx_cdiv(p.OW, Wc);
    a.Tx = QD + 2 * a.Wp + 2;
    if (s2) {  // every plane covers the pixels a K-step can reach: QD + one row + one column (all of it is written by every chunk's DMA)
      a.PP = QD + a.Wp + 2;
      a.Tx = 4 * a.PP;
    }
    a.d_pieces = cvx_cdiv((long long)QD * PD, 1024);
    a.x_pieces = cvx_cdiv((long long)a.Tx * PX, 1024);
    a.slot_bytes = (a.d_pieces + a.x_pieces) * 1024;
    a.upi = cvx_cdiv(p.OH, R) * a.ncb;
    a.total_units = p.B * a.upi;
  };
  // a chunk is R rows of a column block of width Wc with R * (Wc + 2) <= QD padded pixels: the cut that wastes the fewest K-steps and
  // re-reads the least halo
  for (int parts = 1; parts <= 16; ++parts) {
    const int Wc = cvx_cdiv(p.OW, parts);
    if (Wc < 6) break;
    if (Wc > (s2 ? 124 : 250) || Wc + 2 > QD || (force_wc && Wc != force_wc)) continue;
    const int R = std::min(p.OH, QD / (Wc + 2));
    if (R < 1 || (force_r && R != force_r)) continue;
    geom(R, Wc);
    if ((long long)((R + 2) * p.OW) * p.x_ld * 2 >= (1LL << 31) || (long long)(R * p.OW) * p.dy_ld * 2 >= (1LL << 31)) continue;
    if (s2 && ((long long)((2 * R + 4) * p.IW) * p.x_ld * 2 >= (1LL << 31) || 2 * R + 2 > 255)) continue;
    if (a.d_pieces + a.x_pieces > K3_MAXP * NW) continue;
    const int pw = cvx_cdiv(a.d_pieces + a.x_pieces, NW);
    int ns = 0;
    for (int q = 4; q >= 2; --q) {
      if (force_ns && q != force_ns) continue;
      if (q * a.slot_bytes <= budget && (q - 2) * pw <= 24) {
        ns = q;
        break;
      }
    }
    if (!ns) continue;
    const double useful = (double)p.OH * p.OW / a.upi;  // useful pixels per chunk, averaged over an image
    const double kpad = (double)QD / useful;
    const double fill = (double)a.slot_bytes / (useful * (PD + PX));
    double cost = (0.65 * kpad + 0.35 * fill) * (ns >= 3 ? 1.0 : 1.12);
    // stride 2: what the cut costs is the input halo it re-reads ((2 R + 2) x (2 Wc + 2) input pixels for 2 R x 2 Wc) beside the padded K-steps
    if (s2) cost = 0.5 * kpad + 0.5 * ((2.0 * R + 2) * (2.0 * Wc + 2)) / (4.0 * R * Wc);
    if (!found || cost < best) {
      found = true;
      best = cost;
      bR = R;
      bWc = Wc;
      bNS = ns;
    }
  }
  if (!found) return false;
  geom(bR, bWc);
  a.nslot = bNS;
  a.m_wp = k3_magic(a.Wp);
  const int fold = sh.wk > 1 ? (sh.wk - 1) * (NW / sh.wk) * 16 * 1024 : 0;  // (generous: CO_TW * JW <= 16 tiles per wave)
  a.lds_bytes = std::max(a.nslot * a.slot_bytes, fold);
  if (a.lds_bytes > 160 * 1024) return false;
  // ---- pixel splits: as many workgroups as the layer's work needs, no more (they hold their CUs for their whole life) ----
  const int G = a.n_co_blk * a.n_ci_blk;
  int ns = nsplit;
  if (ns <= 0) {
    static const int one_round = cvx_tune_int("CVX_K3_ONE_ROUND", 1);
    static const int mflop_wg = cvx_tune_int("CVX_K3_MFLOP", 60), kb_wg = cvx_tune_int("CVX_K3_KB", 576), wg_max = cvx_tune_int("CVX_K3_WGMAX", 96),
                     wg_min = cvx_tune_int("CVX_K3_WGMIN", 32);
    const double flops = 2.0 * p.B * p.OH * p.OW * 9.0 * p.Cin * p.Cout, bytes = 2.0 * p.B * p.OH * p.OW * ((s2 ? 4.0 : 1.0) * p.Cin + p.Cout);
    int wgs = (int)std::max(flops / (mflop_wg * 1e6), bytes / (kb_wg * 1024.0));
    wgs = std::max(wg_min, std::min(one_round ? wg_max : 4096, wgs));
    // the last layers of the backward pass: nothing is left on the main stream to disturb, the launch is the tail of the step -- the whole chip
    if (wide) wgs = std::max(wgs, std::min(240, (int)std::max(flops / 15e6, bytes / (128 * 1024.0))));
    ns = std::max(1, wgs / G);
    ns = std::min(ns, std::max(1, a.total_units / 2));  // at least two chunks per workgroup
    // one round: a workgroup takes a whole CU, and the launch rounds the split count up to a multiple of 8 (XCD-aware ids)
    while (one_round && ns > 8 && (ns + 7) / 8 * 8 * G > 256) ns -= 1;
  }
  ns = std::max(1, std::min(ns, a.total_units));
  a.units_per_split = cvx_cdiv(a.total_units, ns);
  ns = cvx_cdiv(a.total_units, a.units_per_split);  // (no empty splits)
  if (nsplit > 0) ns = nsplit;                         // a caller's count stands: surplus splits store zero slabs
  *out = a;
  *shape = sh;
  if (nsplit_out) *nsplit_out = ns;
  return true;
}

template <class C>
int launch_k3(const WgradParams& p, const K3Plan& a, hipStream_t st) {
  static unsigned long long optin_mask = 0;
  CVX_TRY(cvx_lds_optin((const void*)conv_wgrad_k3_kernel<C>, 160 * 1024, &optin_mask));
  const int G = a.n_co_blk * a.n_ci_blk;
  const int ns8 = (p.nsplit + 7) / 8 * 8;
  hipLaunchKernelGGL((conv_wgrad_k3_kernel<C>), dim3(G * ns8), dim3(64 * C::NW), a.lds_bytes, st, p, a);
  return 0;
}

}  // namespace

bool cvx_conv_wgrad_k3_supported(const WgradParams& p) {
  static const bool off = cvx_tune_set("CVX_NO_WGRAD_K3");
  if (off) return false;
  static const int max_cc = cvx_tune_int("CVX_K3_MAX_CC", 256 * 144);  // wider layers: the GEMM-shaped kernel (conv_wgrad_gemm.hip)
  if ((long long)p.Cin * p.Cout > max_cc) return false;
  K3Plan a;
  K3Shape sh;
  int ns;
  return k3_plan(p, 0, &a, &sh, &ns);
}

// the planner's pixel-split count for a layer (the engine sizes the layer's slabs with it)
int cvx_conv_wgrad_k3_nsplit(const WgradParams& p, bool wide) {
  K3Plan a;
  K3Shape sh;
  int ns = 1;
  if (!k3_plan(p, 0, &a, &sh, &ns, wide)) return 1;
  return ns;
}

int cvx_conv_wgrad_k3_launch(const WgradParams& p, hipStream_t st) {
  K3Plan a;
  K3Shape sh;
  int ns;
  CVX_CHECK(k3_plan(p, p.nsplit, &a, &sh, &ns), "wgrad k3: unsupported shape");
  a.units_per_split = cvx_cdiv(a.total_units, p.nsplit);
  a.clk = g_cvx_clk;
  switch (sh.cfg) {
    case 0: return launch_k3<CfgA>(p, a, st);
    case 1: return launch_k3<CfgB>(p, a, st);
    case 2: return launch_k3<CfgC>(p, a, st);
    case 3: return launch_k3<CfgD>(p, a, st);
    case 4: return launch_k3<CfgE>(p, a, st);
    case 5: return launch_k3<CfgF>(p, a, st);
    case 6: return launch_k3<CfgG>(p, a, st);
    case 7: return launch_k3<CfgH>(p, a, st);
  }
  CVX_FAIL("wgrad k3: no kernel for the planned tile");
}
