// Streaming weight gradient for the NARROW layers (gfx950) -- round 5.
//
//   dW[co][tap][ci] = sum over pixels  dy[p][co] * x[p + tap][ci]         (3x3 / stride 1 / pad 1, or 1x1 / stride 1)
//
// The narrow layers of YOLOv8-n (16 ... 80 channels at 160^2 ... 20^2, the 1x1 convs of the C2f blocks) have a tiny result (4 KB ... 230 KB)
// and operands of 13 ... 130 MB: their weight gradient is a STREAMING REDUCTION.  The two older kernels (conv_wgrad.hip, conv_wgrad_halo.hip)
// stage one tile through registers into a single LDS buffer with two barriers per tile: one tile in flight per workgroup, 0.04 ... 0.25 of
// the layer's HBM time (profiles/r04_op_profile.txt: 160^2 32 -> 32 1x1 at 612 GB/s).  This kernel is built like a copy kernel instead:
//
//   * a workgroup owns 16 * CO_T output channels x (a block of input channels) x (all nine taps, or one kernel row of three) and walks a
//     contiguous range of CHUNKS (R output rows of one image for 3x3, Qd consecutive pixels for 1x1) -- the pixel split;
//   * both operands of a chunk are DMA'd into one slot of an LDS ring (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction) as they
//     lie in memory, NS - 1 chunks in flight, ONE barrier per chunk.  3x3: the dy rows and the x rows (with halo) are laid out in a
//     PADDED LINEAR pixel space of pitch W + 2, so that tap (dh, dw) is the constant offset (dh + 1) * (W + 2) + dw + 1 between the two
//     images -- a K-step is 32 consecutive padded pixels, whatever the map width (20, 40, 80, 160 all waste only the two gutter columns,
//     whose dy is zero).  Everything that must read as zero -- gutters, rows outside the image, channels past Cin / Cout, the tail of the
//     last chunk -- is an out-of-range DMA lane: the buffer descriptor's bounds check writes the zeros;
//   * the reduction index (pixel) is the slow index of both images, so the MFMA fragments come from the transposing LDS read
//     ds_read_b64_tr_b16; a K-step's 16-lane group kg and read j take pixels 16 j + 4 kg ... + 3: the eight pixel rows one 32-lane half
//     touches are CONSECUTIVE pixels, and 32-byte channel slots are XOR-swizzled by the pixel on the global side (power-of-two pitches),
//     so every transposed read is bank-conflict free;
//   * v_mfma_f32_16x16x32_f16, dy as the A operand: the waves split the (tap, ci tile) column tiles -- and, where there are fewer column
//     tiles than waves (1x1 layers), the K-steps, folded through LDS at the end;
//   * each pixel split stores its fp32 slab (plain stores, deterministic), summed by cvx_reduce_slabs as before.  The split count is the
//     planner's (cvx_conv_wgrad_stream_nsplit): enough workgroups for two per CU, slabs capped at a fraction of the operand bytes.
//
// Roofline: HBM (each operand byte once) -- per layer 2 * M * (Cin + Cout) bytes.  Reference: autograd's weight gradient of nn.Conv2d in
// core/models/yolov8/modules.py:19-33 (Conv), :124-135 (Bottleneck), :189-207 (C2f), :407-455 (Detect).
#include <algorithm>
#include <cstring>

#include "conv_tile_common.h"

namespace {
using namespace cvx_tile;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int WS_MAXP = 16;   // DMA pieces (1 KiB) per wave and chunk, dy + x
constexpr int WS_INVALID_TAG = 0xfff;

struct WsPlan {
  int taps;            // taps per WORKGROUP: 9 (all), 3 (one kernel row), 1 (pointwise)
  int tap_blks;        // 1, or 3 when the kernel rows are split over workgroups
  int W, H, Wp;        // 3x3: map size, padded pitch W + 2.  1x1: W = H = 0
  int R;               // 3x3: output rows per chunk
  int Qd, KS, Tx;      // dy pixels per chunk (multiple of 64 * WK), K-steps, x pixels of the chunk tile
  int COB, CIB;        // channels per pixel of the two LDS images (multiples of 16)
  int d_sh, d_mask, x_sh, x_mask;  // slot swizzle f(t) = (t >> sh) & mask
  int d_pieces, x_pieces, dPW, xPW;
  int x_off, slot_bytes, nslot;
  int upi;             // 3x3: chunks per image
  int total_units, units_per_split;
  int n_co_blk, n_ci_blk;
  int CI_T, NJ;        // ci tiles of the x image, column tiles per workgroup
  int WN, WK;          // waves = WN (column tiles) x WK (K-steps)
  int lds_bytes;
  unsigned m_dupp, m_xupp, m_wp, m_cit;  // ceil(2^32 / d) for d = COB / 8, CIB / 8, Wp, CI_T (0: d == 1)
};

// floor(n / d) by the plan's magic number (d >= 2, n * d < 2^32; magic 0: d == 1)
__device__ __forceinline__ int ws_div(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }

__device__ __forceinline__ unsigned lds_addr32(const void* p) { return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p; }

__device__ __forceinline__ h8 tr_frag2(unsigned a0, unsigned a1) {
  s4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
  const s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(h8, v);
}

// waits until at most n (wave-uniform, 0 .. 32) vector-memory operations are outstanding (compare chains, not a switch: conv_tile.h)
__device__ __forceinline__ void ws_wait_vm(int n) {
#define CVX_WV(N) else if (n == N) wait_vmcnt<N>();
  if (n <= 0) wait_vmcnt<0>();
  else if (n <= 16) {
    if (n == 16) wait_vmcnt<16>();
    CVX_WV(1) CVX_WV(2) CVX_WV(3) CVX_WV(4) CVX_WV(5) CVX_WV(6) CVX_WV(7) CVX_WV(8) CVX_WV(9) CVX_WV(10) CVX_WV(11) CVX_WV(12) CVX_WV(13) CVX_WV(14) CVX_WV(15)
  } else {
    if (n >= 32) wait_vmcnt<32>();
    CVX_WV(17) CVX_WV(18) CVX_WV(19) CVX_WV(20) CVX_WV(21) CVX_WV(22) CVX_WV(23) CVX_WV(24) CVX_WV(25) CVX_WV(26) CVX_WV(27) CVX_WV(28) CVX_WV(29) CVX_WV(30) CVX_WV(31)
  }
#undef CVX_WV
}

template <int CO_T, int JW>
__global__ __launch_bounds__(256) void conv_wgrad_stream_kernel(const WgradParams p, const WsPlan a) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (and has no __amdgpu_buffer_rsrc_t)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // workgroups of one pixel split read the same chunks: their ids share id % 8, i.e. one XCD's L2 (conv_wgrad_halo.hip)
  const int G = a.n_co_blk * a.n_ci_blk * a.tap_blks;
  const int qq = blockIdx.x / (8 * G), r8 = blockIdx.x - qq * (8 * G);
  const int bz = qq * 8 + (r8 & 7);
  if (bz >= p.nsplit) return;
  int g = r8 >> 3;
  const int cob = g % a.n_co_blk;
  g /= a.n_co_blk;
  const int cib = g % a.n_ci_blk;
  const int tapb = g / a.n_ci_blk;  // kernel row of this workgroup when the rows are split
  const int co0 = cob * (16 * CO_T), ci0 = cib * a.CIB;
  const int wn = wave % a.WN, wk = wave / a.WN;
  const int NW = a.WN * a.WK;
  const bool k3 = a.W > 0;
  const int Wp = a.Wp, W = a.W;
  // the x image starts one padded pixel before the first tap's offset; with all nine taps in the workgroup it spans rows y0 - 1 .. y0 + R,
  // with one kernel row (dh = tapb - 1) rows y0 + dh .. y0 + dh + R - 1
  const int dh0 = a.tap_blks == 3 ? tapb - 1 : -1;          // image row of x-image row 0, relative to y0
  const int xrows = a.tap_blks == 3 ? a.R : a.R + 2;         // rows of the x image that can hold data

  const __amdgpu_buffer_rsrc_t rsrc_d = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<half_t*>(p.dy), (short)0, (int)(unsigned)std::min<long long>((long long)p.B * p.dy_bstride * 2, 0xffffffffLL), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<half_t*>(p.x), (short)0, (int)(unsigned)std::min<long long>((long long)p.B * p.x_bstride * 2, 0xffffffffLL), 0x00020000);

  // ---- per-lane DMA table: piece k of this wave -> (tag << 20) | byte offset relative to the chunk's origin ----
  unsigned pk[WS_MAXP];
  const int PW = a.dPW + a.xPW;
#pragma unroll
  for (int k = 0; k < WS_MAXP; ++k) {
    pk[k] = (unsigned)WS_INVALID_TAG << 20;
    if (k >= PW) continue;
    const bool isd = k < a.dPW;
    const int kk = isd ? k : k - a.dPW;
    const int np = isd ? a.d_pieces : a.x_pieces;
    int piece = kk * NW + wave;
    piece = piece < np ? piece : np - 1;  // surplus pieces repeat the last one: same bytes to the same place
    const int u = piece * 64 + lane;
    const int upp = (isd ? a.COB : a.CIB) >> 3;                      // 16-byte units per pixel
    const int t = ws_div(u, isd ? a.m_dupp : a.m_xupp), v = u - t * upp;
    const int f = isd ? ((t >> a.d_sh) & a.d_mask) : ((t >> a.x_sh) & a.x_mask);
    const int gch = ((((v >> 1) ^ f) << 1) | (v & 1)) * 8;             // channel (inside the block) this unit holds
    int tag = WS_INVALID_TAG;
    long long rel = 0;
    if (isd) {
      const bool cok = co0 + gch < p.Cout;
      if (k3) {
        const int row = ws_div(t, a.m_wp), colp = t - row * Wp;
        if (cok && row < a.R && colp >= 1 && colp <= W) {
          tag = row;
          rel = ((long long)(row * W + colp - 1) * p.dy_ld + co0 + gch) * 2;
        }
      } else if (cok && t < a.Qd) {
        tag = t;
        rel = ((long long)t * p.dy_ld + co0 + gch) * 2;
      }
    } else {
      const bool cok = ci0 + gch < p.Cin;
      if (k3) {
        const int tt = t - 1;
        const int rr = tt >= 0 ? ws_div(tt, a.m_wp) : 0, colp = tt - rr * Wp;
        if (cok && tt >= 0 && rr < xrows && colp >= 1 && colp <= W) {
          tag = rr;
          rel = ((long long)(rr * W + colp - 1) * p.x_ld + ci0 + gch) * 2;
        }
      } else if (cok && t < a.Qd) {
        tag = t;
        rel = ((long long)t * p.x_ld + ci0 + gch) * 2;
      }
    }
    pk[k] = ((unsigned)tag << 20) | (unsigned)rel;
  }

  // ---- this split's chunks ----
  const int u_begin = std::min(a.total_units, bz * a.units_per_split);
  const int u_end = std::min(a.total_units, u_begin + a.units_per_split);
  const int nch = u_end - u_begin;
  int iu = u_begin;                     // issue cursor
  int ib = iu / a.upi;                  // its image and chunk-in-image
  int irb = iu - ib * a.upi;
  int islot = 0;
  auto issue = [&]() __attribute__((always_inline)) {
    unsigned d_org, x_org;
    int d_hi, x_lo, x_hi;
    if (k3) {
      const int y0 = irb * a.R;
      d_org = (unsigned)(((long long)ib * p.dy_bstride + (long long)y0 * W * p.dy_ld) * 2);
      x_org = (unsigned)(((long long)ib * p.x_bstride + (long long)(y0 + dh0) * W * p.x_ld) * 2);  // (may wrap below zero: valid lanes add it back)
      d_hi = std::min(a.R, a.H - y0);
      x_lo = std::max(0, -(y0 + dh0));
      x_hi = std::min(xrows, a.H - y0 - dh0);
      if (++irb == a.upi) {
        irb = 0;
        ++ib;
      }
    } else {  // 1x1: Qd consecutive pixels of one image
      const int p0 = irb * a.Qd;
      d_org = (unsigned)(((long long)ib * p.dy_bstride + (long long)p0 * p.dy_ld) * 2);
      x_org = (unsigned)(((long long)ib * p.x_bstride + (long long)p0 * p.x_ld) * 2);
      d_hi = std::min(a.Qd, p.OH * p.OW - p0);
      x_lo = 0;
      x_hi = d_hi;
      if (++irb == a.upi) {
        irb = 0;
        ++ib;
      }
    }
    ++iu;
    unsigned char* sb = smem + islot * a.slot_bytes;
    islot = islot + 1 == a.nslot ? 0 : islot + 1;
#pragma unroll
    for (int k = 0; k < WS_MAXP; ++k) {
      if (k >= PW) continue;
      const bool isd = k < a.dPW;
      const int kk = isd ? k : k - a.dPW;
      const int np = isd ? a.d_pieces : a.x_pieces;
      int piece = kk * NW + wave;
      piece = piece < np ? piece : np - 1;
      const int tag = (int)(pk[k] >> 20);
      const unsigned rel = pk[k] & 0xfffffu;
      const int lo = isd ? 0 : x_lo, hi = isd ? d_hi : x_hi;
      const bool ok = (unsigned)(tag - lo) < (unsigned)(hi - lo);
      const unsigned vo = ok ? (isd ? d_org : x_org) + rel : 0xffffffffu;
      if (isd) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_d, (lds_ptr_t)(sb + piece * 1024), 16, vo, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(sb + a.x_off + piece * 1024), 16, vo, 0, 0, 0);
    }
  };

  f4 acc[CO_T][JW];
#pragma unroll
  for (int i = 0; i < CO_T; ++i)
#pragma unroll
    for (int j = 0; j < JW; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  // column tiles of this wave: jt = jj * WN + wn -> (tap, ci tile); the surplus ones repeat the last valid tile (computed, not stored)
  int jt_of[JW];
#pragma unroll
  for (int jj = 0; jj < JW; ++jj) jt_of[jj] = jj * a.WN + wn;

  if (nch > 0) {
    const int npre = std::min(nch, a.nslot - 1);
    for (int s = 0; s < npre; ++s) issue();

    // ---- transposed-read addresses inside a slot.  16-lane group kg = lane >> 4, lane 4 r + c of it: pixel 4 kg + r (+ 16 for the second
    // read) of the K-step, channels 4 c .. 4 c + 3 of the 16-channel tile; slot = tile ^ f(pixel) ----
    const int kg = lane >> 4, rr = (lane & 15) >> 2, cc = lane & 3;
    const unsigned lds0 = lds_addr32(smem);
    const int Pd = a.COB * 2, Px = a.CIB * 2;
    unsigned dA[CO_T], xB[JW];
    {
      const int pix = 4 * kg + rr;
      const int f = (pix >> a.d_sh) & a.d_mask;
#pragma unroll
      for (int i = 0; i < CO_T; ++i) dA[i] = lds0 + (unsigned)(pix * Pd + ((i ^ f) << 5) + cc * 8);
    }
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
      int jt = jt_of[jj];
      jt = jt < a.NJ ? jt : a.NJ - 1;
      const int tl = ws_div(jt, a.m_cit), cit = jt - tl * a.CI_T;
      // padded-linear offset of the tap: all nine taps (dh, dw) = (tl / 3 - 1, tl % 3 - 1): (dh + 1) * Wp + dw + 1; one kernel row: dw + 1
      const int shift = !k3 ? 0 : (a.tap_blks == 3 ? tl : (tl / 3) * Wp + (tl - (tl / 3) * 3));
      const int pix = 4 * kg + rr + shift;
      const int f = (pix >> a.x_sh) & a.x_mask;
      xB[jj] = lds0 + (unsigned)(a.x_off + pix * Px + ((cit ^ f) << 5) + cc * 8);
    }
    const unsigned d16 = (unsigned)(16 * Pd), x16 = (unsigned)(16 * Px);
    const unsigned dstep = (unsigned)(32 * Pd * a.WK), xstep = (unsigned)(32 * Px * a.WK);
    const unsigned dfirst = (unsigned)(32 * Pd * wk), xfirst = (unsigned)(32 * Px * wk);
    const int KSw = a.KS / a.WK;  // K-steps of this wave per chunk (even)
    // everything above is needed only after the first wait: keep it above it (conv_tile_kernel.inc.h)
#pragma unroll
    for (int i = 0; i < CO_T; ++i) asm volatile("" ::"v"(dA[i]));
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) asm volatile("" ::"v"(xB[jj]));

    h8 fa[2][CO_T], fb[2][JW];
#define CVX_WS_READ(SET, DO, XO)                                                                          \
  {                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < CO_T; ++i) fa[SET][i] = tr_frag2(dA[i] + (DO), dA[i] + (DO) + d16);   \
    _Pragma("unroll") for (int jj = 0; jj < JW; ++jj) fb[SET][jj] = tr_frag2(xB[jj] + (XO), xB[jj] + (XO) + x16); \
  }
#define CVX_WS_MFMA(SET)                                                                                  \
  {                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    _Pragma("unroll") for (int jj = 0; jj < JW; ++jj) _Pragma("unroll") for (int i = 0; i < CO_T; ++i) acc[i][jj] = \
        __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[SET][i], fb[SET][jj], acc[i][jj], 0, 0, 0);             \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
  }
    int issued = npre;
    unsigned so = 0;
    for (int c = 0; c < nch; ++c) {
      ws_wait_vm((issued - c - 1) * PW);  // this wave's pieces of chunk c have landed
      workgroup_barrier();                // ... everybody's; and every wave is done with chunk c - 1: its slot takes chunk c + NS - 1
      if (issued < nch) {
        issue();
        ++issued;
      }
      unsigned dof = so + dfirst, xof = so + xfirst;
      CVX_WS_READ(0, dof, xof);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      for (int s = 0; s < KSw; s += 2) {
        dof += dstep;
        xof += xstep;
        CVX_WS_READ(1, dof, xof);
        CVX_WS_MFMA(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dof += dstep;
        xof += xstep;
        if (s + 2 < KSw) CVX_WS_READ(0, dof, xof);
        CVX_WS_MFMA(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // nothing in flight across the back-edge
      }
      so = so + (unsigned)a.slot_bytes == (unsigned)(a.nslot * a.slot_bytes) ? 0u : so + (unsigned)a.slot_bytes;
    }
#undef CVX_WS_READ
#undef CVX_WS_MFMA
  }

  // ---- K-step waves fold into wk == 0 through the (now idle) ring ----
  if (a.WK > 1) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    if (wk > 0) {
#pragma unroll
      for (int i = 0; i < CO_T; ++i)
#pragma unroll
        for (int jj = 0; jj < JW; ++jj)
          *reinterpret_cast<f4*>(red + ((((wk - 1) * a.WN + wn) * CO_T + i) * JW + jj) * 256 + lane * 4) = acc[i][jj];
    }
    __syncthreads();
    if (wk == 0) {
      for (int w = 1; w < a.WK; ++w) {
#pragma unroll
        for (int i = 0; i < CO_T; ++i)
#pragma unroll
          for (int jj = 0; jj < JW; ++jj) {
            const f4 o = *reinterpret_cast<const f4*>(red + ((((w - 1) * a.WN + wn) * CO_T + i) * JW + jj) * 256 + lane * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][jj][r] += o[r];
          }
      }
    }
  }
  if (wk != 0) return;

  // ---- the split's slab: lane holds column ci = .. + (lane & 15), rows co = .. + 4 * (lane >> 4) + r ----
  const int Jtot = p.ntaps * p.cin_pad16;
  float* slab = p.slabs + (long long)bz * p.Cout * Jtot;
  const int lg = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int jj = 0; jj < JW; ++jj) {
    const int jt = jt_of[jj];
    if (jt >= a.NJ) continue;
    const int tl = ws_div(jt, a.m_cit), cit = jt - tl * a.CI_T;
    const int tap = a.tap_blks == 3 ? tapb * 3 + tl : tl;
    const int ci = ci0 + cit * 16 + lg;
    if (ci >= p.cin_pad16) continue;
    const int j = tap * p.cin_pad16 + ci;
#pragma unroll
    for (int i = 0; i < CO_T; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + i * 16 + fq * 4 + r;
        if (co < p.Cout) slab[(long long)co * Jtot + j] = acc[i][jj][r];
      }
    }
  }
#endif
}

// ---------------------------------------------------------------- host side ----------------------------------------------------------------

void swizzle_of(int cb, int* sh, int* mask) {
  const int slots = cb / 16;
  *sh = 0;
  *mask = 0;
  if (slots >= 2 && (slots & (slots - 1)) == 0 && slots <= 8) {
    int a = 0;
    while ((1 << a) < slots) ++a;
    *sh = 3 - a;
    *mask = slots - 1;
  }
}

// (CO_T, JW) pairs with a kernel instantiation
const int kCfgs[][2] = {{1, 3}, {2, 5}, {4, 3}, {5, 4}, {3, 3}, {2, 1}, {4, 1}, {4, 2}, {5, 2}, {3, 1}, {1, 1}, {3, 2}, {5, 1}, {2, 2}};
bool have_cfg(int co_t, int jw) {
  for (const auto& c : kCfgs)
    if (c[0] == co_t && c[1] == jw) return true;
  return false;
}

// channel tiles per workgroup for a count: the whole count up to 80, else the divisor (4, 3, 5, 2) with the least padding
int pick_co_t(int c) {
  const int t = (c + 15) / 16;
  if (t <= 5) return t;
  int best = 4, waste = 1 << 30;
  for (int q : {4, 3, 5, 2}) {
    const int w = (t + q - 1) / q * q - t;
    if (w < waste) {
      waste = w;
      best = q;
    }
  }
  return best;
}

// input-channel tiles per workgroup: pixel pitches whose transposed reads are bank-conflict free (1, 2, 4, 8 tiles: swizzled; odd counts)
int pick_ci_t(int tiles, int cap) {
  auto okset = [](int t) { return t == 1 || t == 2 || t == 3 || t == 4 || t == 5 || t == 7 || t == 8 || t == 9; };
  if (tiles <= cap && okset(tiles)) return tiles;
  for (int d : {8, 5, 4, 3, 2, 1})
    if (d <= cap && tiles % d == 0) return d;
  return 1;
}
unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned long long)d - 1) / (unsigned long long)d); }

constexpr int kLdsBudget = 76 * 1024;  // two workgroups per CU

bool ws_shape_ok(const WgradParams& p) {
  const bool k3 = p.std3x3 && p.ntaps == 9 && p.stride == 1 && p.IH == p.OH && p.IW == p.OW;
  const bool k1 = p.ntaps == 1 && p.stride == 1 && p.IH == p.OH && p.IW == p.OW;
  if (!k3 && !k1) return false;
  if (p.Cin % 8 || p.Cout % 8 || p.x_ld % 8 || p.dy_ld % 8 || p.Cin < 8) return false;
  if ((long long)p.B * p.x_bstride * 2 >= (1LL << 32) || (long long)p.B * p.dy_bstride * 2 >= (1LL << 32)) return false;
  if (k3 && (p.OW < 4 || p.OW > 1024)) return false;
  return true;
}

// The plan of a launch with `nsplit` pixel splits (nsplit <= 0: the planner's own choice, returned in *nsplit_out).
bool ws_plan(const WgradParams& p, int nsplit, WsPlan* out, int* co_t, int* jw, int* nsplit_out) {
  if (!ws_shape_ok(p)) return false;
  WsPlan a;
  memset(&a, 0, sizeof(a));
  const bool k3 = p.ntaps == 9;
  static const int force_tb = cvx_tune_int("CVX_WS_TAPBLK", 0), force_r = cvx_tune_int("CVX_WS_R", 0), force_ns = cvx_tune_int("CVX_WS_NSLOT", 0);
  static const int force_qd = cvx_tune_int("CVX_WS_QD", 0), wg_target = cvx_tune_int("CVX_WS_WGS", 512), lds_budget = cvx_tune_int("CVX_WS_LDS_KB", 0) * 1024;
  int budget = lds_budget > 0 ? lds_budget : kLdsBudget;
  const int CO_T = pick_co_t(p.Cout);
  a.COB = 16 * CO_T;
  a.n_co_blk = cvx_cdiv(p.Cout, a.COB);
  // input-channel block: the whole (padded) count where its pitch reads conflict-free (3x3: up to 80 channels, 1x1: up to 144), else a divisor
  const int ci_tiles = p.cin_pad16 / 16;
  a.CI_T = pick_ci_t(ci_tiles, k3 ? 5 : 9);
  a.CIB = 16 * a.CI_T;
  a.n_ci_blk = cvx_cdiv(ci_tiles, a.CI_T);
  // column tiles: all nine taps in one workgroup while 9 * CI_T tiles fit the instantiated (CO_T, JW) menu, else one kernel row each
  int JW = 0;
  a.tap_blks = 1;
  a.taps = k3 ? 9 : 1;
  auto try_cfg = [&](int taps, int wn) -> bool {
    const int nj = taps * a.CI_T;
    const int j = cvx_cdiv(nj, wn);
    if (!have_cfg(CO_T, j)) return false;
    a.NJ = nj;
    a.WN = wn;
    JW = j;
    return true;
  };
  bool ok = false;
  if (k3) {
    if (force_tb != 3 && CO_T * cvx_cdiv(9 * a.CI_T, 4) <= 18 && try_cfg(9, 4)) ok = true;
    if (!ok && force_tb != 1 && try_cfg(3, 4)) {
      ok = true;
      a.tap_blks = 3;
      a.taps = 3;
    }
    if (!ok && try_cfg(9, 4)) ok = true;
  } else {
    for (int wn : {4, 3, 2, 1})
      if (!ok && a.CI_T % wn == 0 && try_cfg(1, wn)) ok = true;
    for (int wn : {4, 3, 2, 1})
      if (!ok && try_cfg(1, wn)) ok = true;
  }
  if (!ok) return false;
  a.WK = k3 ? 1 : std::max(1, 4 / a.WN);
  if (a.WN * a.WK > 4) a.WK = 1;
  // the K-step fold needs (WK - 1) * WN * CO_T * JW KiB of the ring
  swizzle_of(a.COB, &a.d_sh, &a.d_mask);
  swizzle_of(a.CIB, &a.x_sh, &a.x_mask);
  const int NW = a.WN * a.WK;
  const int kq = 64 * a.WK;  // Qd granule: an even number of K-steps per K-step wave
  auto geom = [&](int R, int qd_1x1) {
    if (k3) {
      a.W = p.OW;
      a.H = p.OH;
      a.Wp = p.OW + 2;
      a.R = R;
      a.Qd = (R * a.Wp + kq - 1) / kq * kq;
      a.Tx = a.Qd + (a.tap_blks == 3 ? 0 : 2 * a.Wp) + 2;
      a.upi = cvx_cdiv(p.OH, R);
      a.total_units = p.B * a.upi;
    } else {
      a.W = a.H = a.Wp = a.R = 0;
      a.Qd = qd_1x1;
      a.Tx = a.Qd;
      a.upi = cvx_cdiv((long long)p.OH * p.OW, a.Qd);
      a.total_units = p.B * a.upi;
    }
    a.KS = a.Qd / 32;
    a.d_pieces = cvx_cdiv((long long)a.Qd * a.COB * 2, 1024);
    a.x_pieces = cvx_cdiv((long long)a.Tx * a.CIB * 2, 1024);
    a.dPW = cvx_cdiv(a.d_pieces, NW);
    a.xPW = cvx_cdiv(a.x_pieces, NW);
    a.x_off = a.d_pieces * 1024;
    a.slot_bytes = (a.d_pieces + a.x_pieces) * 1024;
  };
  auto fits = [&](int ns) {
    if (a.dPW + a.xPW > WS_MAXP) return false;
    if ((ns - 2) * (a.dPW + a.xPW) > 32) return false;  // counted vmcnt
    return ns * a.slot_bytes <= budget;
  };
  // offsets inside a chunk must fit the 20-bit field of the DMA table
  auto rel_ok = [&]() {
    const long long rows_x = k3 ? (long long)(a.R + 2) * p.OW : a.Qd, rows_d = k3 ? (long long)a.R * p.OW : a.Qd;
    return (rows_x * p.x_ld + p.Cin) * 2 < (1 << 20) && (rows_d * p.dy_ld + p.Cout) * 2 < (1 << 20);
  };
  bool found = false;
  double best_cost = 0;
  int bR = 0, bQ = 0, bNS = 0;
  for (int attempt = 0; attempt < 2 && !found; ++attempt, budget = 152 * 1024)  // two workgroups per CU, else one
  if (k3) {
    for (int R = 1; R <= p.OH; ++R) {
      if (force_r && R != force_r) continue;
      geom(R, 0);
      if (!rel_ok()) break;
      for (int ns = 4; ns >= 2; --ns) {
        if (force_ns && ns != force_ns) continue;
        if (!fits(ns)) continue;
        // cost: LDS fill bytes per useful pixel (halo rows re-read per chunk), K-step padding, ragged last chunk of an image; a deeper
        // ring and more chunks per image are worth a little
        const double useful = (double)p.OH * p.OW / a.upi;  // useful pixels per chunk, averaged over an image
        const double fill = (double)a.slot_bytes / useful;
        const double kpad = (double)a.Qd / useful;
        const double cost = fill * (0.6 + 0.4 * kpad) * (ns >= 3 ? 1.0 : 1.25);
        if (!found || cost < best_cost) {
          found = true;
          best_cost = cost;
          bR = R;
          bNS = ns;
        }
        break;  // the deepest ring that fits this R
      }
    }
  } else {
    // chunks never cross an image: the chunk length with the least K-step padding per image, the longer the better
    const int hw = p.OH * p.OW;
    for (int qd = 512; qd >= kq; qd -= kq) {
      if (force_qd && qd != force_qd) continue;
      geom(0, qd);
      if (!rel_ok()) continue;
      const int ns = force_ns ? force_ns : 3;
      if (!fits(ns)) continue;
      const double cost = (double)a.upi * qd / hw + 0.02 * a.upi * 256.0 / qd;
      if (!found || cost < best_cost - 1e-9) {
        found = true;
        best_cost = cost;
        bQ = qd;
        bNS = ns;
      }
    }
  }
  if (!found) return false;
  geom(bR, bQ);
  a.nslot = bNS;
  a.m_dupp = magic_of(a.COB / 8);
  a.m_xupp = magic_of(a.CIB / 8);
  a.m_wp = magic_of(a.Wp);
  a.m_cit = magic_of(a.CI_T);
  a.lds_bytes = std::max(a.nslot * a.slot_bytes, (a.WK - 1) * a.WN * CO_T * JW * 1024);
  if (a.lds_bytes > 160 * 1024) return false;
  // ---- pixel splits ----
  const int G = a.n_co_blk * a.n_ci_blk * a.tap_blks;
  int ns = nsplit;
  if (ns <= 0) {
    ns = std::max(1, wg_target / G);
    ns = std::min(ns, std::max(1, a.total_units / 2));  // at least two chunks per workgroup
    // slabs: at most half of the operand bytes (written and read back once each), at least 1 MB allowed
    const double operand = 2.0 * p.B * p.OH * p.OW * (p.Cin + p.Cout);
    const double slab = 4.0 * p.Cout * p.ntaps * p.cin_pad16;
    const int cap = (int)std::max(1.0, std::max(1.0 * (1 << 20), 0.25 * operand) / slab);
    ns = std::min(ns, cap);
  }
  ns = std::max(1, std::min(ns, a.total_units));
  a.units_per_split = cvx_cdiv(a.total_units, ns);
  ns = cvx_cdiv(a.total_units, a.units_per_split);  // (no empty splits)
  if (nsplit > 0) ns = nsplit;                         // a caller's count stands: surplus splits store zero slabs
  *out = a;
  *co_t = CO_T;
  *jw = JW;
  if (nsplit_out) *nsplit_out = ns;
  return true;
}

template <int CO_T, int JW>
int launch_ws(const WgradParams& p, const WsPlan& a, hipStream_t st) {
  static unsigned long long optin_mask = 0;
  CVX_TRY(cvx_lds_optin((const void*)conv_wgrad_stream_kernel<CO_T, JW>, 160 * 1024, &optin_mask));
  const int G = a.n_co_blk * a.n_ci_blk * a.tap_blks;
  const int ns8 = (p.nsplit + 7) / 8 * 8;
  hipLaunchKernelGGL((conv_wgrad_stream_kernel<CO_T, JW>), dim3(G * ns8), dim3(64 * a.WN * a.WK), a.lds_bytes, st, p, a);
  return 0;
}

}  // namespace

bool cvx_conv_wgrad_stream_supported(const WgradParams& p) {
  static const bool off = cvx_tune_set("CVX_NO_WGRAD_STREAM");
  if (off) return false;
  static const int max_cc = cvx_tune_int("CVX_WS_MAX_CC", 80 * 80);  // Cin x Cout the kernel is preferred up to
  static const int min_m = cvx_tune_int("CVX_WS_MIN_M", 800000);     // ... and the pixel count from which (below: the older kernels measured equal or ahead inside the step)
  static const int only_k = cvx_tune_int("CVX_WS_ONLY_K", 0);        // tuning: 1 / 3 = only the 1x1 / 3x3 layers
  if ((long long)p.Cin * p.Cout > max_cc && !(p.ntaps == 1 && p.Cout <= 80)) return false;
  if ((long long)p.B * p.OH * p.OW < min_m) return false;
  if (only_k && only_k * only_k != p.ntaps) return false;
  WsPlan a;
  int co_t, jw, ns;
  return ws_plan(p, 0, &a, &co_t, &jw, &ns);
}

// the planner's pixel-split count for a layer (the engine sizes the layer's slabs with it)
int cvx_conv_wgrad_stream_nsplit(const WgradParams& p) {
  WsPlan a;
  int co_t, jw, ns = 1;
  if (!ws_plan(p, 0, &a, &co_t, &jw, &ns)) return 1;
  return ns;
}

int cvx_conv_wgrad_stream_launch(const WgradParams& p, hipStream_t st) {
  WsPlan a;
  int co_t, jw, ns;
  CVX_CHECK(ws_plan(p, p.nsplit, &a, &co_t, &jw, &ns), "wgrad stream: unsupported shape");
  // (a caller's split count: the chunk ranges follow it; splits past the last chunk store zero slabs)
  a.units_per_split = cvx_cdiv(a.total_units, p.nsplit);
#define CVX_WS_CASE(C, J) \
  if (co_t == C && jw == J) return launch_ws<C, J>(p, a, st);
  CVX_WS_CASE(1, 3) CVX_WS_CASE(2, 5) CVX_WS_CASE(4, 3) CVX_WS_CASE(5, 4) CVX_WS_CASE(3, 3) CVX_WS_CASE(2, 1) CVX_WS_CASE(4, 1)
  CVX_WS_CASE(4, 2) CVX_WS_CASE(5, 2) CVX_WS_CASE(3, 1) CVX_WS_CASE(1, 1) CVX_WS_CASE(3, 2) CVX_WS_CASE(5, 1) CVX_WS_CASE(2, 2)
#undef CVX_WS_CASE
  CVX_FAIL("wgrad stream: no kernel for the planned tile");
}
