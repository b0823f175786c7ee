// Implicit-GEMM convolution, second generation: the K loop is fed by LDS-DMA (global_load_lds_dwordx4)
// into a ring of LDS stages, several K-steps in flight, one raw s_barrier per step.
//
// Why: the first-generation kernel (conv_igemm.hip) stages through VGPRs with one K-step of prefetch, so every
// K-step exposes a global-load latency; rocprof showed the small-spatial layers (40x40, 20x20: 100-400 workgroups,
// 18-36 K-steps each) at 40-75 us for ~5 us of work.  Here:
//   * each wave issues PASSES x 1 KiB DMA pieces per K-step straight into LDS (no VGPR staging, no ds_write),
//     STAGES-1 steps ahead, and waits with a counted s_waitcnt vmcnt(N) (never 0 in the steady state);
//   * zero fill (image border, K tail, M/N tail) = the lane points its DMA at a zero page instead of predication,
//     so EXEC stays full and the LDS image is always completely written;
//   * the LDS image is the same 64-byte-row, XOR-swizzled layout as generation one -- the swizzle is applied on the
//     SOURCE side (which k-group a lane fetches), the DMA writes linearly (cdna_hip_programming.md rule 21);
//   * tile shapes: 128x(16 NT), 64x(16 NT) (4 waves along M) and 32x(32 NTW) (2x2 waves) so that the 40x40 and
//     20x20 levels still spread over >= 400 workgroups.
// Operand roles, fragment maps and the epilogues are those of conv_igemm.hip.
// Since the persistent kernels (conv_halo.hip: 3x3 stride 1, conv_pw.hip: 1x1) took over most layers, this one serves the
// stride-2 3x3 layers, the stem, 256-channel 3x3 inputs and any other shape; the output phases of a strided data gradient
// go out as one launch (ConvParams::phase, blockIdx.z).  In-kernel clocks (cvx_debug_clock_buffer) put its K-step at one
// L2->LDS DMA round trip, ~0.5 us, whatever the ring depth.
#include <algorithm>
#include <cstdlib>

#include "conv_igemm.h"

namespace {

constexpr int BK = 32;

__device__ __forceinline__ int lds_row_off(int row, int slot) { return row * BK + ((slot ^ ((row >> 1) & 3)) << 3); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// waits until at most ahead*PASSES DMA pieces are outstanding (ahead is block-uniform, 0..MAXA)
template <int PASSES, int MAXA>
__device__ __forceinline__ void wait_steps_ahead(int ahead) {
  if constexpr (MAXA == 0) {
    wait_vmcnt<0>();
  } else {
    if (ahead >= MAXA) wait_vmcnt<MAXA * PASSES>();
    else wait_steps_ahead<PASSES, MAXA - 1>(ahead);
  }
}

// compile-time LDS geometry of one tile configuration (shared by kernel and launcher)
template <int WM, int WN, int MT, int NTW, int STAGES>
struct Geom {
  static constexpr int BM = 16 * MT * WM, BN = 16 * NTW * WN;
  static constexpr int ROWS = BM + BN;
  static constexpr int PASSES = (ROWS + 63) / 64;
  // a stage holds the real rows plus one 16-row dump piece that every out-of-range DMA piece of the last pass targets
  static constexpr int STAGE_HALVES = (ROWS + 16) * BK;
  static constexpr int RING_BYTES = STAGES * STAGE_HALVES * 2;
  static constexpr int TAP_BYTES = CVX_MAX_TAPS * (int)sizeof(ConvTap);
  static constexpr int STAT_BYTES = WM * BN * 2 * 4;
  static constexpr int LDS_BYTES = RING_BYTES + TAP_BYTES + STAT_BYTES;
};

__device__ __forceinline__ void workgroup_barrier() {  // see conv_tile_common.h
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;

__device__ __forceinline__ void clk_mark(const ConvParams& p, int slot) {  // see conv_tile_common.h
  if (p.clk && threadIdx.x == 0) {
    unsigned long long* q = p.clk + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 8;
    q[slot] = wall_clock64();
    if (slot == 0) q[5] = clock64();
    if (slot == 4) q[6] = clock64();
  }
}

template <int WM, int WN, int MT, int NTW, int STAGES>
__global__ __launch_bounds__(256) void conv_igemm_dma_kernel(const ConvParams p) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int BM = 16 * MT * WM, BN = 16 * NTW * WN;
  using G = Geom<WM, WN, MT, NTW, STAGES>;
  constexpr int ROWS = G::ROWS;        // real rows per stage (A then B)
  constexpr int PASSES = G::PASSES;    // DMA instructions per wave per K-step (64 rows per pass)
  constexpr int PRE = STAGES - 1;      // K-steps in flight ahead of the one being computed
  constexpr int STAGE_HALVES = G::STAGE_HALVES;
  // ---- one dynamic LDS array (a second __shared__ object can make hipcc drain the DMA queue, guide 5.x item 4a) ----
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half_t* ring = reinterpret_cast<half_t*>(smem);
  ConvTap* sTap = reinterpret_cast<ConvTap*>(smem + G::RING_BYTES);
  float* sStat = reinterpret_cast<float*>(smem + G::RING_BYTES + G::TAP_BYTES);  // [WM][BN][2]

  // phase view: the launch's own tap table / output phase, or the one blockIdx.z selects (merged stride-2 data gradients)
  ConvParams::Phase v{p.taps, p.ntaps, p.OH2, p.OW2, p.oph, p.opw};
  if (p.nphase > 1) v = p.phase[blockIdx.z];
  if ((long long)blockIdx.x * BM >= (long long)p.B * v.OH2 * v.OW2) return;  // gridDim.x covers the largest phase

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fq = lane >> 4;
  const int nblk = blockIdx.y;
  const long long M = (long long)p.B * v.OH2 * v.OW2;
  const long long m_base = (long long)blockIdx.x * BM;

  clk_mark(p, 0);
  if (tid < v.ntaps) sTap[tid] = v.taps[tid];

  // ---- per-lane DMA assignment: pass q covers stage rows q*64 + wave*16 + (lane>>2), physical slot lane&3 ----
  const int r16 = lane >> 2;
  const int kg = (lane & 3) ^ ((r16 >> 1) & 3);  // logical k-group this lane fetches (source-side swizzle)
  const half_t* src_base[PASSES];                // pixel base (A rows) or weight row (B rows); nullptr = always zero
  int ih0[PASSES], iw0[PASSES];
#pragma unroll
  for (int q = 0; q < PASSES; ++q) {
    const int row = q * 64 + wave * 16 + r16;
    src_base[q] = nullptr;
    ih0[q] = iw0[q] = 0;
    if (row < BM) {
      long long m = m_base + row;
      if (m < M) {
        const unsigned mu = (unsigned)m;  // M < 2^31 (checked by the launcher): 32-bit divisions
        const unsigned tq = mu / (unsigned)v.OW2;
        int ow2 = (int)(mu - tq * (unsigned)v.OW2);
        int b = (int)(tq / (unsigned)v.OH2);
        int oh2 = (int)(tq - (unsigned)b * (unsigned)v.OH2);
        ih0[q] = oh2 * p.IS;
        iw0[q] = ow2 * p.IS;
        src_base[q] = p.in + (long long)b * p.in_bstride;
      }
    } else if (row < BM + BN) {
      int n = nblk * BN + (row - BM);
      if (n < p.Cout) src_base[q] = p.wt + (long long)n * p.wt_ld;
    }
  }
  int c = kg * 8, tap = 0;
  while (c >= p.Cin) {
    c -= p.Cin;
    ++tap;
  }
  const int nsteps = (v.ntaps * p.Cin + BK - 1) / BK;
  __syncthreads();  // tap table visible (no DMA outstanding yet)
  clk_mark(p, 1);

  auto issue = [&](int stage) {
    const bool kvalid = tap < v.ntaps;
    ConvTap td = sTap[kvalid ? tap : 0];
    half_t* stage_base = ring + stage * STAGE_HALVES;
#pragma unroll
    for (int q = 0; q < PASSES; ++q) {
      int row0 = q * 64 + wave * 16;  // wave-uniform: the 16 rows of this piece are all A, all B or all dummy
      const half_t* g = p.zeros;
      if (row0 >= ROWS) {
        row0 = ROWS;  // dump piece
      } else if (row0 < BM) {
        int ih = ih0[q] + td.dh, iw = iw0[q] + td.dw;
        if (kvalid && src_base[q] && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW)
          g = src_base[q] + ((long long)ih * p.IW + iw) * p.in_ld + c;
      } else {
        if (kvalid && src_base[q]) g = src_base[q] + td.wtap * p.Cin + c;
      }
      __builtin_amdgcn_global_load_lds((gbl_void_ptr)g, (lds_void_ptr)(stage_base + row0 * BK), 16, 0, 0);
    }
    c += BK;
    while (c >= p.Cin) {
      c -= p.Cin;
      ++tap;
    }
  };

  f4 acc[MT][NTW];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  int issued = 0;
  for (; issued < PRE && issued < nsteps; ++issued) issue(issued % STAGES);

  for (int s = 0; s < nsteps; ++s) {
    // DMA pieces of K-step s have landed once at most (steps issued after s) * PASSES pieces remain outstanding
    wait_steps_ahead<PASSES, PRE - 1>(issued - 1 - s);  // block-uniform, in [0, PRE-1]
    workgroup_barrier();  // every wave's pieces of step s are in LDS; everyone is done reading stage (s-1)
    if (s == 0) clk_mark(p, 2);
    if (issued < nsteps) {
      issue(issued % STAGES);  // refills the stage that was computed in the previous iteration
      ++issued;
    }
    const half_t* st = ring + (s % STAGES) * STAGE_HALVES;
    h8 xa[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) xa[i] = *reinterpret_cast<const h8*>(&st[lds_row_off((wm * MT + i) * 16 + fr, fq)]);
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      h8 wb = *reinterpret_cast<const h8*>(&st[lds_row_off(BM + (wn * NTW + j) * 16 + fr, fq)]);
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb, xa[i], acc[i][j], 0, 0, 0);
    }
  }

  clk_mark(p, 3);
  // ---- epilogue: lane holds pixel (fr) x channels 4*fq..4*fq+3 of every (i, j) tile ----
  long long out_off[MT], res_off[MT];
  bool pvalid[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    long long m = m_base + (wm * MT + i) * 16 + fr;
    pvalid[i] = m < M;
    const unsigned mu = pvalid[i] ? (unsigned)m : 0u;
    const unsigned tq = mu / (unsigned)v.OW2;
    int ow2 = (int)(mu - tq * (unsigned)v.OW2);
    int b = (int)(tq / (unsigned)v.OH2);
    int oh2 = (int)(tq - (unsigned)b * (unsigned)v.OH2);
    long long pix = (long long)(oh2 * p.OS + v.oph) * p.OWr + (ow2 * p.OS + v.opw);
    out_off[i] = (long long)b * p.out_bstride + pix * p.out_ld;
    res_off[i] = (long long)b * p.res_bstride + pix * p.res_ld;
  }

  if (p.epi == CVX_EPI_RAW_STATS) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int chl = (wn * NTW + j) * 16 + fq * 4;  // channel inside the block's BN range
      const int n0 = nblk * BN + chl;
      float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (pvalid[i]) {
          if (n0 < p.Cout) {
            cvx_store_raw4(p, out_off[i] + n0, acc[i][j]);  // fp32 (fp16: raw16), see conv_tile_common.h
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s1[r] += acc[i][j][r];
            s2[r] += acc[i][j][r] * acc[i][j][r];
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = cvx_wave_sum16(s1[r]), b2 = cvx_wave_sum16(s2[r]);
        if (fr == 0) {
          sStat[(wm * BN + chl + r) * 2 + 0] = a;
          sStat[(wm * BN + chl + r) * 2 + 1] = b2;
        }
      }
    }
    __syncthreads();
    for (int t = tid; t < BN * 2; t += 256) {
      int ch = t >> 1, which = t & 1;
      int n = nblk * BN + ch;
      if (n < p.Cout) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) v += sStat[(w * BN + ch) * 2 + which];
        cvx_fix_atomic_add(p.stats, ((long long)(blockIdx.x % p.stats_replicas) * p.Cout + n) * 2 + which, v);
      }
    }
    clk_mark(p, 4);
    return;
  }

#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int n0 = nblk * BN + (wn * NTW + j) * 16 + fq * 4;
    if (n0 >= p.Cout) continue;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (!pvalid[i]) continue;
      f4 v = acc[i][j];
      if (p.epi == CVX_EPI_AFFINE_SILU) {
        f4 sc = *reinterpret_cast<const f4*>(p.scale + n0);
        f4 sh = *reinterpret_cast<const f4*>(p.shift + n0);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] * sc[r] + sh[r];
        f4 rv = {0.f, 0.f, 0.f, 0.f};
        if (p.res) {
          h4 rr = *reinterpret_cast<const h4*>(p.res + res_off[i] + n0);
#pragma unroll
          for (int r = 0; r < 4; ++r) rv[r] = (float)rr[r];
        }
        if (p.res_pre) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += rv[r];
        }
        if (p.act_kind == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = cvx_silu(v[r]);
        } else if (p.act_kind == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (!p.res_pre) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += rv[r];
        }
      } else if (p.epi == CVX_EPI_BIAS_F32) {
        f4 bb = *reinterpret_cast<const f4*>(p.bias + n0);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bb[r];
        *reinterpret_cast<f4*>(p.out32 + out_off[i] + n0) = v;
        continue;
      }
      half_t* dst = p.out16 + out_off[i] + n0;
      if (p.accumulate) {
        h4 old = *reinterpret_cast<const h4*>(dst);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)old[r];
      }
      h4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      *reinterpret_cast<h4*>(dst) = o;
    }
  }
  clk_mark(p, 4);
}

int env_int(const char* name, int dflt) { return cvx_tune_int(name, dflt); }

template <int WM, int WN, int MT, int NTW, int STAGES>
int launch_st(const ConvParams& p, hipStream_t stream, dim3 grid) {
  using G = Geom<WM, WN, MT, NTW, STAGES>;
  static unsigned long long optin_mask = 0;  // per device (cvx_lds_optin)
  CVX_TRY(cvx_lds_optin((const void*)conv_igemm_dma_kernel<WM, WN, MT, NTW, STAGES>, G::LDS_BYTES, &optin_mask));
  hipLaunchKernelGGL((conv_igemm_dma_kernel<WM, WN, MT, NTW, STAGES>), grid, dim3(256), G::LDS_BYTES, stream, p);
  return 0;
}

// ring depth per tile family.  Measured on MI355X (bench.py, bs 32): deeper rings LOSE -- 2 stages for the 128/64-row
// tiles and 3 for the 32-row tiles were the fastest of {2,3,4,6,8}; LDS footprint (workgroups per CU) matters more than
// K-steps in flight.  CVX_STAGES_BIG / CVX_STAGES_SMALL select among the compiled depths for A/B runs.
template <int WM, int WN, int MT, int NTW>
int launch_cfg(const ConvParams& p, hipStream_t stream, dim3 grid) {
  static const int st = env_int(WN == 2 ? "CVX_STAGES_SMALL" : "CVX_STAGES_BIG", WN == 2 ? 3 : 2);
  if (st >= 4) return launch_st<WM, WN, MT, NTW, 4>(p, stream, grid);
  if (st == 3) return launch_st<WM, WN, MT, NTW, 3>(p, stream, grid);
  return launch_st<WM, WN, MT, NTW, 2>(p, stream, grid);
}

template <int MT>
int launch_m4(int NT, const ConvParams& p, hipStream_t st, dim3 grid) {  // 4 waves along M
  switch (NT) {
    case 1: return launch_cfg<4, 1, MT, 1>(p, st, grid);
    case 2: return launch_cfg<4, 1, MT, 2>(p, st, grid);
    case 3: return launch_cfg<4, 1, MT, 3>(p, st, grid);
    case 4: return launch_cfg<4, 1, MT, 4>(p, st, grid);
    case 5: return launch_cfg<4, 1, MT, 5>(p, st, grid);
    case 6: return launch_cfg<4, 1, MT, 6>(p, st, grid);
    default: return launch_cfg<4, 1, MT, 8>(p, st, grid);
  }
}

}  // namespace

int cvx_conv_igemm_dma_launch(const ConvParams& p, hipStream_t stream) {
  CVX_CHECK(p.zeros && ((uintptr_t)p.zeros % 16) == 0, "conv_igemm_dma: needs a 16-byte aligned zero page");
  long long M = (long long)p.B * p.OH2 * p.OW2;
  int nz = 1;
  if (p.nphase > 1) {  // merged phases: gridDim.x covers the largest, blockIdx.z selects
    CVX_CHECK(p.nphase <= 4, "conv_igemm_dma: at most 4 merged phases");
    nz = p.nphase;
    M = 0;
    for (int q = 0; q < p.nphase; ++q) M = std::max(M, (long long)p.B * p.phase[q].OH2 * p.phase[q].OW2);
  }
  CVX_CHECK(M < (1LL << 31), "conv_igemm_dma: more than 2^31 output pixels per launch");
  const int tiles = (p.Cout + 15) / 16;
  static const int allowed[] = {1, 2, 3, 4, 5, 6, 8};
  // tile height by problem size (pixels): big M -> 128 rows; below t128 -> 64 rows; below t64 -> 32 rows
  static const long long t128 = env_int("CVX_T128", 128 * 1024), t64 = env_int("CVX_T64", 64 * 1024);
  int BM = 128;
  if (M < t128) BM = (M >= t64) ? 64 : 32;
  // heavy weights (>= 512 K elements: ResNet / DLA layers with 256+ channels in 3x3, 1024+ in 1x1): every workgroup streams the
  // whole [BN][K] weight block from L2, so 32-row tiles re-read it M/32 times -- the L2->LDS path, not the MFMAs, was the limit
  // (DeepLabv3+ ASPP: 6.4 GB per launch); 64 rows from 8 K pixels on (DeepLabv3+ forward 11.85 -> 10.27 ms)
  static const long long heavy = env_int("CVX_HEAVY_W", 512 * 1024), t64h = env_int("CVX_T64_HEAVY", 8192);
  if (BM == 32 && (long long)p.ntaps * p.Cin * p.Cout >= heavy && M >= t64h) BM = 64;
  if (BM == 32) {
    // 2x2 waves: BN = 32 * NTW
    int pairs = (tiles + 1) / 2;
    int gy = (pairs + 3) / 4;
    int ntw = (pairs + gy - 1) / gy;  // 1..4
    gy = (pairs + ntw - 1) / ntw;
    dim3 grid(cvx_cdiv(M, 32), gy, nz);
    switch (ntw) {
      case 1: CVX_TRY((launch_cfg<2, 2, 1, 1>(p, stream, grid))); break;
      case 2: CVX_TRY((launch_cfg<2, 2, 1, 2>(p, stream, grid))); break;
      case 3: CVX_TRY((launch_cfg<2, 2, 1, 3>(p, stream, grid))); break;
      default: CVX_TRY((launch_cfg<2, 2, 1, 4>(p, stream, grid))); break;
    }
  } else {
    int gy = (tiles + 7) / 8;
    int want = (tiles + gy - 1) / gy;
    int NT = 8;
    for (int a : allowed)
      if (a >= want) {
        NT = a;
        break;
      }
    gy = (tiles + NT - 1) / NT;
    dim3 grid(cvx_cdiv(M, BM), gy, nz);
    if (BM == 128) CVX_TRY(launch_m4<2>(NT, p, stream, grid));
    else CVX_TRY(launch_m4<1>(NT, p, stream, grid));
  }
  CVX_HIP(hipGetLastError());
  return 0;
}
