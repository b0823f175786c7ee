// 3x3 / stride-1 convolution with the input HALO TILE resident in LDS (gfx950).
//
// The generic implicit-GEMM kernels re-gather every input pixel once per tap (9x) through the L2->LDS DMA path,
// and that traffic -- not MFMA -- bounds them.  Here a persistent workgroup walks TH x 16 output patches:
//   * the (TH+2) x 18 x Cin input patch of a tile is DMA'd into LDS ONCE (1 KiB pieces, zero page for the image border),
//     double-buffered: the next tile's patch is in flight during the current tile's K loop;
//   * the workgroup's whole weight slice ([BN][9*Cin], capped at ~88 KiB by the launcher) is loaded into LDS once per
//     workgroup lifetime; only the first tile waits for it, in 1-3 groups of K-steps (counted vmcnt + raw s_barrier);
//   * the MFMA pixel operand of K-step (tap, channel chunk) is read straight out of the patch at the tap's
//     (dh, dw) offset: lane = pixel column, so a 16-pixel row segment is one operand sub-tile;
//   * patch layout [pixel][Cin/8 chunks of 16 B], chunk index XOR-swizzled by the pixel column so that 16
//     consecutive columns reading one logical chunk spread over all banks (<= 2-way under the gfx950
//     ds_read_b128 lane groups, checked by script); the swizzle is applied on the DMA source side.
// Used for the forward and the stride-1 data gradient (flipped taps, transposed weights) of every 3x3 conv whose
// gathered channel count is 16, 32, 64, 80, 128 or 144 -- the C2f bottlenecks and the Detect head, ~75 % of the MACs.
#include <cstdlib>
#include <type_traits>

#include "conv_tile_common.h"

namespace {
using namespace cvx_tile;

constexpr int HW = 18;  // halo patch width: 16 output columns + 2

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int WM, int WN, int MT, int NTW, int CIN_, int HV_ = 1>
struct HaloGeom {
  static constexpr int HV = HV_;                  // tile streams per workgroup (4 waves each) sharing the resident weights
  static constexpr int CIN = CIN_;                // gathered channels: 16, 32, 64, 128 (power of two) or 80, 144
  static constexpr bool POW2 = (CIN & (CIN - 1)) == 0;
  static constexpr int P = CIN >> 3;              // 16-byte chunks per pixel
  static constexpr int TH = WM * MT;              // output rows per workgroup
  static constexpr int BN = 16 * NTW * WN;        // output channels per workgroup
  static constexpr int NPIECE = BN / 16;          // 1 KiB weight pieces (16 rows x 64 B) per K-step
  static constexpr int PB = (NPIECE + 4 * HV_ - 1) / (4 * HV_);  // weight DMA instructions per wave per K-step
  static constexpr int STEP_HALVES = BN * BK;     // LDS halves of one K-step of weights
  static constexpr int NSTEPS = (9 * CIN + BK - 1) / BK;
  static constexpr int UNITS = (TH + 2) * HW * P;  // 16-byte units of the patch
  static constexpr int PATCH_PIECES = (UNITS + 63) / 64;
  static constexpr int STAT_BYTES = HV_ * WM * BN * 2 * 4;
  // 2 patch buffers per tile stream (+1 dump piece) | all weights of the workgroup's BN channels | stat scratch
  static constexpr int LDS_BYTES = (2 * HV_ * PATCH_PIECES + 1) * 1024 + NSTEPS * STEP_HALVES * 2 + STAT_BYTES;
};

// P 16-byte chunks per pixel.  Power-of-two P (2, 4, 8, 16): XOR swizzle that spreads 16 consecutive columns of one
// logical chunk over all 64 banks.  P = 10 / 18 (80 / 144 channels): the pixel stride of 40 / 72 dwords already walks
// the banks with period 8 -- at most 2-way conflicts -- and an XOR would not stay inside [0, P): no swizzle.
template <int P>
__device__ __forceinline__ int col_swz(int col) {
  if constexpr ((P & (P - 1)) != 0) {
    return 0;
  } else {
    constexpr int SH = P == 2 ? 3 : P == 4 ? 2 : P == 8 ? 1 : 0;
    return (col >> SH) & (P - 1);
  }
}

// PERSISTENT workgroups: gridDim.x of them per channel block walk the (image, row-tile, column-tile) list with stride
// gridDim.x.  Per workgroup, once: the tap table arrives as two packed kernel arguments (4 bits per tap -- no memory
// round trip between a K-step and its patch address) and the WHOLE weight slice [BN][9*Cin] (capped by the
// launcher), DMA'd into LDS K-step by K-step.  Per tile: the halo patch lands in one of two LDS buffers -- the DMA of
// tile t+1 is issued before the K loop of tile t, so its ~0.5 us L2->LDS latency hides behind the MFMAs -- then ONE
// barrier, a fully unrolled K loop (Cin is a template parameter: every (tap, chunk) is a constant and the scheduler
// overlaps the LDS fragment reads of later steps with the MFMAs of earlier ones) and the stores.  Only the first tile
// waits for the weights, in NG groups of K-steps (counted vmcnt).  BN statistics are kept per lane across tiles and
// folded once per workgroup.  Measured with cvx_debug_clock_buffer: the ring version paid 0.5 us per 32-wide K-step.
// HV = 2: eight waves, two tile streams (waves 0-3 / 4-7) that share the weights and the barriers: two waves per SIMD hide
// each other's LDS and MFMA latencies where a second workgroup per CU would not fit beside 74 KiB of weights.
template <int WM, int WN, int MT, int NTW, int CIN, int HV>
__global__ __launch_bounds__(256 * HV) void conv_halo_kernel(const ConvParams p, int tiles_x, int tiles_y, int total_tiles) {
  using G = HaloGeom<WM, WN, MT, NTW, CIN, HV>;
  constexpr int TH = G::TH, BN = G::BN, PB = G::PB, Cin = G::CIN, P = G::P, NSTEPS = G::NSTEPS;
  constexpr int PER_WAVE = (G::PATCH_PIECES + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half_t* patch0 = reinterpret_cast<half_t*>(smem);
  half_t* patch_dump = patch0 + 2 * HV * G::PATCH_PIECES * 512;
  half_t* wts = patch_dump + 512;
  float* sStat = reinterpret_cast<float*>(wts + NSTEPS * G::STEP_HALVES);

  const int tid = threadIdx.x;
  const int wave8 = tid >> 6, lane = tid & 63;
  const int half = wave8 >> 2, wave = wave8 & 3;  // tile stream, wave inside it
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fq = lane >> 4;
  const int nblk = blockIdx.y;
  const int H = p.IH, W = p.IW;
  clk_mark(p, 0);

  // halo patch of tile `tile` -> patch buffer `buf`.  Every wave issues PER_WAVE pieces (surplus ones hit the dump).
  auto issue_patch = [&](int tile, int buf) {
    const int tx = tile % tiles_x;
    const int t2 = tile / tiles_x;
    const int ty = t2 % tiles_y;
    const int b = t2 / tiles_y;
    const int y0 = ty * TH, x0 = tx * 16;
    const half_t* img = p.in + (long long)b * p.in_bstride;
    half_t* pbuf = patch0 + (half * 2 + buf) * (G::PATCH_PIECES * 512);
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) {
      const int piece = k * 4 + wave;
      const int u = piece * 64 + lane;
      const half_t* g = p.zeros;
      if (u < G::UNITS) {
        const int hp = u / P, phys = u - hp * P;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
          g = img + ((long long)iy * W + ix) * p.in_ld + ((phys ^ col_swz<P>(hx)) << 3);
      }
      half_t* dst = piece < G::PATCH_PIECES ? pbuf + piece * 512 : patch_dump;
      __builtin_amdgcn_global_load_lds((gbl_void_ptr)g, (lds_void_ptr)dst, 16, 0, 0);
    }
  };

  // tile streams: stream `half` of workgroup x takes tiles x*HV + half, + gridDim.x*HV, ...; the loop below runs while
  // stream 0 has a tile, a stream without one skips the work but keeps the barriers
  const int tstride = gridDim.x * HV;
  int tile = blockIdx.x * HV + half;
  bool valid = tile < total_tiles;  // stream 0: always (launcher)
  auto dump_patch = [&]() {         // keeps the per-wave DMA count uniform for the counted waits
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) __builtin_amdgcn_global_load_lds((gbl_void_ptr)p.zeros, (lds_void_ptr)patch_dump, 16, 0, 0);
  };
  if (valid) issue_patch(tile, 0);
  else dump_patch();

  const unsigned long long pos_pack = p.halo_pos, wt_pack = p.halo_wt;  // 4 bits per tap (cvx_halo_pack_taps)

  // ---- all weights of this workgroup's channels -> LDS, K-step after K-step ----
  {
    const int r16 = lane >> 2;
    const int kg = (lane & 3) ^ ((r16 >> 1) & 3);  // logical k-group this lane fetches (source-side swizzle)
    const half_t* wrow[PB];
    half_t* wdst[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int piece = q * (4 * HV) + wave8;  // wave-uniform
      const int n = nblk * BN + piece * 16 + r16;
      wrow[q] = (piece < G::NPIECE && n < p.Cout) ? p.wt + (long long)n * p.wt_ld : nullptr;
      wdst[q] = piece < G::NPIECE ? wts + piece * 512 : nullptr;
    }
    for (int s = 0; s < NSTEPS; ++s) {
      const int k = s * BK + kg * 8;
      const int tapb = k / Cin, cb = k - tapb * Cin;
      const bool kvalid = tapb < 9;
      const int wtap = (int)(wt_pack >> (4 * (kvalid ? tapb : 0))) & 15;
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        const half_t* g = (kvalid && wrow[q]) ? wrow[q] + wtap * Cin + cb : p.zeros;
        half_t* dst = wdst[q] ? wdst[q] + s * G::STEP_HALVES : patch_dump;
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)g, (lds_void_ptr)dst, 16, 0, 0);
      }
    }
  }
  clk_mark(p, 1);

  // per-lane constant parts of the operand addresses (halves)
  const int xbase = ((wm * MT) * HW + fr) * P;                      // patch unit of (row wm*MT, column fr)
  const half_t* wbase = wts + lds_row_off(wn * NTW * 16 + fr, fq);  // + j*16 rows (swizzle term unchanged: 16 | row step)
  f4 st1[NTW], st2[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) st1[j] = st2[j] = f4{0.f, 0.f, 0.f, 0.f};

  constexpr int NG = NSTEPS >= 12 ? 3 : (NSTEPS >= 4 ? 2 : 1);
  bool first = true;
  int buf = 0;
  for (int tile0 = blockIdx.x * HV; tile0 < total_tiles; tile0 += tstride, tile += tstride, buf ^= 1) {
    valid = tile < total_tiles;
    const int next = tile + tstride;
    const bool has_next = tile0 + tstride < total_tiles;  // block-uniform: stream 0 has another tile
    const bool next_valid = next < total_tiles;
    f4 acc[MT][NTW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const half_t* patch = patch0 + (half * 2 + buf) * (G::PATCH_PIECES * 512);
    // output coordinates of this tile
    const int tx = tile % tiles_x;
    const int t2 = tile / tiles_x;
    const int ty = t2 % tiles_y;
    const int b = t2 / tiles_y;
    long long out_off[MT], res_off[MT];
    bool pvalid[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int oy = ty * TH + wm * MT + i, ox = tx * 16 + fr;
      pvalid[i] = oy < H && ox < W;
      const long long pix = pvalid[i] ? (long long)oy * W + ox : 0;
      out_off[i] = (long long)b * p.out_bstride + pix * p.out_ld;
      res_off[i] = (long long)b * p.res_bstride + pix * p.res_ld;
    }

    static_for<0, NG>([&](auto gi) {
      constexpr int g = decltype(gi)::value;
      constexpr int beg = (NSTEPS * g) / NG, end = (NSTEPS * (g + 1)) / NG;
      if (g == 0 || first) {
        // loads retire in order.  First tile: [patch][weights...][next patch]: group g needs K-steps < end, i.e. all but
        // the youngest `pend` weight pieces (+ the next patch, issued after barrier 0).  Later tiles: only their patch.
        constexpr int pend = (NSTEPS - end) * PB;
        if (first) {
          if (g > 0 && has_next) wait_vmcnt<(pend + PER_WAVE > 63 ? 63 : pend + PER_WAVE)>();
          else wait_vmcnt<(pend > 63 ? 63 : pend)>();
        } else {
          wait_vmcnt<0>();
        }
        workgroup_barrier();  // group 0: also "every wave is done with the other patch buffer"
        if (g == 0) {
          if (first) clk_mark(p, 2);
          if (has_next) {
            if (next_valid) issue_patch(next, buf ^ 1);
            else dump_patch();
          }
        }
      }
      // software pipeline inside the group: the LDS fragment reads of K-step s+PD are issued before the MFMAs of
      // step s (one wave per SIMD at low occupancy: nobody else hides the ~128-cycle LDS latency).  sched_barrier pins
      // the order, so the MFMAs wait with a COUNTED lgkmcnt for their own fragments only.
      constexpr int PD = 2;
      h8 xa[PD + 1][MT], wb[PD + 1][NTW];
      auto load_step = [&](auto si) {
        constexpr int s = decltype(si)::value;
        constexpr int slot = (s - beg) % (PD + 1);
        // pixel operand: k-group 4s+fq -> (tap, channel chunk) -> patch address
        int tp, chunk;
        if constexpr (G::POW2 && Cin >= BK) {  // a 32-wide K-step lies inside one tap: tap and chunk base are step constants
          tp = (BK * s) / Cin;
          chunk = (((BK * s) % Cin) >> 3) + fq;
        } else {  // Cin = 16 (two taps per K-step) or 80 / 144 (a K-step may straddle a tap boundary): per k-group
          const int k0 = BK * s + 8 * fq;
          tp = k0 / Cin;
          chunk = (k0 - tp * Cin) >> 3;
          if (tp > 8) tp = 0;  // weights of the K tail are zero; any valid address will do
        }
        const int code = (int)(pos_pack >> (4 * tp)) & 15;
        const int dh = code >> 2, dw = code & 3;  // already +1
        const int sw = col_swz<P>(fr + dw);
        const half_t* xp = patch + ((xbase + (dh * HW + dw) * P + (chunk ^ sw)) << 3);
#pragma unroll
        for (int i = 0; i < MT; ++i) xa[slot][i] = *reinterpret_cast<const h8*>(xp + i * HW * P * 8);
#pragma unroll
        for (int j = 0; j < NTW; ++j) wb[slot][j] = *reinterpret_cast<const h8*>(wbase + s * G::STEP_HALVES + j * 16 * BK);
      };
      static_for<beg, (beg + PD < end ? beg + PD : end)>(load_step);
      static_for<beg, end>([&](auto si) {
        constexpr int s = decltype(si)::value;
        constexpr int slot = (s - beg) % (PD + 1);
        if constexpr (s + PD < end) load_step(std::integral_constant<int, s + PD>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[slot][j], xa[slot][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    if (first) clk_mark(p, 3);

    // ---- stores ----
    if (valid) epilogue_tile<WM, WN, MT, NTW>(p, acc, out_off, res_off, pvalid, wn, fq, nblk, st1, st2);
    first = false;
  }
  if (p.epi == CVX_EPI_RAW_STATS) stats_flush<HV * WM, WN, NTW>(p, st1, st2, half * WM + wm, wn, fr, fq, nblk, sStat, tid);
  clk_mark(p, 4);
}

template <int WM, int WN, int MT, int NTW, int CIN, int HV>
int launch_halo_hv(const ConvParams& p, hipStream_t stream, int gy) {
  using G = HaloGeom<WM, WN, MT, NTW, CIN, HV>;
  const int tiles_x = (p.IW + 15) / 16, tiles_y = (p.IH + G::TH - 1) / G::TH;
  const int total = tiles_x * tiles_y * p.B;
  static unsigned long long optin_mask = 0;  // per device (cvx_lds_optin)
  CVX_TRY(cvx_lds_optin((const void*)conv_halo_kernel<WM, WN, MT, NTW, CIN, HV>, G::LDS_BYTES, &optin_mask));
  // persistent: as many workgroups as fit on the 256 CUs at once (LDS-limited, at most CVX_HALO_OCC per CU), split
  // over the gy channel blocks; each walks the tile list with that stride (HV tiles per workgroup and trip)
  static const int occ_cap = cvx_tune_int("CVX_HALO_OCC", 4);
  int per_cu = (160 * 1024) / G::LDS_BYTES;
  per_cu = per_cu < 1 ? 1 : (per_cu > occ_cap ? occ_cap : per_cu);
  if (HV == 2 && per_cu > 2) per_cu = 2;  // 8 waves each
  int gx = (256 * per_cu) / gy / (g_cvx_grid_div > 0 ? g_cvx_grid_div : 1);
  if (gx < 1) gx = 1;
  if (gx * HV > total) gx = (total + HV - 1) / HV;
  dim3 grid(gx, gy);
  hipLaunchKernelGGL((conv_halo_kernel<WM, WN, MT, NTW, CIN, HV>), grid, dim3(256 * HV), G::LDS_BYTES, stream, p, tiles_x, tiles_y, total);
  return 0;
}

template <int WM, int WN, int MT, int NTW, int CIN>
int launch_halo_c(const ConvParams& p, hipStream_t stream, int gy) {
  using G1 = HaloGeom<WM, WN, MT, NTW, CIN, 1>;
  using G2 = HaloGeom<WM, WN, MT, NTW, CIN, 2>;
  if constexpr (G1::LDS_BYTES > 160 * 1024) {
    CVX_CHECK(false, "conv_halo: weight slice does not fit in LDS (launcher bug)");
  } else {
    // two tile streams per workgroup when only ONE single-stream workgroup would fit a CU (heavy weight slices) and the
    // two-stream layout does; register budget halves (256 per wave), so only the moderate register tiles
    static const bool hv2 = cvx_tune_int("CVX_HALO_HV", 2) != 1;
    if constexpr (G2::LDS_BYTES <= 160 * 1024 && 2 * G1::LDS_BYTES > 160 * 1024 && MT * NTW <= 8 && NTW <= 4) {
      if (hv2) return launch_halo_hv<WM, WN, MT, NTW, CIN, 2>(p, stream, gy);
    }
    return launch_halo_hv<WM, WN, MT, NTW, CIN, 1>(p, stream, gy);
  }
  return 0;
}

template <int WM, int WN, int MT, int NTW>
int launch_halo(const ConvParams& p, hipStream_t stream, int gy, int) {
  switch (p.Cin) {
    case 16: return launch_halo_c<WM, WN, MT, NTW, 16>(p, stream, gy);
    case 32: return launch_halo_c<WM, WN, MT, NTW, 32>(p, stream, gy);
    case 64: return launch_halo_c<WM, WN, MT, NTW, 64>(p, stream, gy);
    case 80: return launch_halo_c<WM, WN, MT, NTW, 80>(p, stream, gy);
    case 128: return launch_halo_c<WM, WN, MT, NTW, 128>(p, stream, gy);
    default: return launch_halo_c<WM, WN, MT, NTW, 144>(p, stream, gy);
  }
}

template <int WM, int MT>
int launch_m(int NT, const ConvParams& p, hipStream_t st, int gy, int l2) {
  switch (NT) {
    case 1: return launch_halo<WM, 1, MT, 1>(p, st, gy, l2);
    case 2: return launch_halo<WM, 1, MT, 2>(p, st, gy, l2);
    case 3: return launch_halo<WM, 1, MT, 3>(p, st, gy, l2);
    case 4: return launch_halo<WM, 1, MT, 4>(p, st, gy, l2);
    case 5: return launch_halo<WM, 1, MT, 5>(p, st, gy, l2);
    case 6: return launch_halo<WM, 1, MT, 6>(p, st, gy, l2);
    default: return launch_halo<WM, 1, MT, 8>(p, st, gy, l2);
  }
}

}  // namespace

bool cvx_conv_halo_supported(const ConvParams& p) {
  static const bool off = cvx_tune_set("CVX_NO_HALO");
  if (off || !p.zeros) return false;
  if (!(p.Cin == 16 || p.Cin == 32 || p.Cin == 64 || p.Cin == 80 || p.Cin == 128 || p.Cin == 144)) return false;
  if (p.IS != 1 || p.OS != 1 || p.oph != 0 || p.opw != 0) return false;
  if (p.OH2 != p.IH || p.OW2 != p.IW || p.OWr != p.IW) return false;
  if (p.ntaps != 9) return false;
  return p.halo_taps_ok != 0;  // all |dh|,|dw| <= 1, verified on the host where the tap table was built
}

// largest number of 16-channel tiles per workgroup whose weights (tiles*16 x 9*Cin fp16) stay within the LDS budget
static int halo_tile_cap(int cin) {
  static const int kb = cvx_tune_int("CVX_HALO_WKB", 88);
  int cap = (kb * 1024) / (16 * 9 * cin * 2);
  return cap < 1 ? 1 : (cap > 8 ? 8 : cap);
}

// LDS bytes of HaloGeom<.., TH rows, BN channels, cin> (mirrors HaloGeom::LDS_BYTES)
static int halo_lds_bytes(int th, int wm, int bn, int cin) {
  const int pieces = ((th + 2) * HW * (cin / 8) + 63) / 64;
  const int nsteps = (9 * cin + BK - 1) / BK;
  return (2 * pieces + 1) * 1024 + nsteps * bn * BK * 2 + wm * bn * 2 * 4;
}

int cvx_conv_halo_launch(const ConvParams& p, hipStream_t stream) {
  const int tiles = (p.Cout + 15) / 16;
  const int cap = halo_tile_cap(p.Cin);
  const long long hw = (long long)p.IH * p.IW;
  static const int allowed[] = {1, 2, 3, 4, 5, 6, 8};
  auto pick = [&](int th, int* gy_out) {  // fewest channel blocks within the cap and the LDS, then the smallest allowed tile count
    int gy = (tiles + cap - 1) / cap, want = (tiles + gy - 1) / gy;
    int NT = 8;
    for (int a : allowed)
      if (a >= want) {
        NT = a;
        break;
      }
    while (NT > cap) --NT;  // 7 is not compiled: cap 7 -> 6
    if (NT == 7) NT = 6;
    while (NT > 1 && halo_lds_bytes(th, 4, NT * 16, p.Cin) > 160 * 1024) NT = (NT == 8) ? 6 : NT - 1;
    *gy_out = (tiles + NT - 1) / NT;
    return NT;
  };
  int th = (hw >= 80 * 80 || hw * p.B >= 128 * 1024) ? 8 : ((hw >= 40 * 40 || cap < 2) ? 4 : 2);
  if (th == 8 && halo_lds_bytes(8, 4, (cap < 2 ? 1 : 2) * 16, p.Cin) > 160 * 1024) th = 4;  // wide channels: the 10-row patches do not fit twice
  // LDS bytes of the two-stream (HV = 2) form of HaloGeom<4, 1, th/4, NT, cin>: four patch buffers instead of two
  auto lds_hv2 = [&](int th_, int nt) {
    const int pieces = ((th_ + 2) * HW * (p.Cin / 8) + 63) / 64;
    const int nsteps = (9 * p.Cin + BK - 1) / BK;
    return (4 * pieces + 1) * 1024 + nsteps * nt * 16 * BK * 2 + 2 * 4 * nt * 16 * 2 * 4;
  };
  // 8-row tiles whose patches only fit a CU once run as ONE 4-wave workgroup per CU -- one wave per SIMD, nobody hides its LDS and MFMA
  // latencies (80 x 80, 64 -> 64: 80 us for 15 GFLOP).  The 4-row tile of the same layer fits with two tile streams (8 waves) and
  // measured ahead on every inference bench (tools/sweeps/ab_hv1.sh: SSD -3.7 %, YOLOv7 -2.4 %, CenterNet -2 %, YOLOv8-n eval -1.6 %,
  // train steps equal); 2: fewer channels per workgroup instead (measured behind), 0: off
  static const int hv1_mode = cvx_tune_int("CVX_HALO_HV1_MODE", 1);
  int nt_force = 0;
  if (th == 8 && hv1_mode != 0) {
    int gy0, NT0 = pick(8, &gy0);
    const bool one_stream = lds_hv2(8, NT0) > 160 * 1024 && 2 * halo_lds_bytes(8, 4, NT0 * 16, p.Cin) > 160 * 1024;
    if (one_stream && hv1_mode == 1) {
      int gy4, NT4 = pick(4, &gy4);
      if (lds_hv2(4, NT4) <= 160 * 1024 || 2 * halo_lds_bytes(4, 4, NT4 * 16, p.Cin) <= 160 * 1024) th = 4;  // only if the 4-row form does get its 8 waves
    }
    if (one_stream && hv1_mode == 2) {
      int nt = NT0;
      while (nt > 2 && lds_hv2(8, nt) > 160 * 1024) --nt;
      if (lds_hv2(8, nt) <= 160 * 1024) nt_force = nt;
    }
  }
  if (th == 8) {
    int gy, NT = pick(8, &gy);
    if (nt_force) {
      NT = nt_force;
      gy = (tiles + NT - 1) / NT;
    }
    CVX_TRY((launch_m<4, 2>(NT, p, stream, gy, 0)));
  } else if (th == 4) {
    int gy, NT = pick(4, &gy);
    CVX_TRY((launch_m<4, 1>(NT, p, stream, gy, 0)));
  } else {  // TH = 2, 2x2 waves, BN = 32 * NTW
    const int pairs = (tiles + 1) / 2, pcap = cap / 2;
    int gy = (pairs + pcap - 1) / pcap;
    int ntw = (pairs + gy - 1) / gy;
    if (ntw > 4) ntw = 4;
    gy = (pairs + ntw - 1) / ntw;
    switch (ntw) {
      case 1: CVX_TRY((launch_halo<2, 2, 1, 1>(p, stream, gy, 0))); break;
      case 2: CVX_TRY((launch_halo<2, 2, 1, 2>(p, stream, gy, 0))); break;
      case 3: CVX_TRY((launch_halo<2, 2, 1, 3>(p, stream, gy, 0))); break;
      default: CVX_TRY((launch_halo<2, 2, 1, 4>(p, stream, gy, 0))); break;
    }
  }
  CVX_HIP(hipGetLastError());
  return 0;
}
