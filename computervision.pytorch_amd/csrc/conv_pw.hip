// Pointwise (1x1, stride 1) convolution and its data gradient as a persistent, barrier-free GEMM (gfx950).
//
// out[m][n] = sum_k in[m][k] * W[n][k] over M = B*H*W pixels.  Every pixel row of the NHWC view is K contiguous fp16
// values, so the activation operand never touches LDS:
//   * the workgroup's weight slice [BN][Cin] (<= ~64 KiB, launcher-capped) is DMA'd into LDS ONCE, in the same
//     swizzled [K-step][BN][32] image the other kernels use -- one vmcnt wait + one barrier per workgroup lifetime;
//   * each WAVE owns whole wave-tiles of MT*16 pixels (round-robin over the launch, persistent): it loads its
//     pixels' complete K range straight from global memory into registers in MFMA operand layout (lane = pixel,
//     16 bytes = 8 channels of one k-group; NSTEPS*MT independent 16-byte loads in flight per lane), then runs
//     NSTEPS x (NTW ds_read_b128 weights + MT*NTW MFMAs) with no synchronisation of any kind;
//   * latency is hidden across waves (2-4 workgroups per CU), not inside one: loads and the epilogue's stores share
//     vmcnt, which the hardware does not retire in order between the two kinds, so a counted in-wave prefetch across
//     the stores is not expressible; a wave simply overlaps its memory wait with its neighbours' MFMAs.
// The ring kernel this replaces for 1x1 layers paid one DMA round trip (~0.5 us, measured with
// cvx_debug_clock_buffer) per 32-wide K-step.  BN statistics are kept per lane across tiles and folded once.
#include <cstdlib>
#include <type_traits>

#include "conv_tile_common.h"

namespace {
using namespace cvx_tile;

template <int MT, int NTW, int NSTEPS>
struct PwGeom {
  static constexpr int BN = 16 * NTW;
  static constexpr int NPIECE = NTW;               // 1 KiB weight pieces (16 rows x 64 B) per K-step
  static constexpr int PB = (NPIECE + 3) / 4;      // weight DMA instructions per wave per K-step
  static constexpr int STEP_HALVES = BN * BK;
  static constexpr int STAT_BYTES = 4 * BN * 2 * 4;
  static constexpr int LDS_BYTES = 1024 + NSTEPS * STEP_HALVES * 2 + STAT_BYTES;  // dump piece | weights | stat scratch
};

template <int MT, int NTW, int NSTEPS>
__global__ __launch_bounds__(256) void conv_pw_kernel(const ConvParams p, long long M, int hw, int n_wave_tiles) {
  using G = PwGeom<MT, NTW, NSTEPS>;
  constexpr int BN = G::BN, PB = G::PB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half_t* dump = reinterpret_cast<half_t*>(smem);
  half_t* wts = dump + 512;
  float* sStat = reinterpret_cast<float*>(wts + NSTEPS * G::STEP_HALVES);

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int fr = lane & 15, fq = lane >> 4;
  const int nblk = blockIdx.y;
  const int Cin = p.Cin;
  clk_mark(p, 0);

  // ---- weights of this workgroup's channels -> LDS, once ----
  {
    const int r16 = lane >> 2;
    const int kg = (lane & 3) ^ ((r16 >> 1) & 3);  // logical k-group this lane fetches (source-side swizzle)
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int piece = q * 4 + wave;  // wave-uniform
      const int n = nblk * BN + piece * 16 + r16;
      const half_t* wrow = (piece < G::NPIECE && n < p.Cout) ? p.wt + (long long)n * p.wt_ld : nullptr;
      half_t* wdst = piece < G::NPIECE ? wts + piece * 512 : nullptr;
#pragma unroll
      for (int s = 0; s < NSTEPS; ++s) {
        const int k = s * BK + kg * 8;
        const half_t* g = (wrow && k < Cin) ? wrow + k : p.zeros;
        half_t* dst = wdst ? wdst + s * G::STEP_HALVES : dump;
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)g, (lds_void_ptr)dst, 16, 0, 0);
      }
    }
  }

  clk_mark(p, 1);
  const bool flat_in = p.in_bstride == (long long)hw * p.in_ld;
  const bool flat_out = p.out_bstride == (long long)hw * p.out_ld;
  const half_t* wbase = wts + lds_row_off(fr, fq);  // + j*16 rows (swizzle term unchanged: 16 | row step)
  const int kfull = Cin / BK;  // K-steps that lie entirely inside the view's channels (block-uniform)
  f4 st1[NTW], st2[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) st1[j] = st2[j] = f4{0.f, 0.f, 0.f, 0.f};

  bool weights_ready = false;
  const int wstride = gridDim.x * 4;
  for (int t = blockIdx.x * 4 + wave; t < n_wave_tiles; t += wstride) {
    // ---- this wave's pixels: every K value, straight into MFMA operand registers ----
    h8 xa[NSTEPS][MT];
    long long out_off[MT], res_off[MT];
    bool pvalid[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const long long m = (long long)t * (MT * 16) + i * 16 + fr;
      pvalid[i] = m < M;
      const long long mm = pvalid[i] ? m : M - 1;  // clamped: the row is loaded but never stored
      long long ioff, ooff, roff;
      if (flat_in && flat_out && !p.res) {
        ioff = mm * p.in_ld;
        ooff = mm * p.out_ld;
        roff = 0;
      } else {
        const unsigned mu = (unsigned)mm;  // M < 2^31 (launcher)
        const unsigned b = mu / (unsigned)hw;
        const unsigned pix = mu - b * (unsigned)hw;
        ioff = (long long)b * p.in_bstride + (long long)pix * p.in_ld;
        ooff = (long long)b * p.out_bstride + (long long)pix * p.out_ld;
        roff = (long long)b * p.res_bstride + (long long)pix * p.res_ld;
      }
      out_off[i] = ooff;
      res_off[i] = roff;
      const half_t* src = p.in + ioff + fq * 8;
#pragma unroll
      for (int s = 0; s < NSTEPS; ++s) {
        if (s < kfull || s * BK + fq * 8 < Cin) {  // first test is scalar: full steps take no exec masking
          xa[s][i] = *reinterpret_cast<const h8*>(src + s * BK);
        } else {
          xa[s][i] = h8{0, 0, 0, 0, 0, 0, 0, 0};  // beyond the view's channels: the weights there are zero, keep 0 * x finite
        }
      }
    }
    if (!weights_ready) {  // first tile only: the weight DMAs are older than this tile's loads, so they are covered by
      wait_vmcnt<0>();     // the same wait the first MFMA needs anyway
      workgroup_barrier();
      weights_ready = true;
      clk_mark(p, 2);
    }
    f4 acc[MT][NTW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NSTEPS; ++s) {
      h8 wb[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) wb[j] = *reinterpret_cast<const h8*>(wbase + s * G::STEP_HALVES + j * 16 * BK);
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[j], xa[s][i], acc[i][j], 0, 0, 0);
    }
    if (t == blockIdx.x * 4) clk_mark(p, 3);
    epilogue_tile<4, 1, MT, NTW>(p, acc, out_off, res_off, pvalid, 0, fq, nblk, st1, st2);
  }
  if (!weights_ready) {  // a wave without tiles still owes the workgroup its weight pieces and the barrier
    wait_vmcnt<0>();
    workgroup_barrier();
  }
  if (p.epi == CVX_EPI_RAW_STATS) stats_flush<4, 1, NTW>(p, st1, st2, wave, 0, fr, fq, nblk, sStat, tid);
  clk_mark(p, 4);
}

template <int MT, int NTW, int NSTEPS>
int launch_pw(const ConvParams& p, hipStream_t stream, int gy) {
  using G = PwGeom<MT, NTW, NSTEPS>;
  if constexpr (G::LDS_BYTES > 160 * 1024) {
    CVX_CHECK(false, "conv_pw: weight slice does not fit in LDS (launcher bug)");
  } else {
    const int hw = p.OH2 * p.OW2;
    const long long M = (long long)p.B * hw;
    const int n_wave_tiles = (int)((M + MT * 16 - 1) / (MT * 16));
    static unsigned long long optin_mask = 0;  // per device (cvx_lds_optin)
    CVX_TRY(cvx_lds_optin((const void*)conv_pw_kernel<MT, NTW, NSTEPS>, G::LDS_BYTES, &optin_mask));
    static const int occ_cap = cvx_tune_int("CVX_PW_OCC", 4);
    int per_cu = (160 * 1024) / G::LDS_BYTES;
    per_cu = per_cu < 1 ? 1 : (per_cu > occ_cap ? occ_cap : per_cu);
    int gx = (256 * per_cu) / gy / (g_cvx_grid_div > 0 ? g_cvx_grid_div : 1);
    const int need = (n_wave_tiles + 3) / 4;
    if (gx > need) gx = need;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL((conv_pw_kernel<MT, NTW, NSTEPS>), dim3(gx, gy), dim3(256), G::LDS_BYTES, stream, p, M, hw, n_wave_tiles);
  }
  return 0;
}

template <int MT, int NSTEPS>
int launch_pw_n(int NT, const ConvParams& p, hipStream_t st, int gy) {
  switch (NT) {
    case 1: return launch_pw<MT, 1, NSTEPS>(p, st, gy);
    case 2: return launch_pw<MT, 2, NSTEPS>(p, st, gy);
    case 3: return launch_pw<MT, 3, NSTEPS>(p, st, gy);
    case 4: return launch_pw<MT, 4, NSTEPS>(p, st, gy);
    case 5: return launch_pw<MT, 5, NSTEPS>(p, st, gy);
    case 6: return launch_pw<MT, 6, NSTEPS>(p, st, gy);
    default: return launch_pw<MT, 8, NSTEPS>(p, st, gy);
  }
}

}  // namespace

bool cvx_conv_pw_supported(const ConvParams& p) {
  static const bool off = cvx_tune_set("CVX_NO_PW");
  if (off || !p.zeros || !p.pointwise) return false;
  if (p.IS != 1 || p.OS != 1 || p.oph != 0 || p.opw != 0 || p.ntaps != 1) return false;
  if (p.OH2 != p.IH || p.OW2 != p.IW || p.OWr != p.IW) return false;
  return p.Cin <= 512;
}

int cvx_conv_pw_launch(const ConvParams& p, hipStream_t stream) {
  const long long M = (long long)p.B * p.OH2 * p.OW2;
  CVX_CHECK(M < (1LL << 31), "conv_pw: more than 2^31 output pixels per launch");
  CVX_CHECK(((uintptr_t)p.in % 16) == 0 && p.in_ld % 8 == 0 && p.in_bstride % 8 == 0, "conv_pw: 16-byte aligned pixel rows needed");
  const int nsteps = (p.Cin + BK - 1) / BK;  // 1..16
  static const int steps_allowed[] = {1, 2, 3, 4, 6, 8, 12, 16};
  int NS = 16;
  for (int a : steps_allowed)
    if (a >= nsteps) {
      NS = a;
      break;
    }
  // channel tiles per workgroup: fewest channel blocks whose weights (NT*16 x NS*32 fp16 = NT*NS KiB) fit the budget
  static const int kb = cvx_tune_int("CVX_PW_WKB", 64);
  const int tiles = (p.Cout + 15) / 16;
  int cap = kb / NS;
  cap = cap < 1 ? 1 : (cap > 8 ? 8 : cap);
  static const int allowed[] = {1, 2, 3, 4, 5, 6, 8};
  int gy = (tiles + cap - 1) / cap, want = (tiles + gy - 1) / gy;
  int NT = 8;
  for (int a : allowed)
    if (a >= want) {
      NT = a;
      break;
    }
  while (NT > cap) --NT;
  if (NT == 7) NT = 6;
  gy = (tiles + NT - 1) / NT;
  switch (NS) {
    case 1: CVX_TRY((launch_pw_n<2, 1>(NT, p, stream, gy))); break;
    case 2: CVX_TRY((launch_pw_n<2, 2>(NT, p, stream, gy))); break;
    case 3: CVX_TRY((launch_pw_n<2, 3>(NT, p, stream, gy))); break;
    case 4: CVX_TRY((launch_pw_n<2, 4>(NT, p, stream, gy))); break;
    case 6: CVX_TRY((launch_pw_n<2, 6>(NT, p, stream, gy))); break;
    case 8: CVX_TRY((launch_pw_n<2, 8>(NT, p, stream, gy))); break;
    case 12: CVX_TRY((launch_pw_n<1, 12>(NT, p, stream, gy))); break;
    default: CVX_TRY((launch_pw_n<1, 16>(NT, p, stream, gy))); break;
  }
  CVX_HIP(hipGetLastError());
  return 0;
}
