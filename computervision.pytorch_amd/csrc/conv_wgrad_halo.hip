// Weight gradient of a 3x3 / stride-1 / pad-1 NHWC convolution with the whole (co-block x 9 taps x ci-block) result
// tile held in registers (gfx950).
//
//   dW[co][tap][ci] = sum over pixels  dy[p][co] * x[p + tap][ci]
//
// The generic kernel (conv_wgrad.hip) gives a workgroup a 48..64 x 64..128 corner of the [Cout][9*Cin] result and makes
// it re-gather dy and the nine shifted copies of x for every corner: 13x the algorithmic bytes through L2 for an
// 80 -> 80 layer.  Here a workgroup owns 16*CO_T output channels x ALL nine taps x 16*CI_T input channels -- up to
// 80 x 720 fp32 = 240 accumulator registers per lane -- and walks its share of the 8 x 16-pixel tiles:
//   * per tile the 10 x 18 x ci-block halo patch of x and the 8 x 16 x co-block tile of dy are loaded ONCE (16-byte
//     global loads issued one tile ahead, parked in LDS pixel-major as they lie in memory);
//   * the nine taps are nine shifted views of that patch: the MFMA operands (reduction index = pixel = the slow index
//     of both tiles) are fetched with the transposing LDS read ds_read_b64_tr_b16, 8 consecutive pixels of one patch
//     row per 16-lane group, so a tap shift is just an address offset;
//   * four waves split the 9*CI_T column tiles, every wave keeps all CO_T row tiles: CO_T + JW fragment reads feed
//     CO_T*JW MFMAs per 32-pixel K-step.
// The pixel tiles are split over blockIdx.z; each split writes its own fp32 slab (plain stores, deterministic), summed by
// cvx_reduce_slabs like the generic kernel's.
#include "conv_igemm.h"

namespace {

constexpr int TH = 8;     // tile rows (x 16 columns = 128 pixels = four 32-pixel MFMA K-steps)
constexpr int PW = 18;    // patch width
constexpr int RPAD = 8;   // halves of padding per LDS pixel row (keeps the tr reads <= 2-way, rows 16-B aligned)

typedef s4 __attribute__((address_space(3))) * lds_s4_ptr;

// fragment of a pixel-major LDS tile: 16-lane group g = lane>>4 takes the 8 consecutive pixel rows starting at
// base + g-dependent offset (caller), channels c0 .. c0+15; two transposing reads of 4 pixels x 16 channels
__device__ __forceinline__ h8 tr_frag8(const half_t* first_pixel, int row_stride, int c0, int lg) {
  const half_t* a = first_pixel + (lg >> 2) * row_stride + c0 + 4 * (lg & 3);
  s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(a));
  s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(a + 4 * row_stride));
  s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(h8, v);
}

// TPB taps per workgroup: 9 (small tiles: everything in one workgroup) or 3 (one kernel row dh per workgroup: a third of
// the accumulators -- two waves per SIMD and room to pipeline the fragment reads -- and three times the workgroups per
// pixel split, i.e. a third of the slab volume for the same parallelism; the patch and the dy tile are loaded 3 times).
template <int CO_T, int CI_T, int TPB>
__global__ __launch_bounds__(256) void conv_wgrad_halo_kernel(const WgradParams p, int tiles_x, int tiles_y, int total_tiles, int tiles_per_split,
                                                              int gy_ci, int gx, int gy, int nsplit) {
  constexpr int CO_B = 16 * CO_T, CI_B = 16 * CI_T;
  constexpr int SD = CO_B + RPAD, SX = CI_B + RPAD;  // LDS pixel-row strides (halves)
  constexpr int NJ = TPB * CI_T;                     // column tiles (tap, ci tile) of this workgroup
  constexpr int JW = (NJ + 3) / 4;                   // per wave
  constexpr int XU = (TH + 2) * PW * (CI_B / 8);     // 16-byte units of the x patch
  constexpr int DU = TH * 16 * (CO_B / 8);           // ... of the dy tile
  constexpr int XN = (XU + 255) / 256, DN = (DU + 255) / 256;
  __shared__ __attribute__((aligned(16))) half_t sX[(TH + 2) * PW * SX];
  __shared__ __attribute__((aligned(16))) half_t sD[TH * 16 * SD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lg = lane & 15, fq = lane >> 4;
  // Block -> (output-channel block, input-channel block x tap group, pixel split).  The gx * gy workgroups of one split read the same
  // pixel tiles (x three times over when the taps are split in three) and consecutive workgroup ids go to consecutive XCDs, each with an
  // L2 of its own: as a 3-D grid the readers of one tile sat on different XCDs and every one fetched it from HBM -- measured 2.0 GB per
  // step for 0.7 GB of operands (profiles/r04_conv_traffic.json).  Here the id is cut so that all workgroups of a split share id % 8.
  const int G = gx * gy;
  const int q = blockIdx.x / (8 * G), r8 = blockIdx.x - q * (8 * G);
  const int bz = q * 8 + (r8 & 7), g = r8 >> 3;
  if (bz >= nsplit) return;
  const int bx = g % gx, by = g / gx;
  const int co0 = bx * CO_B, ci0 = (by % gy_ci) * CI_B;
  const int tap0 = (by / gy_ci) * TPB;
  const int t_begin = bz * tiles_per_split;
  const int t_end = min(total_tiles, t_begin + tiles_per_split);
  const int H = p.OH, W = p.OW;  // stride 1, pad 1: input and output sizes agree

  f4 acc[CO_T][JW];
#pragma unroll
  for (int a = 0; a < CO_T; ++a)
#pragma unroll
    for (int j = 0; j < JW; ++j) acc[a][j] = f4{0.f, 0.f, 0.f, 0.f};

  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  uint4 xreg[XN], dreg[DN];
  // global -> registers for one tile (issued one tile ahead of the MFMAs)
  auto fetch = [&](int tile) {
    const int tx = tile % tiles_x;
    const int t2 = tile / tiles_x;
    const int ty = t2 % tiles_y;
    const int b = t2 / tiles_y;
    const int y0 = ty * TH, x0 = tx * 16;
    const half_t* xb = p.x + (long long)b * p.x_bstride;
    const half_t* db = p.dy + (long long)b * p.dy_bstride;
#pragma unroll
    for (int k = 0; k < XN; ++k) {
      const int u = tid + k * 256;
      uint4 v = zero4;
      if (u < XU) {
        const int pix = u / (CI_B / 8), cg = u - pix * (CI_B / 8);
        const int pr = pix / PW, pc = pix - pr * PW;
        const int iy = y0 - 1 + pr, ix = x0 - 1 + pc, ci = ci0 + cg * 8;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && ci < p.Cin)
          v = *reinterpret_cast<const uint4*>(xb + ((long long)iy * W + ix) * p.x_ld + ci);
      }
      xreg[k] = v;
    }
#pragma unroll
    for (int k = 0; k < DN; ++k) {
      const int u = tid + k * 256;
      uint4 v = zero4;
      if (u < DU) {
        const int pix = u / (CO_B / 8), cg = u - pix * (CO_B / 8);
        const int oy = y0 + (pix >> 4), ox = x0 + (pix & 15), co = co0 + cg * 8;
        if (oy < H && ox < W && co < p.Cout) v = *reinterpret_cast<const uint4*>(db + ((long long)oy * W + ox) * p.dy_ld + co);
      }
      dreg[k] = v;
    }
  };
  auto park = [&]() {  // registers -> LDS tiles
#pragma unroll
    for (int k = 0; k < XN; ++k) {
      const int u = tid + k * 256;
      if (u < XU) {
        const int pix = u / (CI_B / 8), cg = u - pix * (CI_B / 8);
        *reinterpret_cast<uint4*>(&sX[pix * SX + cg * 8]) = xreg[k];
      }
    }
#pragma unroll
    for (int k = 0; k < DN; ++k) {
      const int u = tid + k * 256;
      if (u < DU) {
        const int pix = u / (CO_B / 8), cg = u - pix * (CO_B / 8);
        *reinterpret_cast<uint4*>(&sD[pix * SD + cg * 8]) = dreg[k];
      }
    }
  };

  // this wave's column tiles: jt = wave*JW + jj -> (tap, ci tile); patch offset of the tap in pixels, channel offset in halves
  int xoff[JW];
  bool jok[JW];
#pragma unroll
  for (int jj = 0; jj < JW; ++jj) {
    const int jt = wave * JW + jj;
    jok[jj] = jt < NJ;
    const int tl = jok[jj] ? jt / CI_T : 0, cit = jok[jj] ? jt - tl * CI_T : 0;
    const int tap = tap0 + tl;
    const int dh = tap / 3, dw = tap - dh * 3;  // already +1 (standard 3x3 table: dh-1, dw-1)
    xoff[jj] = (dh * PW + dw) * SX + cit * 16;
  }
  // pixel group of this 16-lane group inside a K-step (2 tile rows x 16 columns): row fq>>1, first column 8*(fq&1)
  const int grow = fq >> 1, gcol = 8 * (fq & 1);

  if (t_begin < t_end) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    park();
    __syncthreads();
    if (tile + 1 < t_end) fetch(tile + 1);  // in flight while the MFMAs below run
#pragma unroll
    for (int ks = 0; ks < TH / 2; ++ks) {
      const int r = 2 * ks + grow;  // tile row of this lane group
      h8 fa[CO_T];
#pragma unroll
      for (int a = 0; a < CO_T; ++a) fa[a] = tr_frag8(sD + (r * 16 + gcol) * SD, SD, a * 16, lg);
      const half_t* xrow = sX + (r * PW + gcol) * SX;  // + tap offset (dh*PW + dw pixels)
#pragma unroll
      for (int jj = 0; jj < JW; ++jj) {
        if (!jok[jj]) continue;  // wave-uniform
        const h8 fb = tr_frag8(xrow + xoff[jj], SX, 0, lg);
#pragma unroll
        for (int a = 0; a < CO_T; ++a) acc[a][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a], fb, acc[a][jj], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- write the slab: lane holds column j = ..+(lane&15), rows co = ..+4*(lane>>4)+r ----
  const int Jtot = p.ntaps * p.cin_pad16;
  float* slab = p.slabs + (long long)bz * p.Cout * Jtot;
#pragma unroll
  for (int jj = 0; jj < JW; ++jj) {
    const int jt = wave * JW + jj;
    if (jt >= NJ) continue;
    const int tl = jt / CI_T, cit = jt - tl * CI_T;
    const int tap = tap0 + tl;
    const int ci = ci0 + cit * 16 + lg;
    if (ci >= p.cin_pad16) continue;
    const int j = tap * p.cin_pad16 + ci;
#pragma unroll
    for (int a = 0; a < CO_T; ++a) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + a * 16 + fq * 4 + r;
        if (co < p.Cout) slab[(long long)co * Jtot + j] = acc[a][jj][r];
      }
    }
  }
}

template <int CO_T, int CI_T>
int launch_wh(const WgradParams& p, hipStream_t st, int gx, int gy) {
  const int tiles_x = (p.OW + 15) / 16, tiles_y = (p.OH + TH - 1) / TH;
  const int total = tiles_x * tiles_y * p.B;
  const int per = (total + p.nsplit - 1) / p.nsplit;
  const int ns8 = (p.nsplit + 7) / 8 * 8;  // ids are dealt to the splits in groups of 8 (one per XCD): the last group's spare workgroups exit at once
  if constexpr (CO_T * CI_T >= 8) {  // must mirror cvx_conv_wgrad_halo_grid
    hipLaunchKernelGGL((conv_wgrad_halo_kernel<CO_T, CI_T, 3>), dim3(gx * gy * 3 * ns8), dim3(256), 0, st, p, tiles_x, tiles_y, total, per, gy, gx, gy * 3, p.nsplit);
  } else {
    hipLaunchKernelGGL((conv_wgrad_halo_kernel<CO_T, CI_T, 9>), dim3(gx * gy * ns8), dim3(256), 0, st, p, tiles_x, tiles_y, total, per, gy, gx, gy, p.nsplit);
  }
  return 0;
}

template <int CO_T>
int launch_wh_ci(int cit, const WgradParams& p, hipStream_t st, int gx, int gy) {
  switch (cit) {
    case 1: return launch_wh<CO_T, 1>(p, st, gx, gy);
    case 2: return launch_wh<CO_T, 2>(p, st, gx, gy);
    case 4: return launch_wh<CO_T, 4>(p, st, gx, gy);
    default: return launch_wh<CO_T, 5>(p, st, gx, gy);
  }
}

// tiles of 16 channels per workgroup for a channel count: one block when it fits 80, else blocks of 64 (multiples of 64) or 80
int pick_tiles(int c) {
  if (c <= 16) return 1;
  if (c <= 32) return 2;
  if (c <= 64) return 4;
  if (c <= 80) return 5;
  return (c % 64 == 0) ? 4 : 5;
}

}  // namespace

bool cvx_conv_wgrad_halo_supported(const WgradParams& p) {
  static const bool off = cvx_tune_set("CVX_NO_WGRAD_HALO");
  // wide layers (Cin * Cout >= 256 * 256) take the generic kernel's 128 x 128 tile: 3-4x faster there (DeepLabv3+ R101: 256 -> 256 at
  // 33 x 33: 260 -> 70 us, 304 -> 256 at 129 x 129: 3.1 -> 1.0 ms); the register-tile kernel is built for YOLO's narrow layers
  // ... and from 128 x 128 channels up the GEMM-shaped kernel (conv_wgrad_gemm.hip) is ahead: 339 vs 427 us at 150 x 150, 100 vs 119 us at 80 x 80
  static const int max_c = cvx_tune_int("CVX_WH_MAX_C", 16383);
  return !off && p.std3x3 && p.stride == 1 && p.ntaps == 9 && p.Cin >= 16 && p.IH == p.OH && p.IW == p.OW && (long long)p.Cin * p.Cout <= max_c;
}

// workgroups per pixel split: gx channel blocks x gy (input-channel blocks x tap groups)
void cvx_conv_wgrad_halo_grid(int cout, int cin, int* gx, int* gy) {
  *gx = cvx_cdiv(cout, 16 * pick_tiles(cout));
  *gy = cvx_cdiv(cin, 16 * pick_tiles(cin)) * (pick_tiles(cout) * pick_tiles(cin) >= 8 ? 3 : 1);
}

int cvx_conv_wgrad_halo_tiles(int B, int OH, int OW) { return ((OW + 15) / 16) * ((OH + TH - 1) / TH) * B; }

int cvx_conv_wgrad_halo_launch(const WgradParams& p, hipStream_t st) {
  const int cot = pick_tiles(p.Cout), cit = pick_tiles(p.Cin);
  const int gx = cvx_cdiv(p.Cout, 16 * cot), gy = cvx_cdiv(p.Cin, 16 * cit);  // gy: input-channel blocks only
  switch (cot) {
    case 1: CVX_TRY((launch_wh_ci<1>(cit, p, st, gx, gy))); break;
    case 2: CVX_TRY((launch_wh_ci<2>(cit, p, st, gx, gy))); break;
    case 4: CVX_TRY((launch_wh_ci<4>(cit, p, st, gx, gy))); break;
    default: CVX_TRY((launch_wh_ci<5>(cit, p, st, gx, gy))); break;
  }
  CVX_HIP(hipGetLastError());
  return 0;
}
