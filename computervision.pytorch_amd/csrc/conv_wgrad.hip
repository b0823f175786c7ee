// Weight gradient of an NHWC convolution on MFMA (gfx950).
//
//   dW[co][tap][ci] = sum_m dy[m][co] * x[pixel(m, tap)][ci]          (m over B*OH*OW output pixels)
//
// GEMM view: D[co][j] with j = tap*cin_pad16 + ci, reduction over pixels.  Both operands are stored
// pixel-major in memory (NHWC), i.e. the reduction index is the SLOW index of both -- so the tiles are
// staged into LDS as they lie ([pixel][channel], 16-byte loads) and the MFMA fragments are fetched
// with the gfx950 transposing LDS read ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group).
// The pixel range is split over the workgroups (nsplit ranges); every split writes its own fp32 slab (plain stores,
// deterministic), summed into the gradient arena by cvx_reduce_slabs.
#include "conv_igemm.h"

namespace {

constexpr int PK = 128;   // pixels per iteration (four MFMA K-steps between barriers)
constexpr int RPAD = 8;   // halves of padding per LDS row (keeps tr reads <= 2-way, rows 16-B aligned)

typedef s4 __attribute__((address_space(3))) * lds_s4_ptr;

__device__ __forceinline__ h8 tr_frag(const half_t* tile, int row_stride, int p0, int c0, int lane) {
  // 16-lane group g = lane>>4 covers pixels p0 + 8g .. p0 + 8g + 7, channels c0 .. c0+15
  const int lg = lane & 15, fq = lane >> 4;
  const half_t* a = tile + (p0 + 8 * fq + (lg >> 2)) * row_stride + c0 + 4 * (lg & 3);
  s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(a));
  s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(a + 4 * row_stride));
  s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(h8, v);
}

template <int TI, int TJ, int WI, int WJ>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p, long long pix_per_split, int gx, int gy) {
  static_assert(WI * WJ == 4, "4 waves");
  constexpr int CO_B = 16 * TI * WI;
  constexpr int J_B = 16 * TJ * WJ;
  constexpr int SD = CO_B + RPAD;  // row strides (halves)
  constexpr int SX = J_B + RPAD;
  constexpr int DG = CO_B / 8, XG = J_B / 8;  // 8-channel groups per row
  __shared__ __attribute__((aligned(16))) half_t sD[PK * SD];
  __shared__ __attribute__((aligned(16))) half_t sX[PK * SX];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave / WJ, wj = wave % WJ;
  // XCD-aware block order: workgroups are dealt to the 8 XCDs round-robin by linear id, and every (co, j) tile of one
  // pixel split reads the same dy / x pixels -- so all tiles of split z are given ids congruent to z mod 8: they share
  // one XCD's L2 instead of fetching the operands into eight.
  const int ntiles = gx * gy;
  const int lin = blockIdx.x;
  const int q = lin >> 3;
  const int tile = q % ntiles;
  const int bz = (q / ntiles) * 8 + (lin & 7);
  if (bz >= p.nsplit) return;
  const int bx = tile % gx, by = tile / gx;
  const int co0 = bx * CO_B;
  const int j0 = by * J_B;
  const long long ohw = (long long)p.OH * p.OW;
  const long long M = (long long)p.B * ohw;
  const long long m_begin = (long long)bz * pix_per_split;
  const long long m_end = min(M, m_begin + pix_per_split);
  const int Jtot = p.ntaps * p.cin_pad16;

  // ---- fixed per-thread column group of the X tile ----
  const int xcg = tid % XG, xslot = tid / XG;
  constexpr int XSLOTS = 256 / XG;
  const int jx = j0 + xcg * 8;
  const int xt = jx / p.cin_pad16;
  const int xci = jx - xt * p.cin_pad16;
  const bool xcol_ok = xslot < XSLOTS && jx < Jtot && xci < p.Cin;
  int xdh = 0, xdw = 0;
  if (xcol_ok) {
    ConvTap td = p.taps[xt];
    xdh = td.dh;
    xdw = td.dw;
  }
  // ---- fixed per-thread column group of the dY tile ----
  const int dcg = tid % DG, dslot = tid / DG;
  constexpr int DSLOTS = 256 / DG;
  const bool dcol_ok = dslot < DSLOTS && (co0 + dcg * 8) < p.Cout;

  f4 acc[TI][TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};

  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  constexpr int DN = (PK + DSLOTS - 1) / DSLOTS, XN = (PK + XSLOTS - 1) / XSLOTS;
  uint4 dreg[DN], xreg[XN];
  // global -> registers for one 64-pixel chunk (issued one chunk ahead of the MFMAs: software prefetch)
  auto fetch = [&](long long mc) {
#pragma unroll
    for (int k = 0; k < DN; ++k) {
      const int pp = dslot + k * DSLOTS;
      const long long m = mc + pp;
      uint4 v = zero4;
      if (dslot < DSLOTS && pp < PK && dcol_ok && m < m_end) {
        const unsigned mu = (unsigned)m, b = mu / (unsigned)ohw, pix = mu - b * (unsigned)ohw;
        v = *reinterpret_cast<const uint4*>(p.dy + (long long)b * p.dy_bstride + (long long)pix * p.dy_ld + co0 + dcg * 8);
      }
      dreg[k] = v;
    }
#pragma unroll
    for (int k = 0; k < XN; ++k) {
      const int pp = xslot + k * XSLOTS;
      const long long m = mc + pp;
      uint4 v = zero4;
      if (xslot < XSLOTS && pp < PK && xcol_ok && m < m_end) {
        const unsigned mu = (unsigned)m, tq = mu / (unsigned)p.OW;
        int ow = (int)(mu - tq * (unsigned)p.OW);
        int b = (int)(tq / (unsigned)p.OH);
        int oh = (int)(tq - (unsigned)b * (unsigned)p.OH);
        int ih = oh * p.stride + xdh, iw = ow * p.stride + xdw;
        if ((unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW)
          v = *reinterpret_cast<const uint4*>(p.x + (long long)b * p.x_bstride + ((long long)ih * p.IW + iw) * p.x_ld + xci);
      }
      xreg[k] = v;
    }
  };
  auto park = [&]() {  // registers -> LDS tiles
#pragma unroll
    for (int k = 0; k < DN; ++k) {
      const int pp = dslot + k * DSLOTS;
      if (dslot < DSLOTS && pp < PK) *reinterpret_cast<uint4*>(&sD[pp * SD + dcg * 8]) = dreg[k];
    }
#pragma unroll
    for (int k = 0; k < XN; ++k) {
      const int pp = xslot + k * XSLOTS;
      if (xslot < XSLOTS && pp < PK) *reinterpret_cast<uint4*>(&sX[pp * SX + xcg * 8]) = xreg[k];
    }
  };
  if (m_begin < m_end) fetch(m_begin);
  for (long long mc = m_begin; mc < m_end; mc += PK) {
    park();
    __syncthreads();
    if (mc + PK < m_end) fetch(mc + PK);  // in flight while the MFMAs below run
#pragma unroll
    for (int ks = 0; ks < PK / 32; ++ks) {
      h8 fa[TI], fb[TJ];
#pragma unroll
      for (int a = 0; a < TI; ++a) fa[a] = tr_frag(sD, SD, ks * 32, (wi * TI + a) * 16, lane);
#pragma unroll
      for (int b = 0; b < TJ; ++b) fb[b] = tr_frag(sX, SX, ks * 32, (wj * TJ + b) * 16, lane);
#pragma unroll
      for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- write the slab: lane holds column j = ..+(lane&15), rows co = ..+4*(lane>>4)+r ----
  float* slab = p.slabs + (long long)bz * p.Cout * Jtot;
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int a = 0; a < TI; ++a) {
#pragma unroll
    for (int b = 0; b < TJ; ++b) {
      int j = j0 + (wj * TJ + b) * 16 + fr;
      if (j >= Jtot) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int co = co0 + (wi * TI + a) * 16 + fq * 4 + r;
        if (co < p.Cout) slab[(long long)co * Jtot + j] = acc[a][b][r];
      }
    }
  }
}

template <int TI, int TJ, int WI, int WJ>
void launch_cfg(const WgradParams& p, hipStream_t st) {
  constexpr int CO_B = 16 * TI * WI, J_B = 16 * TJ * WJ;
  const long long M = (long long)p.B * p.OH * p.OW;
  long long per = (M + p.nsplit - 1) / p.nsplit;
  per = ((per + PK - 1) / PK) * PK;
  const int gx = cvx_cdiv(p.Cout, CO_B), gy = cvx_cdiv(p.ntaps * p.cin_pad16, J_B);
  const int zpad = (p.nsplit + 7) / 8 * 8;  // surplus workgroups (split >= nsplit) exit at once
  hipLaunchKernelGGL((conv_wgrad_kernel<TI, TJ, WI, WJ>), dim3(gx * gy * zpad), dim3(256), 0, st, p, per, gx, gy);
}

}  // namespace

// tile of the generic kernel for a layer (the engine sizes its pixel splits and slabs with it)
void cvx_conv_wgrad_tile(int cout, int jtot, int* co_b, int* j_b) {
  static const int wide = cvx_tune_int("CVX_WGRAD_WIDE", 1);
  static const long long wide_min = cvx_tune_int("CVX_WGRAD_WIDE_MIN", 512 * 1024);  // below (YOLOv8-n's layers) the 64 x 64 tile measured 1 % faster
  if (cout <= 16) {
    *co_b = 16;
    *j_b = 192;
  } else if (cout <= 32) {
    *co_b = 32;
    *j_b = 128;
  } else if (cout % 64 != 0 && (cout % 48 == 0 || cout <= 96)) {
    *co_b = 48;
    *j_b = 128;
  } else if (wide && cout % 128 == 0 && jtot >= 512 && (long long)cout * jtot >= wide_min) {  // ResNet-sized layers: twice the arithmetic intensity per staged byte
    *co_b = 128;
    *j_b = 128;
  } else {
    *co_b = 64;
    *j_b = 64;
  }
}

int cvx_conv_wgrad_launch(const WgradParams& p, hipStream_t st) {
  CVX_CHECK(p.Cin % 8 == 0 && p.x_ld % 8 == 0 && p.dy_ld % 8 == 0 && p.Cout % 8 == 0, "wgrad: channels must be multiples of 8");
  CVX_CHECK(p.cin_pad16 % 16 == 0 && p.cin_pad16 >= p.Cin, "wgrad: cin_pad16");
  CVX_CHECK(p.nsplit >= 1 && p.ntaps >= 1 && p.ntaps <= CVX_MAX_TAPS, "wgrad: nsplit/ntaps");
  CVX_CHECK(((uintptr_t)p.x % 16) == 0 && ((uintptr_t)p.dy % 16) == 0, "wgrad: operands must be 16-byte aligned");
  if (cvx_conv_wgrad_k3_supported(p)) return cvx_conv_wgrad_k3_launch(p, st);
  if (cvx_conv_wgrad_stream_supported(p)) return cvx_conv_wgrad_stream_launch(p, st);
  if (cvx_conv_wgrad_halo_supported(p)) return cvx_conv_wgrad_halo_launch(p, st);
  if (cvx_conv_wgrad_gemm_supported(p)) return cvx_conv_wgrad_gemm_launch(p, st);
  int co_b, j_b;
  cvx_conv_wgrad_tile(p.Cout, p.ntaps * p.cin_pad16, &co_b, &j_b);
  if (co_b == 16)
    launch_cfg<1, 3, 1, 4>(p, st);
  else if (co_b == 32)
    launch_cfg<2, 2, 1, 4>(p, st);
  else if (co_b == 48)
    launch_cfg<3, 2, 1, 4>(p, st);
  else if (co_b == 128)
    launch_cfg<4, 4, 2, 2>(p, st);
  else
    launch_cfg<2, 2, 2, 2>(p, st);
  CVX_HIP(hipGetLastError());
  return 0;
}
