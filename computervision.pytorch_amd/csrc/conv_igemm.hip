// NHWC implicit-GEMM convolution for gfx950 (forward and data-gradient share this kernel).
//
//   out[m][n] = sum_{tap, c}  in[pixel(m, tap)][c] * wt[n][tap][c]
//
// * im2col-free: each K-step gathers a [128 pixels][32 k] slice of the virtual patch matrix straight
//   from the NHWC tensor (16-byte loads, zero-filled outside the image) into LDS, the matching
//   [NT*16 couts][32 k] weight slice beside it, and contracts them with v_mfma_f32_16x16x32_f16.
// * operand roles are swapped (weights = MFMA "A", pixels = MFMA "B") so every lane ends up with
//   4 consecutive output CHANNELS of one pixel -> 8-byte NHWC stores and wave-local channel sums.
// * LDS rows are 64 B (32 halves); the 16-byte slot of a row is XOR-swizzled with (row>>1)&3, which
//   makes both the ds_write_b128 staging pattern and the ds_read_b128 fragment pattern conflict-free
//   under the gfx950 lane-group rules (MI355X_MICROARCH.md, LDS section).
// * register double buffering: the global gather for K-step s+1 is issued before the MFMAs of step s;
//   two LDS buffers, one barrier per K-step.
#include <cstdlib>

#include "conv_igemm.h"

namespace {

constexpr int BM = 128;  // output pixels per workgroup (4 waves x 2 subtiles x 16)
constexpr int BK = 32;   // K per step = one MFMA

__device__ __forceinline__ int lds_off(int row, int slot) { return row * BK + ((slot ^ ((row >> 1) & 3)) << 3); }

template <int NT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
  constexpr int BROWS = NT * 16;
  constexpr int BL = (BROWS * 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) half_t sA[2][BM * BK];
  __shared__ __attribute__((aligned(16))) half_t sB[2][BROWS * BK];
  __shared__ ConvTap sTap[CVX_MAX_TAPS];
  __shared__ float sStat[4][BROWS][2];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int fr = lane & 15, fq = lane >> 4;
  const int nblk = blockIdx.y;
  const long long M = (long long)p.B * p.OH2 * p.OW2;
  const long long m_base = (long long)blockIdx.x * BM;

  if (tid < p.ntaps) sTap[tid] = p.taps[tid];

  // ---- per-thread gather coordinates: 2 pixel rows x one 8-channel group ----
  const int kg = tid & 3;
  const int arow0 = tid >> 2;
  const half_t* rowptr[2];
  int ih0[2], iw0[2];
  bool mvalid[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    long long m = m_base + arow0 + 64 * i;
    mvalid[i] = m < M;
    long long mm = mvalid[i] ? m : 0;
    int ow2 = (int)(mm % p.OW2);
    long long t = mm / p.OW2;
    int oh2 = (int)(t % p.OH2);
    int b = (int)(t / p.OH2);
    ih0[i] = oh2 * p.IS;
    iw0[i] = ow2 * p.IS;
    rowptr[i] = p.in + (long long)b * p.in_bstride;
  }
  int c = kg * 8, tap = 0;
  while (c >= p.Cin) {
    c -= p.Cin;
    ++tap;
  }
  const int nsteps = (p.ntaps * p.Cin + BK - 1) / BK;

  uint4 areg[2];
  uint4 breg[BL];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  __syncthreads();  // sTap visible

  auto gather = [&]() {
    const bool kvalid = tap < p.ntaps;
    ConvTap td = sTap[kvalid ? tap : 0];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int ih = ih0[i] + td.dh, iw = iw0[i] + td.dw;
      bool ok = kvalid && mvalid[i] && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
      areg[i] = ok ? *reinterpret_cast<const uint4*>(rowptr[i] + ((long long)ih * p.IW + iw) * p.in_ld + c) : zero4;
    }
#pragma unroll
    for (int j = 0; j < BL; ++j) {
      int brow = arow0 + 64 * j;
      int n = nblk * BROWS + brow;
      bool ok = kvalid && brow < BROWS && n < p.Cout;
      breg[j] = ok ? *reinterpret_cast<const uint4*>(p.wt + (long long)n * p.wt_ld + td.wtap * p.Cin + c) : zero4;
    }
    c += BK;
    while (c >= p.Cin) {
      c -= p.Cin;
      ++tap;
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(&sA[buf][lds_off(arow0 + 64 * i, kg)]) = areg[i];
#pragma unroll
    for (int j = 0; j < BL; ++j) {
      int brow = arow0 + 64 * j;
      if (brow < BROWS) *reinterpret_cast<uint4*>(&sB[buf][lds_off(brow, kg)]) = breg[j];
    }
  };

  f4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  gather();
  stage(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    const bool more = s + 1 < nsteps;
    if (more) gather();
    h8 xa[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) xa[i] = *reinterpret_cast<const h8*>(&sA[buf][lds_off(wave * 32 + i * 16 + fr, fq)]);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      h8 wb = *reinterpret_cast<const h8*>(&sB[buf][lds_off(j * 16 + fr, fq)]);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb, xa[i], acc[i][j], 0, 0, 0);
    }
    if (more) stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds pixel (fr) x channels 4*fq..4*fq+3 of every (i, j) tile ----
  long long out_off[2];
  long long res_off[2];
  bool pvalid[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    long long m = m_base + wave * 32 + i * 16 + fr;
    pvalid[i] = m < M;
    long long mm = pvalid[i] ? m : 0;
    int ow2 = (int)(mm % p.OW2);
    long long t = mm / p.OW2;
    int oh2 = (int)(t % p.OH2);
    int b = (int)(t / p.OH2);
    long long pix = (long long)(oh2 * p.OS + p.oph) * p.OWr + (ow2 * p.OS + p.opw);
    out_off[i] = (long long)b * p.out_bstride + pix * p.out_ld;
    res_off[i] = (long long)b * p.res_bstride + pix * p.res_ld;
  }

  if (p.epi == CVX_EPI_RAW_STATS) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n0 = nblk * BROWS + j * 16 + fq * 4;
      float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (pvalid[i]) {
          if (n0 < p.Cout) {
            h4 v = {(half_t)acc[i][j][0], (half_t)acc[i][j][1], (half_t)acc[i][j][2], (half_t)acc[i][j][3]};
            *reinterpret_cast<h4*>(p.out16 + out_off[i] + n0) = v;
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
          sStat[wave][j * 16 + fq * 4 + r][0] = a;
          sStat[wave][j * 16 + fq * 4 + r][1] = b2;
        }
      }
    }
    __syncthreads();
    for (int t = tid; t < BROWS * 2; t += 256) {
      int ch = t >> 1, which = t & 1;
      int n = nblk * BROWS + ch;
      if (n < p.Cout) {
        float v = sStat[0][ch][which] + sStat[1][ch][which] + sStat[2][ch][which] + sStat[3][ch][which];
        // a few replica slabs (blockIdx mod R) keep the float atomics spread over many addresses
        cvx_fix_atomic_add(&p.stats[((long long)(blockIdx.x % p.stats_replicas) * p.Cout + n) * 2 + which], v);
      }
    }
    return;
  }

#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n0 = nblk * BROWS + j * 16 + fq * 4;
    if (n0 >= p.Cout) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!pvalid[i]) continue;
      f4 v = acc[i][j];
      if (p.epi == CVX_EPI_AFFINE_SILU) {
        f4 sc = *reinterpret_cast<const f4*>(p.scale + n0);
        f4 sh = *reinterpret_cast<const f4*>(p.shift + n0);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = cvx_silu(v[r] * sc[r] + sh[r]);
        if (p.res) {
          h4 rr = *reinterpret_cast<const h4*>(p.res + res_off[i] + n0);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
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
}

template <int NT>
void launch_nt(const ConvParams& p, hipStream_t stream, dim3 grid) {
  hipLaunchKernelGGL(conv_igemm_kernel<NT>, grid, dim3(256), 0, stream, p);
}

}  // namespace

int cvx_conv_igemm_mblocks(long long M) { return (int)((M + BM - 1) / BM); }

unsigned long long* g_cvx_clk = nullptr;
int g_cvx_grid_div = 1;

int cvx_conv_igemm_launch(const ConvParams& p_in, hipStream_t stream, int* m_blocks) {
  ConvParams p = p_in;
  p.clk = g_cvx_clk;
  CVX_CHECK(p.Cin % 8 == 0 && p.in_ld % 8 == 0 && p.wt_ld % 8 == 0, "conv_igemm: channel counts must be multiples of 8");
  CVX_CHECK(p.Cout % 4 == 0 && p.out_ld % 4 == 0, "conv_igemm: Cout/out_ld must be multiples of 4");
  CVX_CHECK(p.ntaps >= 1 && p.ntaps <= CVX_MAX_TAPS, "conv_igemm: bad tap count");
  CVX_CHECK(((uintptr_t)p.in % 16) == 0 && ((uintptr_t)p.wt % 16) == 0, "conv_igemm: operands must be 16-byte aligned");
  const long long M = (long long)p.B * p.OH2 * p.OW2;
  CVX_CHECK(M > 0, "conv_igemm: empty output");
  static const bool force_v1 = getenv("CVX_CONV_V1") != nullptr;
  if (p.zeros && !force_v1) {
    if (m_blocks) *m_blocks = 0;
    if (p.nphase > 1) return cvx_conv_igemm_dma_launch(p, stream);  // merged phases: the DMA-ring kernel only
    if (cvx_conv_pw_supported(p)) return cvx_conv_pw_launch(p, stream);
    if (cvx_conv_halo_supported(p)) return cvx_conv_halo_launch(p, stream);
    return cvx_conv_igemm_dma_launch(p, stream);
  }
  const int tiles = (p.Cout + 15) / 16;
  static const int allowed[] = {1, 2, 3, 4, 5, 6, 8};
  int gy = (tiles + 7) / 8;
  int want = (tiles + gy - 1) / gy;
  int NT = 8;
  for (int a : allowed)
    if (a >= want) {
      NT = a;
      break;
    }
  gy = (tiles + NT - 1) / NT;
  const int gx = cvx_conv_igemm_mblocks(M);
  if (m_blocks) *m_blocks = gx;
  dim3 grid(gx, gy);
  switch (NT) {
    case 1: launch_nt<1>(p, stream, grid); break;
    case 2: launch_nt<2>(p, stream, grid); break;
    case 3: launch_nt<3>(p, stream, grid); break;
    case 4: launch_nt<4>(p, stream, grid); break;
    case 5: launch_nt<5>(p, stream, grid); break;
    case 6: launch_nt<6>(p, stream, grid); break;
    default: launch_nt<8>(p, stream, grid); break;
  }
  CVX_HIP(hipGetLastError());
  return 0;
}
