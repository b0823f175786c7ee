// Pieces shared by the LDS-DMA convolution kernels (conv_igemm_dma.hip, conv_halo.hip): counted vmcnt waits,
// the 64-byte-row XOR swizzle, and the common epilogue (BN-stat partials / folded BN+SiLU / bias / plain-accumulate).
#pragma once
#include "conv_igemm.h"

namespace cvx_tile {

constexpr int BK = 32;

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;

__device__ __forceinline__ int lds_row_off(int row, int slot) { return row * BK + ((slot ^ ((row >> 1) & 3)) << 3); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// raw s_barrier (no vmcnt drain, unlike __syncthreads) fenced for the COMPILER on both sides: without the clobbers hipcc
// may hoist the LDS fragment reads of the next phase above the barrier -- i.e. between this wave's own vmcnt wait and
// the other waves' -- and read DMA pieces that have not landed yet (seen as a few-percent error in one tile config).
__device__ __forceinline__ void workgroup_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
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

// Epilogue for a block tile of WM x (MT sub-tiles of 16 pixels) by WN x (NTW sub-tiles of 16 channels).
// Lane (fr, fq) holds pixel fr of sub-tile i and channels 4*fq..4*fq+3 of sub-tile j in acc[i][j].
// sStat: WM*BN*2 floats of LDS scratch.  All threads of the block must call it (it synchronises in RAW_STATS mode).
template <int WM, int WN, int MT, int NTW>
__device__ __forceinline__ void epilogue(const ConvParams& p, f4 (&acc)[MT][NTW], const long long (&out_off)[MT], const long long (&res_off)[MT],
                                         const bool (&pvalid)[MT], int wm, int wn, int fr, int fq, int nblk, float* sStat, int tid) {
  constexpr int BN = 16 * NTW * WN;
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
        cvx_fix_atomic_add(&p.stats[((long long)(blockIdx.x % p.stats_replicas) * p.Cout + n) * 2 + which], v);
      }
    }
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

}  // namespace cvx_tile
