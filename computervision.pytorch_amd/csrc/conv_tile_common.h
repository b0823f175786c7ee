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

// phase timestamp (100 MHz wall clock) of thread 0 of a block, slot 0..7
__device__ __forceinline__ void clk_mark(const ConvParams& p, int slot) {
  if (p.clk && threadIdx.x == 0)
    {
    unsigned long long* q = p.clk + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 8;
    q[slot] = wall_clock64();
    if (slot == 0) q[5] = clock64();
    if (slot == 4) q[6] = clock64();
  }
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
// RAW_STATS mode only stores the raw fp32 outputs and adds the lane's values into st1/st2 (per-lane running sums of y and
// y^2 for its 4 channels of every j); a persistent workgroup calls this once per tile and stats_flush once at the end.
template <int WM, int WN, int MT, int NTW>
__device__ __forceinline__ void epilogue_tile(const ConvParams& p, f4 (&acc)[MT][NTW], const long long (&out_off)[MT],
                                              const long long (&res_off)[MT], const bool (&pvalid)[MT], int wn, int fq, int nblk,
                                              f4 (&st1)[NTW], f4 (&st2)[NTW]) {
  constexpr int BN = 16 * NTW * WN;
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int n0 = nblk * BN + (wn * NTW + j) * 16 + fq * 4;
    if (n0 >= p.Cout) continue;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (!pvalid[i]) continue;
      f4 v = acc[i][j];
      if (p.epi == CVX_EPI_RAW_STATS) {
        cvx_store_raw4(p, out_off[i] + n0, v);  // fp32: normalisation reads the un-rounded accumulators (fp16 for CVX_OPF_RAW_F16 layers)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          st1[j][r] += v[r];
          st2[j][r] += v[r] * v[r];
        }
        continue;
      }
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
}

// RAW_STATS: folds the per-lane running sums over the 16 pixel lanes (DPP), over the WM pixel waves (LDS scratch sStat,
// WM*BN*2 floats) and adds the workgroup totals to one replica slab with fixed-point atomics.  All threads must call it.
template <int WM, int WN, int NTW>
__device__ __forceinline__ void stats_flush(const ConvParams& p, const f4 (&st1)[NTW], const f4 (&st2)[NTW], int wm, int wn, int fr, int fq,
                                            int nblk, float* sStat, int tid) {
  constexpr int BN = 16 * NTW * WN;
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int chl = (wn * NTW + j) * 16 + fq * 4;  // channel inside the block's BN range
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = cvx_wave_sum16(st1[j][r]), b2 = cvx_wave_sum16(st2[j][r]);
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
}

}  // namespace cvx_tile
