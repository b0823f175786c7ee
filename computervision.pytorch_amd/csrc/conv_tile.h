// Row-band convolution kernel (conv_tile.hip): launch geometry shared by the host logic and the kernel instantiations, which are spread
// over two translation units (conv_tile_k1 / k2.hip: pixel groups per wave 1-2, 3-4) so that they compile in parallel.
#pragma once
#include "conv_tile_common.h"

namespace cvx_tile_k {
using namespace cvx_tile;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// waves per workgroup: two per SIMD.  With one (the first version) nothing overlapped a wave's LDS requests and its MFMAs: measured 250
// cycles per K-step for 14 MFMAs alone, +280 with its 9 fragment requests, however they were placed or prefetched.
constexpr int kTileWaves = 8;

struct TileArgs {
  int TR, tiles_per_img, ntiles, NB;
  int P, pow2, logP, sh, swmask, phmask;  // 16-byte units per pixel; XOR swizzle f(pc) = (pc >> sh) & swmask (P a power of two; else swmask = 0, phmask = ~0)
  int SPT, NSTEPS, KSC, NCH, R;   // K-steps per tap, in all, per chunk; chunks; ring slots (R > NCH: every chunk has a slot of its own)
  int CP, PW;                     // 1-KiB DMA pieces per chunk, per wave and chunk
  int PP, PPW;                    // pieces of the patch, per wave
  int units, Wp;                  // 16-byte units of the patch; patch width W + 2
  unsigned magic_wp, magic_w, magic_p;  // ceil(2^32 / d) for d = W + 2, W, P
  int wring_off, CB, stat_off;    // LDS byte offsets / bytes per ring slot
  unsigned in_records;            // bytes of the input view (buffer descriptor range)
  const half_t* wpk;              // packed weights
  int dbg;                        // tuning build (CVX_TILE_DBG): 1 = drain every DMA before the K loop, 8 = dump block 0's LDS image after the first barrier and stop (tools/tile_lds_dump.py)
};

template <int OFF>
__device__ __forceinline__ h8 lds_frag(unsigned a) {
  h8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}
template <int NTW>
__device__ __forceinline__ void frag_load_w(h8 (&w)[NTW], unsigned a) {  // NTW channel tiles, 1 KiB apart
  if constexpr (NTW >= 1) w[0] = lds_frag<0>(a);
  if constexpr (NTW >= 2) w[1] = lds_frag<1024>(a);
  if constexpr (NTW >= 3) w[2] = lds_frag<2048>(a);
  if constexpr (NTW >= 4) w[3] = lds_frag<3072>(a);
  if constexpr (NTW >= 5) w[4] = lds_frag<4096>(a);
  if constexpr (NTW >= 6) w[5] = lds_frag<5120>(a);
  if constexpr (NTW >= 7) w[6] = lds_frag<6144>(a);
  if constexpr (NTW >= 8) w[7] = lds_frag<7168>(a);
  if constexpr (NTW >= 9) w[8] = lds_frag<8192>(a);
}

// waits until at most n (wave-uniform, 0 .. 20; more: 0) vector-memory operations are outstanding.  A short compare chain on purpose: a
// 64-way switch compiled to ~750 instructions per use and cost 1,300 cycles per chunk boundary (measured with the MFMAs and LDS reads
// ablated: 5 us of a 7-us K loop).
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
#define CVX_WV(N) else if (n == N) wait_vmcnt<N>();
  if (n <= 0 || n > 20) wait_vmcnt<0>();
  CVX_WV(1) CVX_WV(2) CVX_WV(3) CVX_WV(4) CVX_WV(5) CVX_WV(6) CVX_WV(7) CVX_WV(8) CVX_WV(9) CVX_WV(10)
  CVX_WV(11) CVX_WV(12) CVX_WV(13) CVX_WV(14) CVX_WV(15) CVX_WV(16) CVX_WV(17) CVX_WV(18) CVX_WV(19) CVX_WV(20)
#undef CVX_WV
}

}  // namespace cvx_tile_k

// one per translation unit: launches conv_tile_kernel<MT, NTW> for its MT values
int cvx_conv_tile_launch_k1(int MT, int NTW, const ConvParams& p, const cvx_tile_k::TileArgs& a, int lds, hipStream_t st);
int cvx_conv_tile_launch_k2(int MT, int NTW, const ConvParams& p, const cvx_tile_k::TileArgs& a, int lds, hipStream_t st);
