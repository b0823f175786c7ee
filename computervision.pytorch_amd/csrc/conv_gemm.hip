// GEMM-shaped implicit convolution (gfx950) for every forward / data-gradient launch with ~100 output channels and more (VGG, ResNet-101, ASPP,
// DLA-34, YOLOv7, YOLOv8-s and up, YOLOv8-n's 96+-channel layers).  M = pixels, N = output channels, K = taps * Cin (Cin % 8 == 0: ragged last chunk).
//
//   * 512 threads = 8 waves as 4 (pixels) x 2 (channels); macro tile BM x BN = (4 * MT * 32) x (2 * NT * 32): 256 x 256, 256 x 128,
//     128 x 256 or 128 x 128; every wave owns MT x NT tiles of v_mfma_f32_32x32x16_f16 (weights as the A operand: a lane ends up with
//     4 consecutive channels of one pixel);
//   * BOTH operands stream through one LDS ring of chunks of KC = 32 (or 64: small tiles) K-values = 2 (4) MFMA K-steps, 3 or 4 slots
//     (kVariants below), every chunk beyond the one being computed in flight, issued as `buffer_load_dwordx4 ... lds`
//     (tools/micro/buffer_lds_probe.hip: out-of-range lanes deliver zeros, LDS destinations above 64 KiB work):
//       - pixels: the implicit-GEMM gather -- chunk (tap, KC channels) of pixel p is KC * 2 contiguous bytes of the NHWC view.  A lane's byte offset
//         is computed ONCE; per tap one compare + select makes it ~0 where the tap leaves the image (hardware zero fill); per chunk a scalar moves.
//         LDS unit (16 B) pixel * UPP + (group ^ swizzle(pixel)) -- the swizzle is applied on the global side (which group a lane
//         fetches), so the lanes of a fragment read fall on different 16-byte slots (tools/lds_conflicts.py);
//       - weights: in the ring image order ([channel tile][chunk][K-step][k-half][BN rows][8]), re-ordered for ALL routed layers of a
//         forward by one launch (gemm_pack_jobs_kernel, planned by the engine) or by the launch itself (stand-alone entry points), so a
//         chunk is one contiguous block and every DMA piece reads 1 KiB of consecutive bytes;
//     every wave issues the same number of DMA instructions per chunk (counted vmcnt, no dummy transfers);
//   * K loop: the workgroup barrier that publishes chunk c + 1 stands at the TOP of iteration c and is followed at once by an MFMA block
//     whose fragments were requested a block earlier (MFMA issue blocks the wave and the pipe holds no queue: cycles between a barrier
//     and the next MFMA are lost outright); two fragment sets alternate, the set an MFMA block consumed is refilled with the K-step two
//     ahead.  Inline-asm ds_read / s_waitcnt: the compiler's own waitcnt insertion would put lgkmcnt(0) in front of every MFMA block;
//   * epilogues: fp16 outputs and the training epilogue's raw fp32 rows are transposed through the idle ring (contiguous stores); DPP statistics;
//   * blockIdx -> (pixel tile, channel tile) keeps the channel tiles of one pixel tile on one XCD (shared gathered pixels in its L2).
// Roofline: MFMA.  Measured bounds, ablations, cost model: DESIGN.md 5b, profiles/r03_gemm_*.
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "conv_tile_common.h"

namespace {
using namespace cvx_tile;

typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int GW = 8;        // waves

template <int MT, int NT, int NS, int KC>
struct GemmGeom {
  static_assert(NT % 2 == 0, "channel pieces divide evenly over the 8 waves");
  static_assert(NS >= 3 && NS <= 8, "ring slots");
  static_assert(KC == 32 || KC == 64, "K-values per ring chunk");
  static constexpr int GSLOTS = NS;
  static constexpr int KSTEPS = KC / 16;  // MFMA K-steps per chunk
  static constexpr int UPP = KC / 8;      // 16-byte units per pixel (and per weight row) and chunk
  static constexpr int BM = 4 * MT * 32, BN = 2 * NT * 32;
  static constexpr int A_BYTES = BM * KC * 2, B_BYTES = BN * KC * 2;  // one chunk of each operand
  static constexpr int SLOT_BYTES = A_BYTES + B_BYTES;
  static constexpr int A_PW = MT * KC / 32, B_PW = NT / 2 * KC / 32;  // 1-KiB DMA pieces per wave and chunk
  static constexpr int PW = A_PW + B_PW;
  static constexpr int RING_BYTES = GSLOTS * SLOT_BYTES + 4 * BN * 2 * 4 + CVX_MAX_TAPS * 8;  // ring | statistics scratch | tap offsets
  static constexpr int STAGE_BYTES = GW * (MT * 32 * (NT * 64 + 16) + MT * 32 * 8);         // epilogue staging (GemmStage), reuses the ring
  static constexpr int LDS_BYTES = RING_BYTES > STAGE_BYTES ? RING_BYTES : STAGE_BYTES;
};

template <int OFF>
__device__ __forceinline__ h8 lds_frag(unsigned a) {
  h8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// [rows][taps*Cin] fp16 (row pitch src_ld, tap block wtap[t] at wtap[t]*Cin) -> [n-block][chunk][K-step][k-half][BN rows][8]; one thread per
// 16-byte unit of the image
constexpr int PACK_UNITS_PER_BLOCK = 256 * 8;
__device__ __forceinline__ void gemm_pack_units(const GemmPackJob& a, long long u0, long long u1) {
  const long long per_block = (long long)a.chunks * (a.kc / 8) * a.BN;  // units per n-block
  const int upp = a.kc / 8;                                    // 16-byte units per row and chunk
  const int cpt = (a.Cin + a.kc - 1) / a.kc;                   // chunks per tap; the last one is zero-padded when Cin % kc != 0
  for (long long u = u0 + threadIdx.x; u < u1; u += blockDim.x) {
    const int nb = (int)(u / per_block);
    const long long r0 = u - (long long)nb * per_block;
    const int chunk = (int)(r0 / (upp * a.BN));
    const int r1 = (int)(r0 - (long long)chunk * upp * a.BN);
    const int kh = r1 / a.BN, row = r1 - kh * a.BN;  // kh = K-step * 2 + half
    const int n = nb * a.BN + row;
    const int tap = chunk / cpt, c0 = (chunk - tap * cpt) * a.kc + kh * 8;
    h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (n < a.rows && c0 < a.Cin) v = *reinterpret_cast<const h8*>(a.src + (long long)n * a.src_ld + a.taps[tap].wtap * a.Cin + c0);
    *reinterpret_cast<h8*>(a.dst + u * 8) = v;
  }
}
__global__ __launch_bounds__(256) void gemm_pack_kernel(const GemmPackJob a) {
  const long long total = (long long)a.chunks * (a.kc / 8) * a.BN * a.nblocks;
  const long long u0 = (long long)blockIdx.x * PACK_UNITS_PER_BLOCK;
  gemm_pack_units(a, u0, u0 + PACK_UNITS_PER_BLOCK < total ? u0 + PACK_UNITS_PER_BLOCK : total);
}
// all layers of a forward in one launch: block b belongs to the job whose [blk0, blk0 + nblk) holds it (binary search, jobs sorted by blk0)
__global__ __launch_bounds__(256) void gemm_pack_jobs_kernel(const GemmPackJob* jobs, int njobs) {
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].blk0 <= (int)blockIdx.x)
      lo = mid;
    else
      hi = mid - 1;
  }
  const GemmPackJob a = jobs[lo];
  const long long total = (long long)a.chunks * (a.kc / 8) * a.BN * a.nblocks;
  const long long u0 = (long long)((int)blockIdx.x - a.blk0) * PACK_UNITS_PER_BLOCK;
  if (u0 < total) gemm_pack_units(a, u0, u0 + PACK_UNITS_PER_BLOCK < total ? u0 + PACK_UNITS_PER_BLOCK : total);
}

// Epilogue of the fp32 head outputs (EPI = BIAS_F32): lane holds pixel lr of sub-tile i, channels nbase + j*32 + 8 g + 4 lh + (0..3) in
// acc[i][j][4 g ..] -- 16-byte stores straight from the accumulators.
template <int MT, int NT>
__device__ __forceinline__ void gemm_store_f32(const ConvParams& p, const f16v (&acc)[MT][NT], const long long (&out_off)[MT], const bool (&pvalid)[MT],
                                               int nbase, int lh) {
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n0 = nbase + j * 32 + g * 8 + lh * 4;
      if (n0 < p.Cout) {
        const f4 sh = *reinterpret_cast<const f4*>(p.bias + n0);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if (pvalid[i]) {
            f4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][4 * g + r] + sh[r];
            *reinterpret_cast<f4*>(p.out32 + out_off[i] + n0) = v;
          }
        }
      }
    }
}

// Epilogue of the fp16 outputs (EPI = AFFINE_SILU or PLAIN).  Straight from the accumulators a lane would store 8 bytes at a pixel pitch of
// Cout * 2 bytes -- 64 different cache lines per instruction, 20 us per 256 x 256 tile (measured, 14 % of the kernel).  Instead every
// wave transposes its MT*32 x NT*32 sub-tile through its own slice of the (now idle) ring: 8-byte LDS writes in the MFMA layout (row
// pitch NT*64 + 16 bytes: conflict-free), 16-byte reads along the channel axis, and 16-byte global stores in which the lanes of a pixel
// cover NT*64 contiguous bytes.  The slice is private to the wave: no barrier between the two halves.
template <int MT, int NT>
struct GemmStage {
  static constexpr int RS = NT * 64 + 16;                 // bytes per staged pixel row
  static constexpr int OFFS = MT * 32 * RS;               // the pixels' output offsets (8 bytes each) follow the rows
  static constexpr int WAVE_BYTES = OFFS + MT * 32 * 8;
};
// Sum over the 32 lanes of each half of the wave, valid in lanes 16..31 / 48..63: five DPP adds (quad swaps, half-row and row mirrors, the
// row broadcast) instead of five ds_bpermute round trips -- 640 of those per wave made the statistics the longest part of the epilogue.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float half_wave_sum(float v) {
  v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]: every lane holds its quad's sum
  v = dpp_add<0x141, 0xf>(v);  // row_half_mirror: the other quad of the half row
  v = dpp_add<0x140, 0xf>(v);  // row_mirror: the other half of the 16-lane row
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3: lanes 16..31 / 48..63 now hold the sum of their 32 lanes
  return v;
}

// The training epilogue's raw fp32 rows, staged like the fp16 outputs (a lane would otherwise store 16 bytes at a pixel pitch of Cout * 4
// bytes): two passes of NT / 2 channel tiles each, so that a pass of the wave's sub-tile has the footprint of the fp16 staging.
template <int MT, int NT>
__device__ __forceinline__ void gemm_store_raw(const ConvParams& p, const f16v (&acc)[MT][NT], const long long (&out_off)[MT], const bool (&pvalid)[MT],
                                               int nbase, int lane, unsigned char* wreg) {
  using S = GemmStage<MT, NT>;
  constexpr int JH = NT / 2;
  static_assert(JH * 128 + 16 == S::RS, "an fp32 pass of NT / 2 tiles has the row pitch of the fp16 staging");
  const int lr = lane & 31, lh = lane >> 5;
  if (lh == 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i) *reinterpret_cast<long long*>(wreg + S::OFFS + (i * 32 + lr) * 8) = pvalid[i] ? out_off[i] : -1;
  }
  constexpr int COLS = JH * 8, PPI = 64 / COLS;  // 16-byte columns of a staged row; pixels per store instruction
  const int c16 = lane % COLS, pr = lane / COLS;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int jj = 0; jj < JH; ++jj)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int j = pass * JH + jj;
          *reinterpret_cast<f4*>(wreg + (i * 32 + lr) * S::RS + (jj * 32 + g * 8 + lh * 4) * 4) =
              f4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        }
    const int n = nbase + pass * JH * 32 + c16 * 4;
#pragma unroll
    for (int it = 0; it < MT * 32 / PPI; ++it) {
      const int r = it * PPI + pr;
      const long long off = *reinterpret_cast<const long long*>(wreg + S::OFFS + r * 8);
      const f4 v = *reinterpret_cast<const f4*>(wreg + r * S::RS + c16 * 16);
      if (off >= 0 && n < p.Cout) cvx_store_raw4(p, off + n, v);
    }
  }
}

template <int MT, int NT, int EPI>
__device__ __forceinline__ void gemm_store_f16(const ConvParams& p, const f16v (&acc)[MT][NT], const long long (&out_off)[MT],
                                               const long long (&res_off)[MT], const bool (&pvalid)[MT], int nbase, int lane, unsigned char* wreg) {
  using S = GemmStage<MT, NT>;
  const int lr = lane & 31, lh = lane >> 5;
  const int act_kind = p.act_kind, res_pre = p.res_pre, accumulate = p.accumulate;
  const half_t* res = p.res;
  // pixel-shuffle store (ConvParams::ps_cin, plain epilogue only): channel n of the GEMM is channel n % ps_cin of the phase n / ps_cin, one
  // pixel down / right of the row's base pixel per phase bit; 8 consecutive channels never straddle a phase (ps_cin % 8 == 0)
  const int ps = EPI == CVX_EPI_PLAIN ? p.ps_cin : 0;
  const long long ps_row = (long long)p.OWr * p.out_ld;
  auto chan_off = [&](int n) -> long long {
    if (EPI != CVX_EPI_PLAIN || ps == 0) return n;
    const int ph = n / ps;
    return (ph >> 1) * ps_row + (long long)(ph & 1) * p.out_ld + (n - ph * ps);
  };
  if (lh == 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i) *reinterpret_cast<long long*>(wreg + S::OFFS + (i * 32 + lr) * 8) = pvalid[i] ? out_off[i] : -1;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cl = j * 32 + g * 8 + lh * 4;  // channel inside the wave's sub-tile
      const int n0 = nbase + cl;
      f4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      const bool inside = n0 < p.Cout;
      if constexpr (EPI == CVX_EPI_AFFINE_SILU) {
        if (inside) {
          sc = *reinterpret_cast<const f4*>(p.scale + n0);
          sh = *reinterpret_cast<const f4*>(p.shift + n0);
        }
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        f4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][4 * g + r] * sc[r] + sh[r];
        if constexpr (EPI == CVX_EPI_AFFINE_SILU) {
          f4 rv = {0.f, 0.f, 0.f, 0.f};
          if (res && inside && pvalid[i]) {
            const h4 rr = *reinterpret_cast<const h4*>(res + res_off[i] + n0);
#pragma unroll
            for (int r = 0; r < 4; ++r) rv[r] = (float)rr[r];
          }
          if (res_pre) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += rv[r];
          }
          if (act_kind == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = cvx_silu(v[r]);
          } else if (act_kind == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          if (!res_pre) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += rv[r];
          }
        }
        if (accumulate && inside && pvalid[i]) {  // data gradients that add to what another consumer left there
          const h4 old = *reinterpret_cast<const h4*>(p.out16 + out_off[i] + chan_off(n0));
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)old[r];
        }
        *reinterpret_cast<h4*>(wreg + (i * 32 + lr) * S::RS + cl * 2) = h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      }
    }
  // ---- rows back out: COLS lanes (16 bytes each) per pixel ----
  constexpr int COLS = NT * 4, PPI = 64 / COLS;  // pixels per store instruction
  const int c16 = lane % COLS, pr = lane / COLS;
  const int n = nbase + c16 * 8;
#pragma unroll
  for (int it = 0; it < MT * 32 / PPI; ++it) {
    const int r = it * PPI + pr;
    const long long off = *reinterpret_cast<const long long*>(wreg + S::OFFS + r * 8);
    const h8 v = *reinterpret_cast<const h8*>(wreg + r * S::RS + c16 * 16);
    if (off >= 0) {
      half_t* dst = p.out16 + off + chan_off(n);
      if (n + 8 <= p.Cout)
        *reinterpret_cast<h8*>(dst) = v;
      else if (n + 4 <= p.Cout)
        *reinterpret_cast<h4*>(dst) = h4{v[0], v[1], v[2], v[3]};
    }
  }
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// waits until at most k chunks (of PW pieces each) are outstanding; k is wave-uniform, 0..5
template <int PW>
__device__ __forceinline__ void wait_chunks(int k) {
  switch (k) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<PW>(); break;
    case 2: wait_vmcnt<2 * PW>(); break;
    case 3: wait_vmcnt<3 * PW>(); break;
    case 4: wait_vmcnt<4 * PW>(); break;
    default: wait_vmcnt<5 * PW>(); break;
  }
}
// timing experiments of the tuning build (CVX_GEMM_DBG bits 32: no DMA inside the K loop, 64: no MFMA, 128: no barrier; results are WRONG)
#ifdef CVX_TUNING
#define CVX_GEMM_DBG_BIT(b) ((p.dbg & (b)) != 0)
#else
#define CVX_GEMM_DBG_BIT(b) false
#endif

// issue cursor of the ring: which (tap, channel block) the next chunk is, and the per-lane gather offsets under that tap
template <int AP>
struct GemmCursor {
  int chunk, tap, cc;      // next chunk to issue; its tap and its 32-channel block inside the tap
  int slot;                // ring slot the next chunk goes to
  unsigned vo[AP];         // per lane: byte offset of (pixel + tap, channel group), or ~0 where the tap leaves the image
  unsigned vo_tail[AP];    // the same for the tap's last chunk when it is ragged: ~0 for the channel groups past Cin
};

// the tap table is staged in LDS at kernel start (sTap: dh, dw per tap): a global load inside the K loop would sit in the middle of the
// counted vmcnt queue of the ring
template <int AP>
__device__ __forceinline__ void gemm_enter_tap(GemmCursor<AP>& k, const ConvParams& p, const int* sTap, const int (&a_ih)[AP], const int (&a_iw)[AP],
                                               const unsigned (&a_off)[AP], const bool (&a_tail_ok)[AP]) {
  const int dh = __builtin_amdgcn_readfirstlane(sTap[2 * k.tap]), dw = __builtin_amdgcn_readfirstlane(sTap[2 * k.tap + 1]);
  const int toff = ((dh * p.IW + dw) * p.in_ld) * 2;  // bytes; negative for the taps above / left of the pixel
#pragma unroll
  for (int q = 0; q < AP; ++q) {
    const bool ok = (unsigned)(a_ih[q] + dh) < (unsigned)p.IH && (unsigned)(a_iw[q] + dw) < (unsigned)p.IW;
    k.vo[q] = ok ? a_off[q] + (unsigned)toff : 0xffffffffu;  // inside the image the sum is a valid offset into the view
    k.vo_tail[q] = a_tail_ok[q] ? k.vo[q] : 0xffffffffu;
  }
}

template <int MT, int NT, int NS, int KC>
__global__ __launch_bounds__(64 * GW) void conv_gemm_kernel(const ConvParams p, const half_t* __restrict__ wpk, int m_tiles, int n_tiles, int nchunks,
                                                            unsigned a_records) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (and has no __amdgpu_buffer_rsrc_t)
  using G = GemmGeom<MT, NT, NS, KC>;
  constexpr int BM = G::BM, BN = G::BN, GSLOTS = NS, UPP = G::UPP, KS = G::KSTEPS, AP = G::A_PW;
  static_assert(5 * G::PW <= 63 && GSLOTS - 3 <= 5, "vmcnt field; wait_chunks cases");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sStat = reinterpret_cast<float*>(smem + GSLOTS * G::SLOT_BYTES);
  int* sTap = reinterpret_cast<int*>(smem + GSLOTS * G::SLOT_BYTES + 4 * BN * 2 * 4);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave & 3, wn = wave >> 2;
  const int lr = lane & 31, lh = lane >> 5;
  // tile of this workgroup: the channel tiles of one pixel tile run on one XCD (blocks b and b + 8 share an XCD)
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int n_tile = seq % n_tiles;
  const int m_tile = (seq / n_tiles) * 8 + xcd;
  if (m_tile >= m_tiles) return;
  clk_mark(p, 0);
  if (tid < p.ntaps) {
    const ConvTap td = p.taps[tid];
    sTap[2 * tid] = td.dh;
    sTap[2 * tid + 1] = td.dw;
  }
  __syncthreads();
  const long long M = (long long)p.B * p.OH2 * p.OW2;
  const long long m_base = (long long)m_tile * BM;
  const int cpt = (p.Cin + KC - 1) / KC;  // chunks per tap
  const int tail_groups = (p.Cin % KC) / 8;  // 8-channel groups of a tap's last chunk when Cin % KC != 0 (0: every chunk is full)

  // ---- per-lane gather assignment: pixel piece q of this wave covers LDS units (q * 8 + wave) * 64 + lane = pixel * UPP + slot ----
  int a_ih[AP], a_iw[AP];
  unsigned a_off[AP];
  bool a_tail_ok[AP];
#pragma unroll
  for (int q = 0; q < AP; ++q) {
    const int pix = (q * GW + wave) * (64 / UPP) + lane / UPP;
    // the channel group this lane fetches: the read-side swizzle (gemm_swizzle), applied at the source
    const int grp = (lane % UPP) ^ (UPP == 4 ? (pix >> 2) & 3 : (pix >> 1) & 7);
    a_tail_ok[q] = tail_groups == 0 || grp < tail_groups;
    const long long m = m_base + pix;
    a_ih[q] = a_iw[q] = -(1 << 24);  // rows past the last pixel: no tap is ever inside the image
    a_off[q] = 0;
    if (m < M) {
      const unsigned mu = (unsigned)m;
      const unsigned tq = mu / (unsigned)p.OW2;
      const int ow2 = (int)(mu - tq * (unsigned)p.OW2);
      const int b = (int)(tq / (unsigned)p.OH2);
      const int oh2 = (int)(tq - (unsigned)b * (unsigned)p.OH2);
      a_ih[q] = oh2 * p.IS;
      a_iw[q] = ow2 * p.IS;
      a_off[q] = (unsigned)(((long long)b * p.in_bstride + ((long long)a_ih[q] * p.IW + a_iw[q]) * p.in_ld + grp * 8) * 2);
    }
  }
  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.in), (short)0, (int)a_records, 0x00020000);
  const half_t* wblk = wpk + (long long)n_tile * nchunks * (G::B_BYTES / 2);
  const __amdgpu_buffer_rsrc_t rsrc_b =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(wblk), (short)0, (int)((long long)nchunks * G::B_BYTES), 0x00020000);
  const unsigned vb = (unsigned)(lane * 16);

  GemmCursor<AP> cur;
  cur.chunk = 0;
  cur.tap = 0;
  cur.cc = 0;
  cur.slot = 0;
  gemm_enter_tap<AP>(cur, p, sTap, a_ih, a_iw, a_off, a_tail_ok);
  auto issue = [&]() __attribute__((always_inline)) {
    unsigned char* sb = smem + cur.slot * G::SLOT_BYTES;
    cur.slot = cur.slot + 1 == GSLOTS ? 0 : cur.slot + 1;
    const unsigned sa = (unsigned)(cur.cc * (KC * 2));
    const bool tail = cur.cc == cpt - 1;  // (uniform) the ragged chunk, if any: vo_tail == vo when Cin % 32 == 0
#pragma unroll
    for (int q = 0; q < G::A_PW; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sb + (q * GW + wave) * 1024), 16, tail ? cur.vo_tail[q] : cur.vo[q], sa, 0, 0);
    const unsigned sw = (unsigned)cur.chunk * (unsigned)G::B_BYTES;
#pragma unroll
    for (int q = 0; q < G::B_PW; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(sb + G::A_BYTES + (q * GW + wave) * 1024), 16, vb, sw + (q * GW + wave) * 1024, 0, 0);
    ++cur.chunk;
    if (++cur.cc == cpt && cur.chunk < nchunks) {
      cur.cc = 0;
      ++cur.tap;
      gemm_enter_tap<AP>(cur, p, sTap, a_ih, a_iw, a_off, a_tail_ok);
    }
  };

  f16v acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  clk_mark(p, 1);
  if (p.clk) wait_vmcnt<0>();  // the stamp is a store: keep it out of the counted queue
#pragma unroll
  for (int s = 0; s < GSLOTS - 1; ++s)
    if (s < nchunks) issue();

  // fragment addresses inside a slot (bytes).  Pixel operand: unit pixel * UPP + (group ^ swizzle); K-step ks reads groups 2 ks + lh, i.e.
  // the address of K-step 0 XOR ks * 32; sub-tile i is 32 pixels further (the swizzle term does not change).  Swizzle: (pixel >> 2) & 3
  // with 4 units per pixel, (pixel >> 1) & 7 with 8: the 16 lanes of every ds_read_b128 lane group fall on 16 different 16-byte slots
  // (tools/lds_conflicts.py).  Weight operand: [K-step][k-half][BN][8].
  const int a_sw = UPP == 4 ? (lr >> 2) & 3 : (lr >> 1) & 7;
  const unsigned a_rd0 = lds_addr(smem) + (unsigned)((wm * MT * 32 + lr) * (UPP * 16) + ((lh ^ a_sw) << 4));
  const unsigned b_rd = lds_addr(smem) + (unsigned)(G::A_BYTES + (lh * BN + wn * NT * 32 + lr) * 16);
  h8 xa[2][MT], wb[2][NT];
#define CVX_GEMM_READ(SET, SO, KSTEP)                                                         \
  {                                                                                           \
    const unsigned va_ = (a_rd0 ^ (unsigned)((KSTEP) * 32)) + (SO), vb_ = b_rd + (SO);        \
    if constexpr (MT >= 1) xa[SET][0] = lds_frag<0>(va_);                                     \
    if constexpr (MT >= 2) xa[SET][1] = lds_frag<32 * UPP * 16>(va_);                         \
    if constexpr (NT >= 1) wb[SET][0] = lds_frag<(KSTEP) * 2 * BN * 16 + 0 * 512>(vb_);       \
    if constexpr (NT >= 2) wb[SET][1] = lds_frag<(KSTEP) * 2 * BN * 16 + 1 * 512>(vb_);       \
    if constexpr (NT >= 3) wb[SET][2] = lds_frag<(KSTEP) * 2 * BN * 16 + 2 * 512>(vb_);       \
    if constexpr (NT >= 4) wb[SET][3] = lds_frag<(KSTEP) * 2 * BN * 16 + 3 * 512>(vb_);       \
  }
#define CVX_GEMM_MFMA(SET)                                                                    \
  {                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    if (!CVX_GEMM_DBG_BIT(64))                                                                \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) _Pragma("unroll") for (int i = 0; i < MT; ++i) acc[i][j] =                       \
        __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[SET][j], xa[SET][i], acc[i][j], 0, 0, 0);   \
    __builtin_amdgcn_sched_barrier(0);                                                        \
  }
  static_assert(MT <= 2 && NT <= 4, "fragment macros");
  static_assert(KS == 2 || KS == 4, "the K-step schedule below");
  // Loop invariant at the top of iteration c: the fragments of K-steps 0 and 1 of chunk c are in registers (or on their way), chunk c + 1
  // has landed for this wave.  The barrier then publishes chunk c + 1 and retires chunk c - 1, and the first thing behind it is an MFMA
  // block whose operands are already there: MFMA issue blocks the wave and the pipe holds no queue, so cycles in which all eight waves
  // sit between a barrier and their next MFMA are lost outright (measured: 1.43x the MFMA time with the barrier in mid-chunk).  Two
  // fragment sets alternate: the set an MFMA block has just consumed is refilled with the K-step two ahead (the next chunk's at the end).
  // chunk 0 has landed when at most the chunks issued behind it (up to NS - 2 of them) are outstanding
  const int issued0 = nchunks < GSLOTS - 1 ? nchunks : GSLOTS - 1;
  wait_chunks<G::PW>(issued0 - 1);
  workgroup_barrier();  // chunk 0 published
  if (p.clk) {  // tuning runs only: the stamp's store would sit in the counted vmcnt queue, so the ring is drained once here
    clk_mark(p, 2);
    wait_vmcnt<0>();
    workgroup_barrier();
  }
  CVX_GEMM_READ(0, 0u, 0);
  CVX_GEMM_READ(1, 0u, 1);
  wait_chunks<G::PW>(issued0 >= 2 ? issued0 - 2 : 0);  // chunk 1 landed
  unsigned so = 0;  // slot of chunk c
  for (int c = 0; c + 1 < nchunks; ++c) {
    const unsigned sn = so + G::SLOT_BYTES == GSLOTS * G::SLOT_BYTES ? 0u : so + G::SLOT_BYTES;  // slot of chunk c + 1
    if (!CVX_GEMM_DBG_BIT(128)) workgroup_barrier();  // chunk c + 1 published; every wave is done reading chunk c - 1: its slot takes chunk c + NS - 1
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT + NT) : "memory");  // this K-step's fragments (LDS returns in order; the next set may be pending)
      if (ks & 1) {
        CVX_GEMM_MFMA(1);
      } else {
        CVX_GEMM_MFMA(0);
      }
      if (ks == 0 && cur.chunk < nchunks && !CVX_GEMM_DBG_BIT(32)) issue();
      // refill the set just consumed: K-step ks + 2 of this chunk, or K-step ks + 2 - KS of the next one
      if (ks == 0) {
        if constexpr (KS > 2) { CVX_GEMM_READ(0, so, 2); } else { CVX_GEMM_READ(0, sn, 0); }
      } else if (ks == 1) {
        if constexpr (KS > 2) { CVX_GEMM_READ(1, so, 3); } else { CVX_GEMM_READ(1, sn, 1); }
      } else if (ks == 2) {
        CVX_GEMM_READ(0, sn, 0);
      } else {
        CVX_GEMM_READ(1, sn, 1);
      }
    }
    // chunk c + 2 landed; the chunks issued behind it (up to NS - 3, fewer at the end of K) may stay in flight
    const int behind = nchunks - 3 - c;
    wait_chunks<G::PW>(behind < 0 ? 0 : behind < GSLOTS - 3 ? behind : GSLOTS - 3);
    so = sn;
  }
  // the last chunk: nothing behind it
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + 1 < KS) {
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT + NT) : "memory");
    } else {
      wait_lgkm();
    }
    if (ks & 1) {
      CVX_GEMM_MFMA(1);
    } else {
      CVX_GEMM_MFMA(0);
    }
    if (ks == 0 && KS > 2) CVX_GEMM_READ(0, so, 2);
    if (ks == 1 && KS > 2) CVX_GEMM_READ(1, so, 3);
  }
#undef CVX_GEMM_READ
#undef CVX_GEMM_MFMA
  clk_mark(p, 3);
  if (p.dbg & 16) return;

  // ---- epilogue: lane holds pixel lr of sub-tile i, channels n_tile*BN + (wn*NT + j)*32 + 8 g + 4 lh + (0..3) in acc[i][j][4 g ..] ----
  long long out_off[MT], res_off[MT];
  bool pvalid[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const long long m = m_base + (wm * MT + i) * 32 + lr;
    pvalid[i] = m < M;
    const unsigned mu = pvalid[i] ? (unsigned)m : 0u;
    const unsigned tq = mu / (unsigned)p.OW2;
    const int ow2 = (int)(mu - tq * (unsigned)p.OW2);
    const int b = (int)(tq / (unsigned)p.OH2);
    const int oh2 = (int)(tq - (unsigned)b * (unsigned)p.OH2);
    const long long pix = (long long)(oh2 * p.OS + p.oph) * p.OWr + (ow2 * p.OS + p.opw);
    out_off[i] = (long long)b * p.out_bstride + pix * p.out_ld;
    res_off[i] = (long long)b * p.res_bstride + pix * p.res_ld;
  }
  const int nbase = n_tile * BN + wn * NT * 32;
  if (p.epi == CVX_EPI_RAW_STATS) {
    // raw fp32 output + per-channel (sum, sum of squares): lanes of one k-half hold 32 pixels of the same 4 channels
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int chl = (wn * NT + j) * 32 + g * 8 + lh * 4;  // channel inside the block's BN range
        const int n0 = n_tile * BN + chl;
        float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if (!pvalid[i]) continue;
          f4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s1[r] += v[r];
            s2[r] += v[r] * v[r];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = half_wave_sum(s1[r]), b2 = half_wave_sum(s2[r]);
          if (lr == 31) {
            sStat[(wm * BN + chl + r) * 2 + 0] = a;
            sStat[(wm * BN + chl + r) * 2 + 1] = b2;
          }
        }
      }
    __syncthreads();
    for (int t = tid; t < BN * 2; t += 64 * GW) {
      const int ch = t >> 1, which = t & 1;
      const int n = n_tile * BN + ch;
      if (n < p.Cout) {
        const float v = (sStat[(0 * BN + ch) * 2 + which] + sStat[(1 * BN + ch) * 2 + which]) + (sStat[(2 * BN + ch) * 2 + which] + sStat[(3 * BN + ch) * 2 + which]);
        cvx_fix_atomic_add(p.stats, ((long long)(m_tile % p.stats_replicas) * p.Cout + n) * 2 + which, v);
      }
    }
    __syncthreads();  // the statistics scratch has been read, every wave is done with the ring: its space stages the raw output
    gemm_store_raw<MT, NT>(p, acc, out_off, pvalid, nbase, lane, smem + wave * GemmStage<MT, NT>::WAVE_BYTES);
    clk_mark(p, 4);
    return;
  }
  // one straight-line body per epilogue kind: with the kinds tested inside the (j, g, i) loops hipcc 7.2 structurised the control flow
  // into a path that left the channel offset of the PLAIN store undefined (a stale pointer was used: the faults of the first bring-up)
  if (p.epi == CVX_EPI_BIAS_F32) {
    gemm_store_f32<MT, NT>(p, acc, out_off, pvalid, nbase, lh);
  } else {
    __syncthreads();  // every wave is done with the ring: its space stages the output tile
    unsigned char* wreg = smem + wave * GemmStage<MT, NT>::WAVE_BYTES;
    if (p.epi == CVX_EPI_AFFINE_SILU)
      gemm_store_f16<MT, NT, CVX_EPI_AFFINE_SILU>(p, acc, out_off, res_off, pvalid, nbase, lane, wreg);
    else
      gemm_store_f16<MT, NT, CVX_EPI_PLAIN>(p, acc, out_off, res_off, pvalid, nbase, lane, wreg);
  }
  clk_mark(p, 4);
#endif
}

// packed weights of a (weight pointer, tap table) pair: re-packed at EVERY launch (the fp16 shadows change with every optimiser step),
// the buffer itself is kept
struct PackSlot {
  half_t* buf = nullptr;
  size_t bytes = 0;
};
std::map<std::pair<const void*, std::pair<const void*, int>>, PackSlot> g_pack;  // stand-alone launches only (the engine packs at plan time)
std::mutex g_pack_mu;

template <int MT, int NT, int NS, int KC>
int launch_gemm(const ConvParams& p, hipStream_t stream) {
  using G = GemmGeom<MT, NT, NS, KC>;
  const long long M = (long long)p.B * p.OH2 * p.OW2;
  const int m_tiles = (int)((M + G::BM - 1) / G::BM), n_tiles = (p.Cout + G::BN - 1) / G::BN;
  const int nchunks = p.ntaps * ((p.Cin + KC - 1) / KC);
  // ---- weights -> ring image order: done for all layers at once by the engine (wt_packed), or here for a stand-alone launch ----
  static const int dbg = cvx_tune_int("CVX_GEMM_DBG", 0);  // tuning build: 1 no pack, 2 no main kernel, 16 main kernel stops after the K loop, 32.. see CVX_GEMM_DBG_BIT
  const half_t* packed = (p.wt_packed_bn == G::BN && p.wt_packed_kc == KC) ? p.wt_packed : nullptr;
  if (!packed) {
    std::lock_guard<std::mutex> lk(g_pack_mu);
    PackSlot& ps = g_pack[{p.wt, {p.taps, G::BN * 1024 + KC}}];
    const size_t need = (size_t)n_tiles * nchunks * G::B_BYTES;
    if (ps.bytes < need) {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing(stream, &cs);
      CVX_CHECK(cs == hipStreamCaptureStatusNone, "conv_gemm: the packed-weight buffer cannot be (re)allocated while the stream is capturing -- run the launch once before the capture");
      if (ps.buf) CVX_HIP(hipFree(ps.buf));
      CVX_HIP(hipMalloc((void**)&ps.buf, need));
      ps.bytes = need;
    }
    GemmPackJob a;
    memset(&a, 0, sizeof(a));
    a.src = p.wt;
    a.dst = ps.buf;
    a.src_ld = p.wt_ld;
    a.rows = p.Cout;
    a.Cin = p.Cin;
    a.ntaps = p.ntaps;
    a.BN = G::BN;
    a.kc = KC;
    a.nblocks = n_tiles;
    a.chunks = nchunks;
    a.taps = p.taps;
    const long long units = (long long)n_tiles * nchunks * G::UPP * G::BN;
    if (!(dbg & 1)) hipLaunchKernelGGL(gemm_pack_kernel, dim3((unsigned)((units + PACK_UNITS_PER_BLOCK - 1) / PACK_UNITS_PER_BLOCK)), dim3(256), 0, stream, a);
    packed = ps.buf;
  }
  if (dbg & 2) return 0;
  ConvParams pd = p;
  pd.dbg = dbg;
  pd.clk = g_cvx_clk;
  static unsigned long long optin_mask = 0;
  CVX_TRY(cvx_lds_optin((const void*)conv_gemm_kernel<MT, NT, NS, KC>, G::LDS_BYTES, &optin_mask));
  const int grid = ((m_tiles + 7) / 8) * 8 * n_tiles;
  const unsigned a_records = (unsigned)std::min<long long>((long long)p.B * p.in_bstride * 2, 0xffffffffLL);
  hipLaunchKernelGGL((conv_gemm_kernel<MT, NT, NS, KC>), dim3(grid), dim3(64 * GW), G::LDS_BYTES, stream, pd, packed, m_tiles, n_tiles, nchunks,
                     a_records);
  return 0;
}

// The variants of the kernel and what the cost model knows about each (calibrated on the phase stamps of tools/gemm_debug.py, MI355X, clock as
// held under this load): us per chunk with the CU saturated by this variant, prologue + epilogue, workgroups that share a CU.
struct GemmVariant {
  int bm, bn, slots, kc, per_cu;
  double chunk_us, fixed_us;
  long long min_wgs;  // only for grids at least this large
};
constexpr GemmVariant kVariants[] = {
    {256, 256, 4, 32, 1, 0.80, 12.0, 0},   // 1
    {256, 128, 4, 32, 1, 0.50, 7.0, 0},    // 2
    {128, 256, 4, 32, 1, 0.52, 7.6, 0},    // 3
    {128, 128, 4, 32, 2, 0.56, 6.0, 0},    // 4: two workgroups per CU
    {256, 128, 3, 32, 2, 1.03, 7.0, 1024}, // 5: two per CU (their barrier gaps fill each other); a workgroup alone on its CU is slower than variant 2
    {128, 128, 4, 64, 1, 0.58, 5.0, 0},    // 6: 64-deep chunks -- per-chunk costs (barrier, DMA issue) amortised over twice the MFMAs; one workgroup per CU
    {256, 128, 3, 64, 1, 0.98, 7.5, 0},    // 7: (measured 5-7 % ahead of variant 2 on every probe shape)
};
constexpr int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));

}  // namespace

bool cvx_conv_gemm_shape_ok(const ConvParams& p) {
  const long long M = (long long)p.B * p.OH2 * p.OW2;
  // buffer_load offsets are 32-bit: the gathered view and one channel tile's packed weights stay below 4 GiB / 2 GiB
  return p.nphase <= 1 && p.Cin % 8 == 0 && p.ntaps <= CVX_MAX_TAPS && p.Cout % 4 == 0 && M < (1LL << 31) &&
         (long long)p.B * p.in_bstride * 2 < (1LL << 32) && (long long)p.ntaps * (p.Cin + 64) * 256 * 2 < (1LL << 31);
}

bool cvx_conv_gemm_supported(const ConvParams& p) {
  static const bool off = cvx_tune_set("CVX_NO_GEMM");
  if (off || !cvx_conv_gemm_shape_ok(p)) return false;
  if (p.ps_cin > 0) {  // pixel-shuffle data gradient (its alternative is the merged-phase launch of the ring kernel, not the kernels below)
    static const int ps_cmin = cvx_tune_int("CVX_PS_CMIN", 64);
    return p.Cout >= ps_cmin && p.Cin % 32 == 0;
  }
  const long long M = (long long)p.B * p.OH2 * p.OW2;
  const long long K = (long long)p.ntaps * p.Cin;
  // measured against the pointwise / halo / LDS-DMA ring kernels (tools/gemm_probe.py + tools/gemm_trace.py, profiles/r03_gemm_*): ahead
  // on every shape from ~100 channels up (1.5 - 2x at 256+); the gates below come from whole-model A/B runs of all ten workloads
  // (profiles/r03_gemm_model_ab.txt: lower K / FLOP gates were equal or better everywhere); the tuning build moves them
  static const int kmin = cvx_tune_int("CVX_GEMM_KMIN", 64);
  static const int mmin = cvx_tune_int("CVX_GEMM_MMIN", 2048);
  static const int cmin = cvx_tune_int("CVX_GEMM_CMIN", 96);
  static const int gfmin = cvx_tune_int("CVX_GEMM_GFMIN", 0);  // GFLOP per launch
  // channel tiles are 128 wide: a layer that fills under 70 % of them (YOLOv8-n's Detect convs, 64 + 80 = 144 channels: 56 %) computes mostly
  // padding -- measured 114 vs 56 us (80x80 64->144) and 153 vs 114 us (40x40 128->144) against the halo / ring kernels
  const int padded = (p.Cout + 127) / 128 * 128;
  // ... except where K is so deep that the ring kernel's re-gathering (every input pixel once per tap, per 64-channel output block) costs more
  // than computing padding: SSD's loc / conf heads (3x3 on 512..1024 channels, 16..88 outputs)
  static const int bigk = cvx_tune_int("CVX_GEMM_BIGK", 2304);  // SSD inference 7895 -> 8134 img/s, the other workloads unchanged
  const bool deep = K >= bigk && p.Cout >= 8;
  return K >= kmin && M >= mmin && 2.0 * (double)M * (double)K * p.Cout >= gfmin * 1e9 && (deep || (p.Cout >= cmin && p.Cout * 10 >= padded * 7));
}

// Variant by cost model: time of one workgroup = chunks * chunk_us + fixed_us, times the rounds the grid needs on 256 CUs (x per_cu).  The model
// reproduces the measured launches within 10 % (GEMM-shaped kernel).  Returns the variant number (index + 1).
static int gemm_pick_variant(long long M, int Cin, int Cout, int ntaps, double* est_us) {
  int best = 4;
  double best_t = 1e30;
  for (int i = 0; i < kNumVariants; ++i) {
    const GemmVariant& v = kVariants[i];
    if (v.bn > 128 && Cout <= 128) continue;  // half of every channel tile would be padding
    const long long wgs = ((M + v.bm - 1) / v.bm) * ((Cout + v.bn - 1) / v.bn);
    if (wgs < v.min_wgs) continue;
    const int nchunks = ntaps * ((Cin + v.kc - 1) / v.kc);
    const long long rounds = (wgs + 256 * v.per_cu - 1) / (256 * v.per_cu);
    double t = (double)rounds * (nchunks * v.chunk_us + v.fixed_us);
    // a last round that fills under half of the slots of a two-per-CU variant runs its workgroups alone on their CUs: ~0.6 of the pair time
    if (v.per_cu == 2 && (wgs - (rounds - 1) * 512) <= 256) t -= 0.4 * (nchunks * v.chunk_us + v.fixed_us);
    if (t < best_t) {
      best_t = t;
      best = i + 1;
    }
  }
  if (est_us) *est_us = best_t;
  return best;
}

static int gemm_variant_for(const ConvParams& p) {
  static const int force = cvx_tune_int("CVX_GEMM_TILE", 0);  // tuning build: variant number 1..7 (kVariants)
  if (p.gemm_variant >= 1 && p.gemm_variant <= kNumVariants) return p.gemm_variant;
  if (force >= 1 && force <= kNumVariants) return force;
  return gemm_pick_variant((long long)p.B * p.OH2 * p.OW2, p.Cin, p.Cout, p.ntaps, nullptr);
}

bool cvx_conv_gemm_plan(const ConvParams& p, GemmPackJob* job, size_t* bytes) {
  if (p.nphase > 1 || !cvx_conv_gemm_supported(p)) return false;
  const GemmVariant& v = kVariants[gemm_variant_for(p) - 1];
  memset(job, 0, sizeof(*job));
  job->src = p.wt;
  job->taps = p.taps;
  job->src_ld = p.wt_ld;
  job->rows = p.Cout;
  job->Cin = p.Cin;
  job->ntaps = p.ntaps;
  job->BN = v.bn;
  job->kc = v.kc;
  job->nblocks = (p.Cout + v.bn - 1) / v.bn;
  job->chunks = p.ntaps * ((p.Cin + v.kc - 1) / v.kc);
  const long long units = (long long)job->nblocks * job->chunks * (v.kc / 8) * v.bn;
  job->nblk = (int)((units + PACK_UNITS_PER_BLOCK - 1) / PACK_UNITS_PER_BLOCK);
  *bytes = (size_t)units * 16;
  return true;
}

int cvx_conv_gemm_pack_jobs(const GemmPackJob* d_jobs, int njobs, int nblocks, hipStream_t stream) {
  if (njobs <= 0 || nblocks <= 0) return 0;
  hipLaunchKernelGGL(gemm_pack_jobs_kernel, dim3(nblocks), dim3(256), 0, stream, d_jobs, njobs);
  CVX_HIP(hipGetLastError());
  return 0;
}

int cvx_conv_gemm_launch(const ConvParams& p, hipStream_t stream) {
  switch (gemm_variant_for(p)) {
    case 1: CVX_TRY((launch_gemm<2, 4, 4, 32>(p, stream))); break;
    case 2: CVX_TRY((launch_gemm<2, 2, 4, 32>(p, stream))); break;
    case 3: CVX_TRY((launch_gemm<1, 4, 4, 32>(p, stream))); break;
    case 5: CVX_TRY((launch_gemm<2, 2, 3, 32>(p, stream))); break;
    case 6: CVX_TRY((launch_gemm<1, 2, 4, 64>(p, stream))); break;
    case 7: CVX_TRY((launch_gemm<2, 2, 3, 64>(p, stream))); break;
    default: CVX_TRY((launch_gemm<1, 2, 4, 32>(p, stream))); break;
  }
  CVX_HIP(hipGetLastError());
  return 0;
}

// frees the packed-weight buffers of the stand-alone launches (cvx_engine_destroy calls it: entries are keyed by weight pointers that die with their engine)
void cvx_conv_gemm_release() {
  std::lock_guard<std::mutex> lk(g_pack_mu);
  for (auto& kv : g_pack)
    if (kv.second.buf) (void)hipFree(kv.second.buf);
  g_pack.clear();
}
