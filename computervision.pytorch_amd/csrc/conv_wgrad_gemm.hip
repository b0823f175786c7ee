// GEMM-shaped weight gradient for the big-channel layers (gfx950): the sibling of conv_gemm.hip for
//
//   dW[co][tap][ci] = sum_m dy[m][co] * x[pixel(m, tap)][ci]            (m over B*OH*OW output pixels)
//
// i.e. D[co][j] with j = tap * cin_pad16 + ci, the REDUCTION running over pixels.  Both operands lie pixel-major in memory (NHWC), so the
// reduction index is the slow index of both: their tiles are staged as they lie ([32 pixels][channels], rows of 256 or 512 bytes) and the
// MFMA fragments -- 8 consecutive pixels of one channel per lane -- are fetched with the transposing LDS read ds_read_b64_tr_b16 (two per
// fragment: 4 pixels x 16 channels per 16-lane group each).
//
//   * 512 threads = 8 waves as 4 (output channels) x 2 (columns j); tile 256 x 256 or 128 x 128; every wave owns MT x NT tiles of
//     v_mfma_f32_32x32x16_f16 with dy as the A operand: a lane ends up with one column j and 16 rows co -- the slab rows are written as
//     128-byte runs;
//   * both operands stream through one 4-slot LDS ring of 32-pixel chunks by `buffer_load_dwordx4 ... lds` (conv_gemm.hip's recipe: per-lane
//     offsets, hardware zero fill).  dy of a BatchNorm layer is dense, so its rows advance by a SCALAR offset per chunk and the pixels past
//     the end of a workgroup's split fall outside the buffer descriptor (num_records = end of the split): zeros, which also silence
//     whatever x holds there.  The x pixel of a lane is stepped by 32 per chunk (row / image wraps by compares), one compare + select per
//     chunk turns its offset into ~0 where the tap leaves the image;
//   * 16-byte channel groups are XOR-swizzled by (pixel & 3) << 2 on the global side: the four pixel rows a transposed read touches fall
//     on disjoint banks (cdna_hip_programming.md T10);
//   * the K loop is conv_gemm.hip's: barrier at the top of the iteration, an MFMA block with ready operands right behind it, the next
//     fragments requested a block ahead, counted vmcnt;
//   * the pixel range is split over workgroups (nsplit); every split stores its own fp32 slab (plain stores, deterministic), summed into
//     the gradient arena by cvx_reduce_slabs.
#include <algorithm>
#include <cstring>

#include "conv_tile_common.h"

namespace {
using namespace cvx_tile;

typedef float f16v __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int GW = 8;      // waves
constexpr int NSLOT = 4;   // ring slots
constexpr int PKC = 32;    // pixels per chunk (two MFMA K-steps)

template <int MT, int NT>
struct WgGeom {
  static_assert(NT % 2 == 0, "column pieces divide evenly over the 8 waves");
  static constexpr int CO_B = 4 * MT * 32, J_B = 2 * NT * 32;
  static constexpr int D_ROW = CO_B * 2, X_ROW = J_B * 2;  // bytes of one pixel row of each tile
  static constexpr int D_BYTES = PKC * D_ROW, X_BYTES = PKC * X_ROW;
  static constexpr int SLOT_BYTES = D_BYTES + X_BYTES;
  static constexpr int D_PW = MT, X_PW = NT / 2;  // 1-KiB DMA pieces per wave and chunk
  static constexpr int PW = D_PW + X_PW;
  static constexpr int LDS_BYTES = NSLOT * SLOT_BYTES;
};

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int PW>
__device__ __forceinline__ void wait_chunks(int k) {
  switch (k) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<PW>(); break;
    default: wait_vmcnt<2 * PW>(); break;
  }
}
// 8 consecutive pixels (rows R0 .. R0 + 7 of the image) of one channel: two transposed reads of 4 rows each
template <int OFF, int ROW>
__device__ __forceinline__ h8 tr_frag(unsigned a) {
  s4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(OFF + 4 * ROW));
  const s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(h8, v);
}

template <int MT, int NT>
__global__ __launch_bounds__(64 * GW) void conv_wgrad_gemm_kernel(const WgradParams p, long long pix_per_split, int gx, int gy) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (and has no __amdgpu_buffer_rsrc_t)
  using G = WgGeom<MT, NT>;
  constexpr int CO_B = G::CO_B, J_B = G::J_B, D_ROW = G::D_ROW, X_ROW = G::X_ROW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave & 3, wn = wave >> 2;
  // XCD-aware order (conv_wgrad.hip): every (co, j) tile of pixel split z gets an id congruent to z mod 8 -- one XCD's L2 serves them
  const int ntiles = gx * gy;
  const int lin = blockIdx.x;
  const int qq = lin >> 3;
  const int tile = qq % ntiles;
  const int bz = (qq / ntiles) * 8 + (lin & 7);
  if (bz >= p.nsplit) return;
  const int bx = tile % gx, by = tile / gx;
  const int co0 = bx * CO_B, j0 = by * J_B;
  const long long M = (long long)p.B * p.OH * p.OW;
  const long long m_begin = std::min<long long>(M, (long long)bz * pix_per_split);
  const long long m_end = std::min<long long>(M, m_begin + pix_per_split);
  const int nchunks = (int)((m_end - m_begin + PKC - 1) / PKC);
  const int Jtot = p.ntaps * p.cin_pad16;

  // ---- per-lane DMA assignment ----
  // dy piece q of this wave covers LDS units (q * 8 + wave) * 64 + lane of the [32 px][CO_B] image
  constexpr int DG = D_ROW / 16, XG = X_ROW / 16;  // 16-byte groups per pixel row
  unsigned d_vo[G::D_PW];
#pragma unroll
  for (int q = 0; q < G::D_PW; ++q) {
    const int u = (q * GW + wave) * 64 + lane;
    const int px = u / DG, g = (u % DG) ^ ((px & 3) << 2);  // the group this lane fetches: the read-side swizzle, applied at the source
    const int co = co0 + g * 8;
    d_vo[q] = co < p.Cout ? (unsigned)(((long long)px * p.dy_ld + co) * 2) : 0xffffffffu;
  }
  // x piece: the lane's column group is fixed; its pixel advances by 32 per chunk
  int x_b[G::X_PW], x_oh[G::X_PW], x_ow[G::X_PW], x_dh[G::X_PW], x_dw[G::X_PW];
  unsigned x_col[G::X_PW];  // byte offset of the channel inside a pixel, ~0: column outside the layer
#pragma unroll
  for (int q = 0; q < G::X_PW; ++q) {
    const int u = (q * GW + wave) * 64 + lane;
    const int px = u / XG, g = (u % XG) ^ ((px & 3) << 2);
    const int j = j0 + g * 8;
    const int tap = j / p.cin_pad16, ci = j - tap * p.cin_pad16;
    x_col[q] = 0xffffffffu;
    x_dh[q] = x_dw[q] = 0;
    if (j < Jtot && ci < p.Cin) {
      const ConvTap td = p.taps[tap];
      x_dh[q] = td.dh;
      x_dw[q] = td.dw;
      x_col[q] = (unsigned)(ci * 2);
    }
    const long long m = m_begin + px;
    const unsigned mu = (unsigned)(m < M ? m : M - 1), tq = mu / (unsigned)p.OW;
    x_ow[q] = (int)(mu - tq * (unsigned)p.OW);
    x_b[q] = (int)(tq / (unsigned)p.OH);
    x_oh[q] = (int)(tq - (unsigned)x_b[q] * (unsigned)p.OH);
    if (m >= M) x_b[q] = p.B;  // past the last pixel: outside the buffer descriptor, reads as zero
  }
  const __amdgpu_buffer_rsrc_t rsrc_d =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.dy), (short)0, (int)(unsigned)std::min<long long>(m_end * p.dy_ld * 2, 0xffffffffLL), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<half_t*>(p.x), (short)0, (int)(unsigned)std::min<long long>((long long)p.B * p.x_bstride * 2, 0xffffffffLL), 0x00020000);

  int issued = 0, islot = 0;
  auto issue = [&]() __attribute__((always_inline)) {
    unsigned char* sb = smem + islot * G::SLOT_BYTES;
    islot = islot + 1 == NSLOT ? 0 : islot + 1;
    const unsigned sd = (unsigned)((m_begin + (long long)issued * PKC) * p.dy_ld * 2);
#pragma unroll
    for (int q = 0; q < G::D_PW; ++q) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_d, (lds_ptr_t)(sb + (q * GW + wave) * 1024), 16, d_vo[q], sd, 0, 0);
#pragma unroll
    for (int q = 0; q < G::X_PW; ++q) {
      const int ih = x_oh[q] * p.stride + x_dh[q], iw = x_ow[q] * p.stride + x_dw[q];
      const bool ok = x_col[q] != 0xffffffffu && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
      const unsigned vo = ok ? (unsigned)(((long long)x_b[q] * p.x_bstride + ((long long)ih * p.IW + iw) * p.x_ld) * 2) + x_col[q] : 0xffffffffu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(sb + G::D_BYTES + (q * GW + wave) * 1024), 16, vo, 0, 0, 0);
      // this lane's pixel of the next chunk: 32 further (OW >= 8: at most four row wraps; OH >= 4: at most one image wrap)
      x_ow[q] += PKC;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (x_ow[q] >= p.OW) {
          x_ow[q] -= p.OW;
          ++x_oh[q];
        }
      if (x_oh[q] >= p.OH) {
        x_oh[q] -= p.OH;
        ++x_b[q];
      }
    }
    ++issued;
  };

  f16v acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nchunks > 0) {
#pragma unroll
    for (int s = 0; s < NSLOT - 1; ++s)
      if (s < nchunks) issue();

    // transposed-read addresses (bytes inside a slot): 16-lane group gi = lane >> 4 reads channels 16 (gi & 1) .. + 15 of the 32-channel
    // block, pixels 8 (gi >> 1) .. + 7 of the K-step; lane 4 q + p of the group supplies row (pixel) q, channels 4 p .. 4 p + 3
    const int gi = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int prow = (gi >> 1) * 8 + tq;
    unsigned d_rd[MT], x_rd[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int gidx = (wm * MT + i) * 4 + (gi & 1) * 2 + (tp >> 1);
      d_rd[i] = lds_addr(smem) + (unsigned)(prow * D_ROW + ((gidx ^ (tq << 2)) << 4) + (tp & 1) * 8);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int gidx = (wn * NT + j) * 4 + (gi & 1) * 2 + (tp >> 1);
      x_rd[j] = lds_addr(smem) + (unsigned)(G::D_BYTES + prow * X_ROW + ((gidx ^ (tq << 2)) << 4) + (tp & 1) * 8);
    }
    h8 fa[2][MT], fb[2][NT];
#define CVX_WG_READ(SET, SO, KSTEP)                                                                \
  {                                                                                                \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) fa[SET][i] = tr_frag<(KSTEP) * 16 * D_ROW, D_ROW>(d_rd[i] + (SO)); \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) fb[SET][j] = tr_frag<(KSTEP) * 16 * X_ROW, X_ROW>(x_rd[j] + (SO)); \
  }
#define CVX_WG_MFMA(SET)                                                                           \
  {                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) _Pragma("unroll") for (int i = 0; i < MT; ++i) acc[i][j] =                            \
        __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[SET][i], fb[SET][j], acc[i][j], 0, 0, 0);        \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  }
    constexpr int NRD = 2 * (MT + NT);  // LDS reads of one fragment set
    const int issued0 = nchunks < NSLOT - 1 ? nchunks : NSLOT - 1;
    wait_chunks<G::PW>(issued0 - 1);
    workgroup_barrier();  // chunk 0 published
    CVX_WG_READ(0, 0u, 0);
    CVX_WG_READ(1, 0u, 1);
    wait_chunks<G::PW>(issued0 >= 2 ? issued0 - 2 : 0);  // chunk 1 landed
    unsigned so = 0;
    for (int c = 0; c + 1 < nchunks; ++c) {
      const unsigned sn = so + G::SLOT_BYTES == NSLOT * G::SLOT_BYTES ? 0u : so + G::SLOT_BYTES;
      workgroup_barrier();  // chunk c + 1 published; every wave is done reading chunk c - 1: its slot takes chunk c + 3
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NRD) : "memory");
      CVX_WG_MFMA(0);
      if (issued < nchunks) issue();
      CVX_WG_READ(0, sn, 0);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NRD) : "memory");
      CVX_WG_MFMA(1);
      CVX_WG_READ(1, sn, 1);
      const int behind = nchunks - 3 - c;
      wait_chunks<G::PW>(behind < 0 ? 0 : behind < 1 ? behind : 1);
      so = sn;
    }
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NRD) : "memory");
    CVX_WG_MFMA(0);
    wait_lgkm();
    CVX_WG_MFMA(1);
#undef CVX_WG_READ
#undef CVX_WG_MFMA
  }

  // ---- the split's slab: lane holds column j = .. + (lane & 31), rows co = .. + (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
  float* slab = p.slabs + (long long)bz * p.Cout * Jtot;
  const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int jj = j0 + (wn * NT + j) * 32 + lr;
    if (jj < Jtot) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (co < p.Cout) slab[(long long)co * Jtot + jj] = acc[i][j][r];
        }
    }
  }
#endif
}

template <int MT, int NT>
int launch_wg(const WgradParams& p, hipStream_t st) {
  using G = WgGeom<MT, NT>;
  const long long M = (long long)p.B * p.OH * p.OW;
  long long per = (M + p.nsplit - 1) / p.nsplit;
  per = ((per + PKC - 1) / PKC) * PKC;
  const int gx = cvx_cdiv(p.Cout, G::CO_B), gy = cvx_cdiv(p.ntaps * p.cin_pad16, G::J_B);
  const int zpad = (p.nsplit + 7) / 8 * 8;  // surplus workgroups (split >= nsplit) exit at once
  static unsigned long long optin_mask = 0;
  CVX_TRY(cvx_lds_optin((const void*)conv_wgrad_gemm_kernel<MT, NT>, G::LDS_BYTES, &optin_mask));
  hipLaunchKernelGGL((conv_wgrad_gemm_kernel<MT, NT>), dim3(gx * gy * zpad), dim3(64 * GW), G::LDS_BYTES, st, p, per, gx, gy);
  return 0;
}

}  // namespace

// Layers the kernel takes: 128+ output channels and 256+ weight columns, dy dense (a BatchNorm layer's dy buffer: pixel rows advance by a
// scalar), output maps of at least 8 x 4 (the per-chunk pixel step wraps by compares), operands addressable with 32-bit byte offsets.
bool cvx_conv_wgrad_gemm_supported(const WgradParams& p) {
  static const bool off = cvx_tune_set("CVX_NO_WGRAD_GEMM");
  if (off) return false;
  const long long M = (long long)p.B * p.OH * p.OW;
  const long long ohw = (long long)p.OH * p.OW;
  static const int cmin = cvx_tune_int("CVX_WGG_CMIN", 128), jmin = cvx_tune_int("CVX_WGG_JMIN", 256);
  return p.Cout >= cmin && p.Cout % 8 == 0 && p.ntaps * p.cin_pad16 >= jmin && p.dy_ld % 8 == 0 && p.x_ld % 8 == 0 && p.Cin % 8 == 0 &&
         p.dy_bstride == ohw * p.dy_ld && p.OW >= 8 && p.OH >= 4 && M * p.dy_ld * 2 < (1LL << 32) && (long long)p.B * p.x_bstride * 2 < (1LL << 32);
}

// (co, j) tile: 256 x 256 where the layer has at least that much of both, 128 x 128 otherwise
void cvx_conv_wgrad_gemm_tile(int cout, int jtot, int* co_b, int* j_b) {
  static const int force = cvx_tune_int("CVX_WGG_TILE", 0);  // tuning build: 1 = 256 x 256, 2 = 128 x 128
  const bool big = force ? force == 1 : (cout >= 256 && jtot >= 512 && cout % 256 == 0);
  *co_b = *j_b = big ? 256 : 128;
}

int cvx_conv_wgrad_gemm_launch(const WgradParams& p, hipStream_t st) {
  int co_b, j_b;
  cvx_conv_wgrad_gemm_tile(p.Cout, p.ntaps * p.cin_pad16, &co_b, &j_b);
  if (co_b == 256)
    CVX_TRY((launch_wg<2, 4>(p, st)));
  else
    CVX_TRY((launch_wg<1, 2>(p, st)));
  CVX_HIP(hipGetLastError());
  return 0;
}
