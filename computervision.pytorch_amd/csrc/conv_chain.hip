// Tile-resident convolution chains on gfx950: see conv_chain.h for the program model.
//
// Kernel structure (512 threads = 8 waves, two per SIMD, one workgroup per CU):
//   * LDS = [weight ring: CHAIN_RING_SLOTS slots of KSUB K-steps x RR rows x 64 B | 1 KiB dump | planes | fp32 parameter table];
//   * the weight stream of ALL passes is one sequence of chunks; every wave issues its PB 1-KiB LDS-DMA pieces of chunk
//     t + SLOTS - 1 right after the barrier of chunk t (which also proves slot (t - 1) % SLOTS free), and waits for its own
//     pieces of chunk t with a COUNTED vmcnt before that barrier -- the L2->LDS latency of the weights is paid once per
//     launch, not once per layer, and never drains inside the K loops;
//   * the pixel operand never moves: K-step (segment, channel chunk) of output pixel q is one ds_read_b128 at
//     in_off[q] + segoff + chunk of the resident plane (pixel stride = 2 mod 4 in 16-byte units: conflict-free under the gfx950 ds_read_b128 lane groups);
//   * waves are arranged WM x WN over (16-pixel sub-tiles) x (16-channel tiles); v_mfma_f32_16x16x32_f16, weights as the A
//     operand so that a lane ends up with 4 consecutive channels of one pixel (8-byte plane stores).
#include <algorithm>
#include <cstring>
#include <type_traits>
#include <vector>

#include "conv_chain.h"
#include "conv_tile_common.h"

namespace {
using namespace cvx_tile;

constexpr int SLOTS = CHAIN_RING_SLOTS;
constexpr int KS = 16;  // K per MFMA step (v_mfma_f32_32x32x16_f16)
constexpr int NW = 8;   // waves per workgroup: two per SIMD -- one wave's address arithmetic, LDS requests and epilogue overlap the other's MFMAs
                        // (a single wave per SIMD issues in order: its ~250 cycles of non-MFMA work per K-step left the matrix pipe idle 55 % of the time)

typedef float f16v __attribute__((ext_vector_type(16)));

// Ring image of one weight chunk (KSUB K-steps of 16): [K-step][k-half h][RR rows][8 halves] -- lane (row r, half h) of the
// MFMA weight operand reads 16 bytes at ((ks * 2 + h) * RR + row) * 16: consecutive lanes, consecutive 16-byte slots, no bank
// conflict.  The weights are PRE-PACKED in this order in global memory (chain_pack_kernel, once per forward), so a chunk is
// one contiguous block and every LDS-DMA piece reads 1 KiB of consecutive bytes.
template <int MT, int NT, int KSUB>
struct ChainGeom {
  static constexpr int RR = 32 * NT;                  // weight rows per pass
  static constexpr int STEP_HALVES = 2 * RR * 8;      // one K-step of weights
  static constexpr int SLOT_HALVES = KSUB * STEP_HALVES;
  static constexpr int NPIECE = SLOT_HALVES / 512;    // 1 KiB pieces per chunk
  static constexpr int PB = (NPIECE + NW - 1) / NW;   // DMA instructions per wave per chunk
  static constexpr int RING_BYTES = SLOTS * SLOT_HALVES * 2 + 1024;  // + dump piece
};

__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// MFMA fragment read the COMPILER does not track: its own waitcnt insertion loses the pending-read state at the K loop's joins
// and puts lgkmcnt(0) in front of every MFMA block, i.e. also waits for the prefetch that was just issued.  With the read in
// inline asm the wait is placed by hand -- after the MFMA block the prefetch overlaps (each use is fenced by sched_barrier).
__device__ __forceinline__ h8 lds_read_frag(const void* p) {
  h8 v;
  const unsigned a = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a));
  return v;
}

// exact u / d for u < 2^20, d < 2^11 with m = ceil(2^32 / d): one v_mul_hi_u32 instead of the ~20-instruction division sequence
__device__ __forceinline__ unsigned fast_div(unsigned u, unsigned m) { return __umulhi(u, m); }

// fields of a pass the epilogue needs, copied into registers once per pass (a reference into the descriptor would be re-read
// from memory at every use: the compiler must assume the LDS-DMA intrinsic may have written it)
struct EpiArgs {
  int rows, par_off, par_off2, act, out_kind, has_res;
  float* out32;  // already advanced by the image
};

// ---- device helpers (plain force-inlined functions on references to the kernel's locals: closures that captured the mutable
// cursors by reference ended up in scratch memory, with a vmcnt(0) drain at every access) ----
struct RingCursor {  // issue side of the weight ring: chunk `ic` of pass `ip` goes into slot `tissue % SLOTS` next
  int ip, ic, tissue, nchunks, npass;
  const half_t* wpk;
};
struct KCursor {  // plane unit offset of the next K-step (2 chunks of one segment), advanced in scalar registers
  int ch, seg, soff, dw;
};
struct PassK {  // K-loop constants of a pass
  int P, nseg, inW, inps, nchunks;
  bool taps3;
};

template <typename G>
__device__ __forceinline__ void ring_issue(RingCursor& rc, const ChainDesc& d, half_t* ring, half_t* dump, const half_t* zeros, int wave, int lane) {
  half_t* slot = ring + (rc.tissue % SLOTS) * G::SLOT_HALVES;
  ++rc.tissue;
  if (rc.ip < rc.npass) {
    const half_t* src = rc.wpk + (long long)rc.ic * G::SLOT_HALVES + lane * 8;
#pragma unroll
    for (int q = 0; q < G::PB; ++q) {
      const int piece = q * NW + wave;
      const bool real = piece < G::NPIECE;
      __builtin_amdgcn_global_load_lds((gbl_void_ptr)(real ? src + piece * 512 : zeros), (lds_void_ptr)(real ? slot + piece * 512 : dump), 16, 0, 0);
    }
    if (++rc.ic == rc.nchunks) {
      rc.ic = 0;
      ++rc.ip;
      if (rc.ip < rc.npass) {
        rc.wpk = d.pass[rc.ip].wpk;
        rc.nchunks = d.pass[rc.ip].nchunks;
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < G::PB; ++q) __builtin_amdgcn_global_load_lds((gbl_void_ptr)zeros, (lds_void_ptr)dump, 16, 0, 0);
  }
}

// beyond K the packed weights are zero and the offset freezes at a valid one
__device__ __forceinline__ int koff_next(KCursor& k, const PassK& pk) {
  const int koff = k.soff + k.ch;
  k.ch += 2;
  if (k.ch >= pk.P) {
    k.ch = 0;
    ++k.seg;
    if (k.seg >= pk.nseg) {
      k.seg = pk.nseg;
      k.soff = 0;
    } else if (pk.taps3) {
      if (++k.dw == 3) {
        k.dw = 0;
        k.soff += (pk.inW - 2) * pk.inps;
      } else {
        k.soff += pk.inps;
      }
    }  // (a 1x1 stage has one segment: concat segments are not compiled in -- a select over per-segment offsets became an indexed
       //  scratch load with a vmcnt(0) drain inside the K loop)
  }
  return koff;
}

// LOAD: plane fill by LDS-DMA.  Unit u of the plane = (pixel u / ps, chunk u % ps); zero page outside the image / pad units
__device__ __forceinline__ void plane_load(const ChainLoad& Lr, unsigned char* planes, half_t* dump, const half_t* zeros, int b, int ty0, int tx0, int wave,
                                           int lane) {
  const int kind = Lr.kind, ps = Lr.ps, PW = Lr.PW, nunits = Lr.nunits, P = Lr.P, IH = Lr.IH, IW = Lr.IW, up = Lr.up, ld = Lr.ld, base = Lr.base,
            per_wave = Lr.per_wave;
  const unsigned m_ps = 0xffffffffu / (unsigned)(ps > 0 ? ps : 1) + 1u, m_pw = 0xffffffffu / (unsigned)(PW > 0 ? PW : 1) + 1u;
  const half_t* src = reinterpret_cast<const half_t*>(Lr.src) + (kind == 1 ? 0 : (long long)b * Lr.bstride);
  const int oy = ty0 * Lr.scale + Lr.y0, ox = tx0 * Lr.scale + Lr.x0, sw = IW >> up;
  for (int k = 0; k < per_wave; ++k) {
    const int piece = k * NW + wave;
    const int u = piece * 64 + lane;
    const half_t* g = zeros;
    if (kind == 1) {
      if (u < nunits) g = src + (long long)u * 8;
    } else if (u < nunits) {
      const unsigned pix = fast_div((unsigned)u, m_ps);
      const int ch = u - (int)pix * ps;
      const unsigned py = fast_div(pix, m_pw);
      const int px = (int)(pix - py * (unsigned)PW);
      const int iy = oy + (int)py, ix = ox + px;
      if (ch < P && (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW) g = src + ((long long)(iy >> up) * sw + (ix >> up)) * ld + ch * 8;
    }
    unsigned char* dst = piece * 64 < nunits ? planes + 16 * ((long long)base + piece * 64) : reinterpret_cast<unsigned char*>(dump);
    __builtin_amdgcn_global_load_lds((gbl_void_ptr)g, (lds_void_ptr)dst, 16, 0, 0);
  }
}

// K loop + epilogue of one pass for a compile-time register tile MTU x NTU (32-pixel sub-tiles x 32-channel tiles)
template <typename G, int MT, int KSUB, int MTU, int NTU>
__device__ __forceinline__ void run_pass_t(const ChainDesc& d, RingCursor& rc, const PassK& pk, const EpiArgs& E, const int (&in_off)[MT],
                                           const int (&pixo)[MT], const int (&pixr)[MT], const int (&inmask)[MT], half_t* ring, half_t* dump,
                                           unsigned char* planes, const float* par, const half_t* zeros, int t, int wave, int lane, int dbg,
                                           unsigned long long* clk_slot) {
  constexpr int NWAIT = (SLOTS - 2) * G::PB;
  const int lr = lane & 31, lh = lane >> 5;
  f16v acc[MTU][NTU];
#pragma unroll
  for (int i = 0; i < MTU; ++i)
#pragma unroll
    for (int j = 0; j < NTU; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const half_t* wlane = ring + (lh * G::RR + lr) * 8;
  KCursor kc{0, 0, 0, 0};
  h8 xa[2][MTU], wb[2][NTU];
  for (int c = 0;;) {
    if (!(dbg & 2)) ring_issue<G>(rc, d, ring, dump, zeros, wave, lane);
    const half_t* slot = wlane + ((t + c) % SLOTS) * G::SLOT_HALVES;
    {
      const int koff = koff_next(kc, pk);
#pragma unroll
      for (int i = 0; i < MTU; ++i) xa[0][i] = lds_read_frag(planes + 16 * (long long)(in_off[i] + koff));
#pragma unroll
      for (int j = 0; j < NTU; ++j) wb[0][j] = lds_read_frag(slot + j * 32 * 8);
      wait_lgkm0();
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < KSUB; ++ks) {
      if (ks + 1 < KSUB && !(dbg & 32)) {  // fragments of the next K-step are requested before this step's MFMAs and waited for after them
        const int koff = koff_next(kc, pk);
#pragma unroll
        for (int i = 0; i < MTU; ++i) xa[(ks + 1) & 1][i] = lds_read_frag(planes + 16 * (long long)(in_off[i] + koff));
#pragma unroll
        for (int j = 0; j < NTU; ++j) wb[(ks + 1) & 1][j] = lds_read_frag(slot + (ks + 1) * G::STEP_HALVES + j * 32 * 8);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NTU; ++j)
#pragma unroll
        for (int i = 0; i < MTU; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[ks & 1][j], xa[ks & 1][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < KSUB) {
        wait_lgkm0();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (++c == pk.nchunks) break;
    if (dbg & 2) wait_vmcnt<0>();
    else if (!(dbg & 8)) wait_vmcnt<NWAIT>();
    if (!(dbg & 16)) workgroup_barrier();
  }
  if (clk_slot && threadIdx.x == 0) *clk_slot = wall_clock64();
  if (dbg & 4) return;
  // ---- epilogue: lane holds pixel lr of sub-tile i, channels 32 j + 8 g + 4 lh + (0..3), g = 0..3, in acc[i][j][4 g + (0..3)] ----
  // Branch-free per element: a sub-tile lane beyond the region is a clamped duplicate of the region's last pixel (same value, same
  // address: the store is harmless), an out-of-image pixel stores zeros (the next 3x3 stage's padding); only whole 8-channel
  // groups beyond the pass's rows are skipped (wave-uniform).
  if (E.out_kind == 1) {
#pragma unroll
    for (int j = 0; j < NTU; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (j * 32 + g * 8 >= E.rows) continue;
        const int n0 = j * 32 + g * 8 + lh * 4;
        const f4 bb = *reinterpret_cast<const f4*>(par + E.par_off + n0);
#pragma unroll
        for (int i = 0; i < MTU; ++i) {
          f4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[i][j][4 * g + r] + bb[r];
          if (inmask[i]) *reinterpret_cast<f4*>(E.out32 + pixo[i] + n0) = v;
        }
      }
    return;
  }
  const bool silu = E.act == 0, relu = E.act == 1;
#pragma unroll
  for (int j = 0; j < NTU; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (j * 32 + g * 8 >= E.rows) continue;
      const int n0 = j * 32 + g * 8 + lh * 4;
      const f4 sc = *reinterpret_cast<const f4*>(par + E.par_off + n0);
      const f4 sh = *reinterpret_cast<const f4*>(par + E.par_off2 + n0);
      f4 v[MTU];
#pragma unroll
      for (int i = 0; i < MTU; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[i][r] = acc[i][j][4 * g + r] * sc[r] + sh[r];
      if (silu) {
#pragma unroll
        for (int i = 0; i < MTU; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[i][r] = cvx_silu(v[i][r]);
      } else if (relu) {
#pragma unroll
        for (int i = 0; i < MTU; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[i][r] = fmaxf(v[i][r], 0.f);
      }
      if (E.has_res) {
#pragma unroll
        for (int i = 0; i < MTU; ++i) {
          const h4 rr = *reinterpret_cast<const h4*>(planes + pixr[i] + 2 * n0);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[i][r] += (float)rr[r];
        }
      }
#pragma unroll
      for (int i = 0; i < MTU; ++i) {
        h4 o = {(half_t)v[i][0], (half_t)v[i][1], (half_t)v[i][2], (half_t)v[i][3]};
        o = inmask[i] ? o : h4{0, 0, 0, 0};
        *reinterpret_cast<h4*>(planes + pixo[i] + 2 * n0) = o;
      }
    }
}

// NW waves: wave w owns the 32-pixel sub-tiles w, w + NW, ... of the pass's region and ALL of its channel tiles; the fragments of
// K-step s + 1 are requested before the MFMAs of s.  (A 256-register budget also keeps the accumulators in VGPRs: with 512 the compiler
// parked them in AGPRs and copied all of them to VGPRs and back around every barrier of the K loop, 96 + 96 moves per chunk.)
template <int MT, int NT, int KSUB>
__global__ __launch_bounds__(64 * NW) void conv_chain_kernel(const ChainDesc* __restrict__ dp, float* out32_base) {
  using G = ChainGeom<MT, NT, KSUB>;
  constexpr int NWAIT = (SLOTS - 2) * G::PB;
  static_assert(NWAIT <= 63, "vmcnt field");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half_t* ring = reinterpret_cast<half_t*>(smem);
  half_t* dump = ring + SLOTS * G::SLOT_HALVES;
  unsigned char* planes = smem + G::RING_BYTES;  // unit u = planes + 16 * u
  const ChainDesc& d = *dp;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lr = lane & 31, lh = lane >> 5;  // MFMA 32x32x16 operand lane map: row / column lr, k-half lh

  // tile of this workgroup
  const int bid = blockIdx.x;
  const int tx = bid % d.tiles_x;
  const int t2 = bid / d.tiles_x;
  const int ty = t2 % d.tiles_y;
  const int b = t2 / d.tiles_y;
  const int ty0 = ty * d.TH, tx0 = tx * d.TW;
  const float* par = reinterpret_cast<const float*>(planes + 16 * (long long)d.par_base);
  unsigned long long* const clk = d.clk ? d.clk + (long long)blockIdx.x * 32 : nullptr;
#define CHAIN_MARK(slot)                                              \
  do {                                                                \
    if (clk && tid == 0 && (slot) < 32) clk[(slot)] = wall_clock64(); \
  } while (0)
  CHAIN_MARK(0);
  if (clk && tid == 0) clk[28] = clock64();
  const int npass = d.npass, nload = d.nload;
  const int dbg = d.dbg;
  const half_t* const zeros = d.zeros;

  RingCursor rc{0, 0, 0, d.pass[0].nchunks, npass, d.pass[0].wpk};
  for (int l = 0; l < nload; ++l)
    if (d.load[l].at_pass == 0) plane_load(d.load[l], planes, dump, zeros, b, ty0, tx0, wave, lane);
#pragma unroll
  for (int s = 0; s < SLOTS - 1; ++s) ring_issue<G>(rc, d, ring, dump, zeros, wave, lane);
  CHAIN_MARK(1);

  int t = 0;  // chunk being computed
  for (int p = 0; p < npass; ++p) {
    // ---- per-pass constants, in registers ----
    PassK pk;
    int p_npix, p_rows, p_drain;
    // per lane and sub-tile: plane unit of the pixel operand, destination / residual offsets of the epilogue (plane: bytes, fp32
    // rows: floats), "pixel lies inside the image" -- computed here so that the epilogue keeps no per-pass scalars alive
    int in_off[MT], pixo[MT], pixr[MT], inmask[MT];
    EpiArgs E;
    {
      const ChainPass& P = d.pass[p];
      p_npix = P.npix, p_rows = P.rows, p_drain = P.drain;
      pk.nchunks = P.nchunks, pk.inW = P.in_W, pk.inps = P.in_ps, pk.P = P.P, pk.nseg = P.nseg;
      pk.taps3 = P.ksz == 3;
      const int p_RW = P.RW, p_s = P.s, p_cy = P.cy, p_cx = P.cx, p_inbase = P.in_base;
      const unsigned m_rw = 0xffffffffu / (unsigned)p_RW + 1u;
      const int iy0 = ty0 * P.img_scale + P.img_y0, ix0 = tx0 * P.img_scale + P.img_x0;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int q = (wave + i * NW) * 32 + lr;
        const unsigned qc = (unsigned)(q < p_npix ? q : p_npix - 1);  // beyond the region: a duplicate of its last pixel
        const int qy = (int)fast_div(qc, m_rw);
        const int qx = (int)qc - qy * p_RW;
        in_off[i] = p_inbase + ((p_cy + qy * p_s) * pk.inW + p_cx + qx * p_s) * pk.inps + lh;  // + k-half of this lane
        const int iy = iy0 + qy, ix = ix0 + qx;
        inmask[i] = ((unsigned)iy < (unsigned)P.IH && (unsigned)ix < (unsigned)P.IW) ? -1 : 0;
        pixo[i] = P.out_kind == 1 ? (iy * P.IW + ix) * P.out_ld + P.out_c0
                                  : 16 * (P.out_base + ((P.oy + qy) * P.out_W + P.ox + qx) * P.out_ps) + 2 * P.out_c0;
        pixr[i] = 16 * (P.res_base + ((P.ry + qy) * P.res_W + P.rx + qx) * P.res_ps) + 2 * P.res_c0;
      }
      E.rows = P.rows, E.par_off = P.par_off, E.par_off2 = P.par_off2, E.act = P.act, E.out_kind = P.out_kind, E.has_res = P.has_res;
      E.out32 = (P.out32 ? P.out32 : out32_base) + (P.out_kind == 1 ? P.out32_off + (long long)b * P.out_bstride : 0);
    }
    const int subtiles = (p_npix + 31) >> 5;
    const int mt_pass = min(MT, (subtiles + NW - 1) / NW);  // sub-tiles per wave this pass (workgroup-uniform)
    const int nt_pass = min(NT, (p_rows + 31) >> 5);   // channel tiles
    const bool active = wave < subtiles && !(dbg & 1);
    bool late = false;
    for (int l = 0; l < nload; ++l) late = late || (p > 0 && d.load[l].at_pass == p);

    // ---- top of the pass: chunk t has landed for every wave, the previous pass's plane stores are visible ----
    if (p_drain || (dbg & 2)) wait_vmcnt<0>();
    else wait_vmcnt<NWAIT>();
    wait_lgkm0();
    workgroup_barrier();
    if (t == 0) CHAIN_MARK(2);
    if (late)
      for (int l = 0; l < nload; ++l)
        if (d.load[l].at_pass == p) plane_load(d.load[l], planes, dump, zeros, b, ty0, tx0, wave, lane);

    if (!active) {  // keeps the barriers and its share of the weight DMA
      for (int c = 0;;) {
        if (!(dbg & 2)) ring_issue<G>(rc, d, ring, dump, zeros, wave, lane);
        if (++c == pk.nchunks) break;
        if (dbg & 2) wait_vmcnt<0>();
        else wait_vmcnt<NWAIT>();
        workgroup_barrier();
      }
    } else {
      bool done = false;
      unsigned long long* const clk_k = (clk && 4 + 2 * p < 28) ? clk + 4 + 2 * p : nullptr;  // end of the K loop (even slots)
#define CHAIN_TRY(MTU, NTU)                                                                                                                    \
  if (!done && MTU <= MT && NTU <= NT && mt_pass <= MTU && nt_pass <= NTU) {                                                                   \
    run_pass_t<G, MT, KSUB, (MTU <= MT ? MTU : MT), (NTU <= NT ? NTU : NT)>(d, rc, pk, E, in_off, pixo, pixr, inmask, ring, dump, planes, par, \
                                                                             zeros, t, wave, lane, dbg, clk_k);                                     \
    done = true;                                                                                                                               \
  }
      // smallest compiled register tile that covers this pass (every variant is one more copy of the K loop: keep the list short)
      CHAIN_TRY(1, 1) CHAIN_TRY(2, 1) CHAIN_TRY(3, 1)
      CHAIN_TRY(1, 2) CHAIN_TRY(2, 2) CHAIN_TRY(3, 2)
      CHAIN_TRY(1, 3) CHAIN_TRY(2, 3) CHAIN_TRY(3, 3)
      if (NT > 3) {
        CHAIN_TRY(1, NT) CHAIN_TRY(2, NT) CHAIN_TRY(3, NT)
      }
      if (!done) run_pass_t<G, MT, KSUB, MT, NT>(d, rc, pk, E, in_off, pixo, pixr, inmask, ring, dump, planes, par, zeros, t, wave, lane, dbg, clk_k);
#undef CHAIN_TRY
    }
    CHAIN_MARK(3 + 2 * p);
    t += pk.nchunks;
  }

  // ---- STORE: plane regions -> global, whole pixel rows ----
  CHAIN_MARK(30);
  if (d.nstore > 0) {
    wait_lgkm0();
    workgroup_barrier();
    for (int s = 0; s < d.nstore; ++s) {
      const ChainStore& Sr = d.store[s];
      const int RW = Sr.RW, P = Sr.P, total = RW * Sr.RH * P, OH = Sr.OH, OW = Sr.OW, ld = Sr.ld, sbase = Sr.base, PW = Sr.PW, sps = Sr.ps,
                py0 = Sr.py0, px0 = Sr.px0;
      const unsigned m_p = 0xffffffffu / (unsigned)P + 1u, m_rw = 0xffffffffu / (unsigned)RW + 1u;
      half_t* dst = Sr.dst + (long long)b * Sr.bstride;
      const int oy = ty0 * Sr.scale + Sr.y0, ox = tx0 * Sr.scale + Sr.x0;
      for (int u = tid; u < total; u += 64 * NW) {
        const unsigned pix = fast_div((unsigned)u, m_p);
        const int ch = u - (int)pix * P;
        const unsigned qy = fast_div(pix, m_rw);
        const int qx = (int)(pix - qy * (unsigned)RW);
        const int iy = oy + (int)qy, ix = ox + qx;
        if ((unsigned)iy < (unsigned)OH && (unsigned)ix < (unsigned)OW) {
          const h8 v = *reinterpret_cast<const h8*>(planes + 16 * (long long)(sbase + ((py0 + (int)qy) * PW + px0 + qx) * sps + ch));
          *reinterpret_cast<h8*>(dst + ((long long)iy * OW + ix) * ld + ch * 8) = v;
        }
      }
    }
  }
  wait_vmcnt<0>();  // the surplus ring prefetches (dump pieces) retire before the LDS is released
  CHAIN_MARK(31);
  if (clk && tid == 0) clk[29] = clock64();
#undef CHAIN_MARK
}

// Weight pre-pack: [rows][K] fp16 (row pitch src_ld) -> chunk images [chunk][K-step][k-half][RR rows][8] (zero rows / zero K tail).
__global__ void chain_pack_kernel(const ChainPackJob* __restrict__ jobs) {
  const ChainPackJob J = jobs[blockIdx.y];
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < J.units; u += gridDim.x * blockDim.x) {
    const int row = u % J.RR;
    const int hk = u / J.RR;  // global K-step * 2 + half
    const int k = hk * 8;
    h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < J.rows && k < J.K) v = *reinterpret_cast<const h8*>(J.src + (long long)row * J.src_ld + k);
    *reinterpret_cast<h8*>(J.dst + (long long)u * 8) = v;
  }
}

// ---- compiled configurations ----
struct CfgInfo {
  int MT, NT, KSUB;  // 32-pixel sub-tiles per wave (4 waves), 32-channel tiles per pass, K-steps (of 16) per ring chunk
};
constexpr CfgInfo kCfg[] = {
    {2, 2, 4},  // 0: 64 rows / pass, <= 512 region pixels, 8 KiB chunks
    {1, 5, 2},  // 1: 160 rows, <= 256 pixels, 10 KiB chunks (Detect: 144 rows in one pass)
    {2, 1, 8},  // 2: 32 rows, <= 512 pixels
    {1, 4, 2},  // 3: 128 rows, <= 256 pixels
    {2, 3, 2},  // 4: 96 rows, <= 512 pixels
    {2, 4, 2},  // 5: 128 rows, <= 512 pixels
};
constexpr int kNumCfg = sizeof(kCfg) / sizeof(kCfg[0]);

template <int I>
int launch_cfg_i(const ChainPlan& plan, hipStream_t stream, float* out32_base) {
  constexpr CfgInfo c = kCfg[I];
  auto kern = conv_chain_kernel<c.MT, c.NT, c.KSUB>;
  static unsigned long long optin_mask = 0;
  CVX_TRY(cvx_lds_optin((const void*)kern, 160 * 1024, &optin_mask));
  hipLaunchKernelGGL(kern, dim3(plan.blocks), dim3(64 * NW), plan.lds_bytes, stream, (const ChainDesc*)plan.d_desc, out32_base);
  return 0;
}

int ring_bytes_of(int cfg) {
  const CfgInfo& c = kCfg[cfg];
  return SLOTS * c.KSUB * 2 * (32 * c.NT) * 8 * 2 + 1024;
}

inline int odd_stride(int P) { return cvx_chain_pixel_units(P * 8); }

}  // namespace

int cvx_chain_launch(const ChainPlan& plan, hipStream_t stream, float* out32_base) {
  CVX_CHECK(plan.d_desc && plan.cfg >= 0 && plan.cfg < kNumCfg, "chain: plan not built");
  switch (plan.cfg) {
    case 0: CVX_TRY(launch_cfg_i<0>(plan, stream, out32_base)); break;
    case 1: CVX_TRY(launch_cfg_i<1>(plan, stream, out32_base)); break;
    case 2: CVX_TRY(launch_cfg_i<2>(plan, stream, out32_base)); break;
    case 3: CVX_TRY(launch_cfg_i<3>(plan, stream, out32_base)); break;
    case 4: CVX_TRY(launch_cfg_i<4>(plan, stream, out32_base)); break;
    default: CVX_TRY(launch_cfg_i<5>(plan, stream, out32_base)); break;
  }
  CVX_HIP(hipGetLastError());
  return 0;
}

int cvx_chain_pack_jobs(const ChainPackJob* d_jobs, int njobs, int max_job_units, hipStream_t stream) {
  CVX_CHECK(d_jobs && njobs > 0, "chain: no pack jobs");
  const int gx = std::min(64, (max_job_units + 255) / 256);
  hipLaunchKernelGGL(chain_pack_kernel, dim3(gx, njobs), dim3(256), 0, stream, d_jobs);
  CVX_HIP(hipGetLastError());
  return 0;
}
int cvx_chain_pack(const ChainPlan& plan, hipStream_t stream) {
  CVX_CHECK(plan.d_jobs && plan.njobs > 0, "chain: plan not built");
  return cvx_chain_pack_jobs((const ChainPackJob*)plan.d_jobs, plan.njobs, plan.max_job_units, stream);
}

double cvx_chain_cost_us(const ChainSpec& sp, const ChainPlan& plan) {
  const CfgInfo& c = kCfg[plan.cfg];
  const int RR = 32 * c.NT;
  double clk = 0;
  for (int s = 0; s < sp.nstages; ++s) {
    const ChainStageSpec& st = sp.stages[s];
    const int subt = (st.RH * st.RW + 31) / 32;
    const int mt = std::min(c.MT, (subt + NW - 1) / NW);
    const int K = st.k * st.k * (1 + st.nextra) * st.cin;
    const int ksteps = ((K + KS * c.KSUB - 1) / (KS * c.KSUB)) * c.KSUB;
    for (int r0 = 0; r0 < st.cout; r0 += RR) {
      const int nt = (std::min(RR, st.cout - r0) + 31) / 32;
      clk += (double)ksteps * 2 * mt * nt * 32 / 0.45;  // two waves per SIMD, 32 cycles per MFMA
      clk += 2500.0 * mt * nt;                           // epilogue
    }
  }
  const double per_wg = clk / 2300.0 + 6.0;  // us at ~2.3 GHz + prologue (plane load) and store
  const int rounds = (plan.blocks + 255) / 256;
  return per_wg * rounds + 2.0;
}

// Turns a ChainSpec into the device descriptor: lays the planes out in LDS (odd pixel strides, aliases), splits every stage
// into passes of at most RR weight rows, gathers the per-channel epilogue parameters into one table (a raw LOAD), checks that
// every read stays inside what an earlier LOAD / pass produced, and picks the compiled configuration with the fewest passes
// (then the least wasted weight rows) that covers the largest region and fits the LDS.
int cvx_chain_plan(const ChainSpec& sp, ChainPlan* out, void** d_alloc, bool dry_run) {
  CVX_CHECK(sp.nplanes > 0 && sp.nplanes <= 8 && sp.nstages > 0 && sp.nloads > 0, "chain: empty spec");
  CVX_CHECK(sp.nstages <= CHAIN_MAX_PASSES && sp.nloads + sp.nstages <= CHAIN_MAX_LOADS && sp.nstores <= CHAIN_MAX_STORES, "chain: spec too long");
  // ---- planes ----
  int base[8], ps[8], units = 0;
  struct Rect {
    int y0, x0, y1, x1;  // produced pixels [y0, y1) x [x0, x1) in plane coordinates
  } valid[8];
  for (int i = 0; i < sp.nplanes; ++i) {
    const ChainPlaneSpec& pl = sp.planes[i];
    CVX_CHECK(pl.C % 8 == 0 && pl.C > 0 && pl.PH > 0 && pl.PW > 0, "chain: plane channels must be a positive multiple of 8");
    ps[i] = odd_stride(pl.C / 8);
    // a LOAD writes whole 64-unit DMA pieces: every plane owns its size rounded up to 64 units, so the tail of the last piece
    // never lands in a neighbour
    const int need = (pl.PH * pl.PW * ps[i] + 63) & ~63;
    if (pl.alias >= 0) {
      CVX_CHECK(pl.alias < i && pl.alias_off >= 0 && sp.planes[pl.alias].alias < 0, "chain: alias must name an earlier, un-aliased plane");
      const ChainPlaneSpec& a = sp.planes[pl.alias];
      CVX_CHECK(pl.alias_off + need <= ((a.PH * a.PW * ps[pl.alias] + 63) & ~63), "chain: aliased plane does not fit inside its host");
      base[i] = base[pl.alias] + pl.alias_off;
    } else {
      base[i] = units;
      units += need;
    }
    valid[i] = Rect{0, 0, 0, 0};
  }
  for (int l = 0; l < sp.nloads; ++l) valid[sp.loads[l].plane] = Rect{0, 0, sp.planes[sp.loads[l].plane].PH, sp.planes[sp.loads[l].plane].PW};
  // parameter table: one segment per stage, [scale cout | shift cout] or [bias cout], each a whole number of DMA pieces (256 floats)
  int seg_off[CHAIN_MAX_PASSES], seg_floats[CHAIN_MAX_PASSES], par_floats = 0;
  for (int s = 0; s < sp.nstages; ++s) {
    seg_off[s] = par_floats;
    seg_floats[s] = (sp.stages[s].out_plane >= 0 ? 2 : 1) * sp.stages[s].cout;
    par_floats += (seg_floats[s] + 255) & ~255;
  }
  const int par_base = units;
  units += par_floats / 4;

  // ---- configuration ----
  int max_pix = 0;
  for (int s = 0; s < sp.nstages; ++s) max_pix = std::max(max_pix, sp.stages[s].RH * sp.stages[s].RW);
  const int subtiles = (max_pix + 31) / 32;
  int cfg = -1;
  long long best = -1;
  for (int c = 0; c < kNumCfg; ++c) {
    if (NW * kCfg[c].MT < subtiles) continue;
    if (ring_bytes_of(c) + units * 16 > 160 * 1024) continue;
    const int RRc = 32 * kCfg[c].NT;
    int np = 0, waste = 0;  // passes; weight rows streamed but not used
    for (int s = 0; s < sp.nstages; ++s) {
      const int n = (sp.stages[s].cout + RRc - 1) / RRc;
      np += n;
      waste += n * RRc - sp.stages[s].cout;
    }
    if (np > CHAIN_MAX_PASSES) continue;
    const long long score = (long long)np * 1000000 + waste * 1000 + kCfg[c].MT * kCfg[c].NT;
    if (best < 0 || score < best) {
      best = score;
      cfg = c;
    }
  }
  CVX_CHECK(cfg >= 0, "chain: no compiled configuration covers " + std::to_string(subtiles) + " sub-tiles with " + std::to_string(units * 16) +
                          " bytes of planes");
  const int RR = 32 * kCfg[cfg].NT;
  const int KSUB = kCfg[cfg].KSUB;
  out->cfg = cfg;
  out->lds_bytes = ring_bytes_of(cfg) + units * 16;
  const int tiles_x = (sp.OW + sp.TW - 1) / sp.TW, tiles_y = (sp.OH + sp.TH - 1) / sp.TH;
  out->blocks = tiles_x * tiles_y * sp.B;

  ChainDesc d;
  memset(&d, 0, sizeof(d));
  d.zeros = sp.zeros;
  d.clk = g_cvx_clk;
  d.dbg = cvx_tune_int("CVX_CHAIN_DBG", 0);
  d.TH = sp.TH;
  d.TW = sp.TW;
  d.tiles_x = tiles_x;
  d.tiles_y = tiles_y;
  d.par_base = par_base;

  struct ParCopy {
    const float* src;
    int off, n;
  };
  std::vector<ParCopy> par_copies;
  std::vector<ChainPackJob> jobs;

  // ---- passes ----
  int np = 0;
  double flops = 0;
  int first_pass_of_stage[CHAIN_MAX_PASSES + 1];
  for (int s = 0; s < sp.nstages; ++s) {
    const ChainStageSpec& st = sp.stages[s];
    first_pass_of_stage[s] = np;
    CVX_CHECK(st.cin % 8 == 0 && st.cout % 4 == 0 && (st.k == 1 || st.k == 3) && (st.stride == 1 || st.stride == 2), "chain: stage shape");
    CVX_CHECK(st.nextra == 0, "chain: concat segments are not compiled into the kernel");
    CVX_CHECK(st.cout % 8 == 0, "chain: output channels of a stage must be a multiple of 8");
    CVX_CHECK(st.cin % 16 == 0, "chain: input channels of a stage must be a multiple of 16 (one MFMA K-step = 2 chunks of one segment)");
    CVX_CHECK(((uintptr_t)st.wt % 16) == 0 && st.wt_ld % 8 == 0, "chain: weight alignment");
    const ChainPlaneSpec& pin = sp.planes[st.in_plane];
    const int nseg = st.k * st.k * (1 + st.nextra);
    const int K = nseg * st.cin;
    const int pad = st.k / 2;
    int so = 1, yo0 = 0, xo0 = 0;
    if (st.out_plane >= 0) {
      const ChainPlaneSpec& po = sp.planes[st.out_plane];
      so = po.scale;
      yo0 = po.y0;
      xo0 = po.x0;
      CVX_CHECK(st.ry0 >= 0 && st.rx0 >= 0 && st.ry0 + st.RH <= po.PH && st.rx0 + st.RW <= po.PW, "chain: output region outside its plane");
      CVX_CHECK(st.out_c0 % 4 == 0 && st.out_c0 + st.cout <= po.C, "chain: output channels outside the plane");
    }
    CVX_CHECK(pin.scale == so * st.stride, "chain: input plane resolution does not match the stage's stride");
    const int cy = (yo0 + st.ry0) * st.stride - pad - pin.y0, cx = (xo0 + st.rx0) * st.stride - pad - pin.x0;
    const int cy1 = cy + (st.RH - 1) * st.stride + st.k, cx1 = cx + (st.RW - 1) * st.stride + st.k;  // one past the last pixel read
    const Rect& vi = valid[st.in_plane];
    CVX_CHECK(cy >= vi.y0 && cx >= vi.x0 && cy1 <= vi.y1 && cx1 <= vi.x1, "chain: stage " + std::to_string(s) + " reads pixels nothing produced");
    CVX_CHECK(st.in_c0 % 8 == 0 && st.in_c0 + st.cin <= pin.C, "chain: input channels outside the plane");
    for (int r0 = 0; r0 < st.cout; r0 += RR) {
      CVX_CHECK(np < CHAIN_MAX_PASSES, "chain: too many passes");
      ChainPass& P = d.pass[np];
      P.rows = std::min(RR, st.cout - r0);
      P.K = K;
      P.nchunks = (K + KS * KSUB - 1) / (KS * KSUB);
      jobs.push_back(ChainPackJob{st.wt + (long long)r0 * st.wt_ld, nullptr, st.wt_ld, P.rows, K, RR, P.nchunks * KSUB * 2 * RR});
      P.in_base = base[st.in_plane] + st.in_c0 / 8;
      P.in_ps = ps[st.in_plane];
      P.in_W = pin.PW;
      P.P = st.cin / 8;
      P.nseg = nseg;
      P.ksz = st.k;
      for (int e = 0; e < st.nextra; ++e) {
        const int ep = st.extra_plane[e];
        const ChainPlaneSpec& pe = sp.planes[ep];
        CVX_CHECK(pe.scale == pin.scale && ps[ep] == P.in_ps && pe.PW == pin.PW && pe.y0 == pin.y0 && pe.x0 == pin.x0,
                  "chain: concat segments must live in planes of one geometry");
        CVX_CHECK(st.extra_c0[e] % 8 == 0 && st.extra_c0[e] + st.cin <= pe.C, "chain: concat segment outside its plane");
        CVX_CHECK(cy >= valid[ep].y0 && cx >= valid[ep].x0 && cy1 <= valid[ep].y1 && cx1 <= valid[ep].x1, "chain: concat segment reads pixels nothing produced");
        P.segoff[e + 1] = (base[ep] + st.extra_c0[e] / 8) - P.in_base;
      }
      P.cy = cy;
      P.cx = cx;
      P.s = st.stride;
      P.RW = st.RW;
      P.npix = st.RH * st.RW;
      P.act = st.act;
      P.img_scale = so;
      P.img_y0 = yo0 + st.ry0;
      P.img_x0 = xo0 + st.rx0;
      P.IH = sp.OH * so;
      P.IW = sp.OW * so;
      P.par_off = seg_off[s] + r0;
      P.par_off2 = seg_off[s] + st.cout + r0;
      if (st.out_plane >= 0) {
        CVX_CHECK(st.scale && st.shift, "chain: fp16 output needs the folded scale / shift");
        P.out_kind = 0;
        P.out_base = base[st.out_plane];
        P.out_ps = ps[st.out_plane];
        P.out_W = sp.planes[st.out_plane].PW;
        P.oy = st.ry0;
        P.ox = st.rx0;
        P.out_c0 = st.out_c0 + r0;
        if (st.res_plane >= 0) {
          const ChainPlaneSpec& pr = sp.planes[st.res_plane];
          CVX_CHECK(pr.scale == so, "chain: residual plane resolution");
          P.has_res = 1;
          P.res_base = base[st.res_plane];
          P.res_ps = ps[st.res_plane];
          P.res_W = pr.PW;
          P.ry = yo0 + st.ry0 - pr.y0;
          P.rx = xo0 + st.rx0 - pr.x0;
          const Rect& vr = valid[st.res_plane];
          CVX_CHECK(P.ry >= vr.y0 && P.rx >= vr.x0 && P.ry + st.RH <= vr.y1 && P.rx + st.RW <= vr.x1, "chain: residual region outside what was produced");
          P.res_c0 = st.res_c0 + r0;
          CVX_CHECK(st.res_c0 % 4 == 0 && st.res_c0 + st.cout <= pr.C, "chain: residual channels");
        }
      } else {
        CVX_CHECK(st.bias && ((uintptr_t)st.out32 % 16) == 0 && st.out32_off % 4 == 0 && st.out_ld % 4 == 0 && st.out32_c0 % 4 == 0,
                  "chain: fp32 output needs an aligned destination and a bias");
        P.out_kind = 1;
        P.out32 = st.out32;
        P.out32_off = st.out32_off;
        P.out_bstride = st.out_bstride;
        P.out_ld = st.out_ld;
        P.out_c0 = st.out32_c0 + r0;
      }
      // the pass after one with global stores must drain: stores and loads share vmcnt and do not retire in order between the kinds
      if (np > 0 && d.pass[np - 1].out_kind == 1) P.drain = 1;
      ++np;
    }
    if (st.out_plane >= 0) valid[st.out_plane] = Rect{st.ry0, st.rx0, st.ry0 + st.RH, st.rx0 + st.RW};
    flops += 2.0 * sp.B * sp.OH * sp.OW * so * so * (double)st.cout * K;
  }
  first_pass_of_stage[sp.nstages] = np;
  d.npass = np;
  for (int s = 0; s < sp.nstages; ++s) {
    const ChainStageSpec& st = sp.stages[s];
    if (st.live_params) {
      CVX_CHECK(st.out_plane < 0 || st.shift == st.scale + st.cout, "chain: live parameters need shift == scale + cout");
      CVX_CHECK(((uintptr_t)(st.out_plane >= 0 ? st.scale : st.bias) % 16) == 0, "chain: live parameter arrays must be 16-byte aligned");
    } else if (st.out_plane >= 0) {
      par_copies.push_back({st.scale, seg_off[s], st.cout});
      par_copies.push_back({st.shift, seg_off[s] + st.cout, st.cout});
    } else {
      par_copies.push_back({st.bias, seg_off[s], st.cout});
    }
  }

  // ---- loads ----
  int nl = 0;
  double bytes = 0;
  for (int l = 0; l < sp.nloads; ++l) {
    const ChainLoadSpec& ls = sp.loads[l];
    const ChainPlaneSpec& pl = sp.planes[ls.plane];
    CVX_CHECK(ls.c0 == 0 && ls.c == pl.C, "chain: a LOAD fills whole plane pixels");
    CVX_CHECK(((uintptr_t)ls.src % 16) == 0 && ls.ld % 8 == 0 && ls.bstride % 8 == 0, "chain: LOAD source alignment");
    CVX_CHECK(ls.IH == sp.OH * pl.scale && ls.IW == sp.OW * pl.scale, "chain: LOAD image size does not match the plane's resolution");
    ChainLoad& L = d.load[nl++];
    L.src = ls.src;
    L.bstride = ls.bstride;
    L.ld = ls.ld;
    L.kind = 0;
    L.IH = ls.IH;
    L.IW = ls.IW;
    L.up = ls.up;
    L.base = base[ls.plane];
    L.ps = ps[ls.plane];
    L.P = pl.C / 8;
    L.PW = pl.PW;
    L.nunits = pl.PH * pl.PW * L.ps;
    L.scale = pl.scale;
    L.y0 = pl.y0;
    L.x0 = pl.x0;
    L.per_wave = ((L.nunits + 63) / 64 + NW - 1) / NW;
    L.at_pass = ls.at_pass > 0 ? first_pass_of_stage[std::min(ls.at_pass, sp.nstages)] : 0;  // spec: stage index -> first pass of that stage
    if (L.at_pass > 0) {
      CVX_CHECK(L.at_pass < np, "chain: late load beyond the last pass");
      if (L.at_pass + 1 < np) d.pass[L.at_pass + 1].drain = 1;  // the counted waits are exact again after one full drain
    }
    bytes += 2.0 * sp.B * (ls.IH >> ls.up) * (ls.IW >> ls.up) * ls.c;
  }
  const int first_par_load = nl;
  CVX_CHECK(nl + sp.nstages <= CHAIN_MAX_LOADS, "chain: too many loads");
  for (int s = 0; s < sp.nstages; ++s) {  // one raw load per stage segment (source: the live arrays, or the plan's copy -- set below)
    ChainLoad& L = d.load[nl++];
    L.kind = 1;
    L.base = par_base + seg_off[s] / 4;
    L.nunits = (seg_floats[s] + 3) / 4;
    L.per_wave = ((L.nunits + 63) / 64 + NW - 1) / NW;
    L.at_pass = 0;
    const ChainStageSpec& st = sp.stages[s];
    L.src = st.live_params ? (const void*)(st.out_plane >= 0 ? st.scale : st.bias) : nullptr;
  }
  d.nload = nl;
  // ---- stores ----
  d.nstore = sp.nstores;
  for (int s = 0; s < sp.nstores; ++s) {
    const ChainStoreSpec& ss = sp.stores[s];
    const ChainPlaneSpec& pl = sp.planes[ss.plane];
    CVX_CHECK(ss.c0 % 8 == 0 && ss.c % 8 == 0 && ss.c0 + ss.c <= pl.C, "chain: STORE channels");
    CVX_CHECK(((uintptr_t)ss.dst % 16) == 0 && ss.ld % 8 == 0 && ss.bstride % 8 == 0, "chain: STORE destination alignment");
    const Rect& vs = valid[ss.plane];
    CVX_CHECK(ss.py0 >= vs.y0 && ss.px0 >= vs.x0 && ss.py0 + ss.RH <= vs.y1 && ss.px0 + ss.RW <= vs.x1, "chain: STORE region outside what was produced");
    ChainStore& S = d.store[s];
    S.dst = ss.dst;
    S.bstride = ss.bstride;
    S.ld = ss.ld;
    S.OH = ss.OH;
    S.OW = ss.OW;
    S.base = base[ss.plane] + ss.c0 / 8;
    S.ps = ps[ss.plane];
    S.PW = pl.PW;
    S.P = ss.c / 8;
    S.py0 = ss.py0;
    S.px0 = ss.px0;
    S.RW = ss.RW;
    S.RH = ss.RH;
    S.scale = pl.scale;
    S.y0 = pl.y0 + ss.py0;
    S.x0 = pl.x0 + ss.px0;
    bytes += 2.0 * sp.B * ss.OH * ss.OW * ss.c;
  }
  out->flops = flops;
  out->bytes = bytes;
  if (dry_run) return 0;

  // ---- device copies: [parameter table source | pack jobs | descriptor | packed weights] ----
  const size_t par_bytes = (size_t)par_floats * sizeof(float);
  const size_t jobs_off = (par_bytes + 255) & ~(size_t)255;
  const size_t desc_off = (jobs_off + jobs.size() * sizeof(ChainPackJob) + 255) & ~(size_t)255;
  size_t wpk_off = (desc_off + sizeof(ChainDesc) + 255) & ~(size_t)255;
  size_t total = wpk_off;
  for (const ChainPackJob& j : jobs) total += (size_t)j.units * 16;
  unsigned char* dev = nullptr;
  CVX_HIP(hipMalloc((void**)&dev, total));
  *d_alloc = dev;
  CVX_HIP(hipMemset(dev, 0, jobs_off));
  for (const ParCopy& c : par_copies) CVX_HIP(hipMemcpy(dev + (size_t)c.off * 4, c.src, (size_t)c.n * 4, hipMemcpyDeviceToDevice));
  int maxu = 0;
  for (size_t i = 0; i < jobs.size(); ++i) {
    jobs[i].dst = reinterpret_cast<half_t*>(dev + wpk_off);
    d.pass[i].wpk = jobs[i].dst;
    wpk_off += (size_t)jobs[i].units * 16;
    maxu = std::max(maxu, jobs[i].units);
  }
  CVX_HIP(hipMemcpy(dev + jobs_off, jobs.data(), jobs.size() * sizeof(ChainPackJob), hipMemcpyHostToDevice));
  for (int s = 0; s < sp.nstages; ++s)
    if (!d.load[first_par_load + s].src) d.load[first_par_load + s].src = dev + (size_t)seg_off[s] * 4;
  CVX_CHECK((int)jobs.size() <= CHAIN_MAX_JOBS, "chain: too many pack jobs");
  for (size_t i = 0; i < jobs.size(); ++i) out->jobs[i] = jobs[i];
  CVX_HIP(hipMemcpy(dev + desc_off, &d, sizeof(d), hipMemcpyHostToDevice));
  out->d_jobs = dev + jobs_off;
  out->njobs = (int)jobs.size();
  out->max_job_units = maxu;
  out->d_desc = reinterpret_cast<ChainDesc*>(dev + desc_off);
  return 0;
}

// =============================================================================================
// Chain specs of the fusion groups the engine uses (eval mode), and unit entry points for the parity tests
// =============================================================================================
namespace {
void plane(ChainSpec* sp, int idx, int scale, int y0, int x0, int PH, int PW, int C, int alias = -1, int alias_off = 0) {
  sp->planes[idx] = ChainPlaneSpec{scale, y0, x0, PH, PW, C, alias, alias_off};
  sp->nplanes = std::max(sp->nplanes, idx + 1);
}
ChainStageSpec stage_fp16(int in_plane, int in_c0, int cin, int k, int stride, const ChainConvArgs& c, int out_plane, int out_c0, int ry0, int rx0, int RH,
                          int RW, int res_plane = -1, int res_c0 = 0) {
  ChainStageSpec st;
  memset(&st, 0, sizeof(st));
  st.in_plane = in_plane;
  st.in_c0 = in_c0;
  st.cin = cin;
  st.k = k;
  st.stride = stride;
  st.wt = c.wt;
  st.wt_ld = c.wt_ld;
  st.cout = c.cout;
  st.scale = c.scale;
  st.shift = c.shift;
  st.bias = c.bias;
  st.act = c.act;
  st.live_params = c.live;
  st.out_plane = out_plane;
  st.out_c0 = out_c0;
  st.res_plane = res_plane;
  st.res_c0 = res_c0;
  st.ry0 = ry0;
  st.rx0 = rx0;
  st.RH = RH;
  st.RW = RW;
  return st;
}
}  // namespace

// Bottleneck (core/models/yolov8/modules.py:124-135): y = [x +] cv2(cv1(x)), both 3x3 Conv+BN+SiLU, C -> C -> C.
// Planes: X = tile + 2 halo, M = tile + 1 (cv1's output, zero outside the image), Y = tile.
int cvx_chain_spec_pair(ChainSpec* sp, const half_t* x, long long x_bs, int x_ld, int B, int H, int W, int C, const ChainConvArgs& c1,
                        const ChainConvArgs& c2, bool shortcut, half_t* out, long long out_bs, int out_ld, int TH, int TW, const half_t* zeros) {
  memset(sp, 0, sizeof(*sp));
  CVX_CHECK(c1.cout == C && c2.cout == C, "chain pair: C -> C -> C only");
  sp->B = B;
  sp->TH = TH;
  sp->TW = TW;
  sp->OH = H;
  sp->OW = W;
  sp->zeros = zeros;
  plane(sp, 0, 1, -2, -2, TH + 4, TW + 4, C);
  plane(sp, 1, 1, -1, -1, TH + 2, TW + 2, C);
  plane(sp, 2, 1, 0, 0, TH, TW, C);
  sp->loads[0] = ChainLoadSpec{0, x, x_bs, x_ld, H, W, 0, 0, 0, C};
  sp->nloads = 1;
  sp->stages[0] = stage_fp16(0, 0, C, 3, 1, c1, 1, 0, 0, 0, TH + 2, TW + 2);
  sp->stages[1] = stage_fp16(1, 0, C, 3, 1, c2, 2, 0, 0, 0, TH, TW, shortcut ? 0 : -1, 0);
  sp->nstages = 2;
  sp->stores[0] = ChainStoreSpec{2, 0, C, out, out_bs, out_ld, H, W, 0, 0, TH, TW};
  sp->nstores = 1;
  return 0;
}

// One conv (1x1 / 3x3, stride 1 / 2) + folded BN + activation as a one-stage chain: X = input patch of the tile, Y = tile.
int cvx_chain_spec_single(ChainSpec* sp, const half_t* x, long long x_bs, int x_ld, int B, int IH, int IW, int Cin, int k, int stride, int up,
                          const ChainConvArgs& c, half_t* out, long long out_bs, int out_ld, int TH, int TW, const half_t* zeros) {
  memset(sp, 0, sizeof(*sp));
  const int pad = k / 2;
  const int OH = (IH + 2 * pad - k) / stride + 1, OW = (IW + 2 * pad - k) / stride + 1;
  CVX_CHECK(OH * stride == IH && OW * stride == IW, "chain single: input size must be a multiple of the stride");
  sp->B = B;
  sp->TH = TH;
  sp->TW = TW;
  sp->OH = OH;
  sp->OW = OW;
  sp->zeros = zeros;
  plane(sp, 0, stride, -pad, -pad, (TH - 1) * stride + k, (TW - 1) * stride + k, Cin);
  plane(sp, 1, 1, 0, 0, TH, TW, c.cout);
  sp->loads[0] = ChainLoadSpec{0, x, x_bs, x_ld, IH, IW, up, 0, 0, Cin};
  sp->nloads = 1;
  sp->stages[0] = stage_fp16(0, 0, Cin, k, stride, c, 1, 0, 0, 0, TH, TW);
  sp->nstages = 1;
  sp->stores[0] = ChainStoreSpec{1, 0, c.cout, out, out_bs, out_ld, OH, OW, 0, 0, TH, TW};
  sp->nstores = 1;
  return 0;
}

// One Detect level (core/models/yolov8/modules.py:407-455, train-mode output rows): x -> A = 3x3 (cb + cc channels: box | class branch)
// -> B1 = 3x3 on A[0:cb], B2 = 3x3 on A[cb:] -> 1x1 + bias each -> fp32 rows pred[b][a_off + pixel][0:64 | 64:64+ncp].
// Planes: X = tile + 2, A = tile + 1, HB / HC = tile (inside X's space when they fit: X is dead after stage A).
int cvx_chain_spec_detect(ChainSpec* sp, const half_t* x, long long x_bs, int x_ld, int B, int H, int W, int Cin, int cb, int cc, int ncp,
                          const ChainConvArgs& a, const ChainConvArgs& b1, const ChainConvArgs& b2, const ChainConvArgs& o1, const ChainConvArgs& o2,
                          float* pred, long long pred_bs, int pred_ld, int a_off, int TH, int TW, const half_t* zeros) {
  memset(sp, 0, sizeof(*sp));
  CVX_CHECK(a.cout == cb + cc && b1.cout == cb && b2.cout == cc && o1.cout == 64 && o2.cout == ncp && cb % 8 == 0 && cc % 8 == 0, "chain detect: channel counts");
  sp->B = B;
  sp->TH = TH;
  sp->TW = TW;
  sp->OH = H;
  sp->OW = W;
  sp->zeros = zeros;
  plane(sp, 0, 1, -2, -2, TH + 4, TW + 4, Cin);
  plane(sp, 1, 1, -1, -1, TH + 2, TW + 2, cb + cc);
  auto r64 = [](int u) { return (u + 63) & ~63; };  // planes own whole 64-unit DMA pieces (cvx_chain_plan)
  const int xu = r64((TH + 4) * (TW + 4) * cvx_chain_pixel_units(Cin));
  const int hbu = r64(TH * TW * cvx_chain_pixel_units(cb)), hcu = r64(TH * TW * cvx_chain_pixel_units(cc));
  if (hbu + hcu <= xu) {
    plane(sp, 2, 1, 0, 0, TH, TW, cb, 0, 0);
    plane(sp, 3, 1, 0, 0, TH, TW, cc, 0, hbu);
  } else if (hcu <= xu) {
    plane(sp, 2, 1, 0, 0, TH, TW, cb);
    plane(sp, 3, 1, 0, 0, TH, TW, cc, 0, 0);
  } else {
    plane(sp, 2, 1, 0, 0, TH, TW, cb);
    plane(sp, 3, 1, 0, 0, TH, TW, cc);
  }
  sp->loads[0] = ChainLoadSpec{0, x, x_bs, x_ld, H, W, 0, 0, 0, Cin};
  sp->nloads = 1;
  sp->stages[0] = stage_fp16(0, 0, Cin, 3, 1, a, 1, 0, 0, 0, TH + 2, TW + 2);
  sp->stages[1] = stage_fp16(1, 0, cb, 3, 1, b1, 2, 0, 0, 0, TH, TW);
  sp->stages[2] = stage_fp16(1, cb, cc, 3, 1, b2, 3, 0, 0, 0, TH, TW);
  ChainStageSpec s3 = stage_fp16(2, 0, cb, 1, 1, o1, -1, 0, 0, 0, TH, TW);
  s3.out32 = pred;  // NULL: supplied with every launch
  s3.out32_off = (long long)a_off * pred_ld;
  s3.out_bstride = pred_bs;
  s3.out_ld = pred_ld;
  s3.out32_c0 = 0;
  ChainStageSpec s4 = stage_fp16(3, 0, cc, 1, 1, o2, -1, 0, 0, 0, TH, TW);
  s4.out32 = s3.out32;
  s4.out32_off = s3.out32_off;
  s4.out_bstride = pred_bs;
  s4.out_ld = pred_ld;
  s4.out32_c0 = 64;
  sp->stages[3] = s3;
  sp->stages[4] = s4;
  sp->nstages = 5;
  sp->nstores = 0;
  return 0;
}

namespace {
struct UnitZeros {
  half_t* p = nullptr;
  int get(const half_t** out) {
    if (!p) {
      CVX_HIP(hipMalloc((void**)&p, 256));
      CVX_HIP(hipMemset(p, 0, 256));
    }
    *out = p;
    return 0;
  }
} g_unit_zeros;

int run_spec_once(const ChainSpec& sp, hipStream_t st, int reps, float* elapsed_us) {
  ChainPlan plan;
  void* mem = nullptr;
  int rc = cvx_chain_plan(sp, &plan, &mem);
  if (rc == 0) rc = cvx_chain_pack(plan, st);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (rc == 0 && elapsed_us) {
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    rc = cvx_chain_launch(plan, st);  // warm-up (first-launch costs stay out of the figure)
    (void)hipEventRecord(e0, st);
  }
  for (int r = 0; rc == 0 && r < (reps < 1 ? 1 : reps); ++r) rc = cvx_chain_launch(plan, st);
  if (e1) (void)hipEventRecord(e1, st);
  hipError_t e = hipStreamSynchronize(st);
  if (e0 && e1) {
    float ms = 0.f;
    if (e == hipSuccess && rc == 0 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) *elapsed_us = ms * 1e3f / (reps < 1 ? 1 : reps);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  if (mem) (void)hipFree(mem);
  if (rc != 0) return rc;
  CVX_HIP(e);
  return 0;
}
}  // namespace

// ---- unit entry points (tests/test_gpu_parity.py): raw NHWC fp16 tensors, weights [cout][k][k][cin] fp16, folded scale / shift fp32 ----
extern "C" int cvx_chain_pair_unit(const void* x, int32_t batch, int32_t h, int32_t w, int32_t c, const void* w1, const float* sc1, const float* sh1,
                                   const void* w2, const float* sc2, const float* sh2, int32_t shortcut, void* out, int32_t th, int32_t tw,
                                   int32_t reps, float* elapsed_us, void* hip_stream) {
  CVX_CHECK(x && w1 && w2 && out && batch > 0 && c % 8 == 0 && th > 0 && tw > 0, "bad arguments");
  const half_t* zeros;
  CVX_TRY(g_unit_zeros.get(&zeros));
  ChainSpec sp;
  ChainConvArgs c1{(const half_t*)w1, 9 * c, c, sc1, sh1, nullptr, 0, 0}, c2{(const half_t*)w2, 9 * c, c, sc2, sh2, nullptr, 0, 0};
  CVX_TRY(cvx_chain_spec_pair(&sp, (const half_t*)x, (long long)h * w * c, c, batch, h, w, c, c1, c2, shortcut != 0, (half_t*)out, (long long)h * w * c, c, th,
                              tw, zeros));
  return run_spec_once(sp, (hipStream_t)hip_stream, reps, elapsed_us);
}

extern "C" int cvx_chain_conv_unit(const void* x, int32_t batch, int32_t ih, int32_t iw, int32_t cin, const void* wt, int32_t cout, int32_t k, int32_t stride,
                                   int32_t upsample, const float* scale, const float* shift, int32_t act, void* out, int32_t th, int32_t tw,
                                   int32_t reps, float* elapsed_us, void* hip_stream) {
  CVX_CHECK(x && wt && out && batch > 0 && cin % 8 == 0 && cout % 8 == 0 && th > 0 && tw > 0, "bad arguments");
  const half_t* zeros;
  CVX_TRY(g_unit_zeros.get(&zeros));
  ChainSpec sp;
  ChainConvArgs c{(const half_t*)wt, k * k * cin, cout, scale, shift, nullptr, act, 0};
  const int sih = upsample ? ih / 2 : ih, siw = upsample ? iw / 2 : iw;
  const int oh = ih / stride, ow = iw / stride;
  CVX_TRY(cvx_chain_spec_single(&sp, (const half_t*)x, (long long)sih * siw * cin, cin, batch, ih, iw, cin, k, stride, upsample ? 1 : 0, c, (half_t*)out,
                                (long long)oh * ow * cout, cout, th, tw, zeros));
  return run_spec_once(sp, (hipStream_t)hip_stream, reps, elapsed_us);
}

extern "C" int cvx_chain_detect_unit(const void* x, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cb, int32_t cc, int32_t ncp, const void* wa,
                                     const float* sca, const float* sha, const void* wb1, const void* wb2, const float* scb, const float* shb,
                                     const void* wo1, const void* wo2, const float* bias, float* pred, int32_t anchors, int32_t a_off, int32_t th,
                                     int32_t tw, int32_t reps, float* elapsed_us, void* hip_stream) {
  CVX_CHECK(x && wa && wb1 && wb2 && wo1 && wo2 && pred && batch > 0, "bad arguments");
  const half_t* zeros;
  CVX_TRY(g_unit_zeros.get(&zeros));
  ChainSpec sp;
  const int no = 64 + ncp;
  ChainConvArgs a{(const half_t*)wa, 9 * cin, cb + cc, sca, sha, nullptr, 0, 0};
  ChainConvArgs b1{(const half_t*)wb1, 9 * cb, cb, scb, shb, nullptr, 0, 0}, b2{(const half_t*)wb2, 9 * cc, cc, scb + cb, shb + cb, nullptr, 0, 0};
  ChainConvArgs o1{(const half_t*)wo1, cb, 64, nullptr, nullptr, bias, 2, 0}, o2{(const half_t*)wo2, cc, ncp, nullptr, nullptr, bias + 64, 2, 0};
  CVX_TRY(cvx_chain_spec_detect(&sp, (const half_t*)x, (long long)h * w * cin, cin, batch, h, w, cin, cb, cc, ncp, a, b1, b2, o1, o2, pred,
                                (long long)anchors * no, no, a_off, th, tw, zeros));
  return run_spec_once(sp, (hipStream_t)hip_stream, reps, elapsed_us);
}
