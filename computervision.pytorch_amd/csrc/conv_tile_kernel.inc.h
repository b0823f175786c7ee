// The row-band convolution kernel itself (see conv_tile.hip for the design); included by conv_tile_k1..k4.hip, which define CVX_TILE_MT_A /
// CVX_TILE_MT_B (the two pixel-group counts they instantiate) and CVX_TILE_LAUNCH_FN.
#include "conv_tile.h"

namespace {
using namespace cvx_tile_k;

template <int MT, int NTW>
__global__ __launch_bounds__(64 * kTileWaves) void conv_tile_kernel(const ConvParams p, const TileArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (and has no __amdgpu_buffer_rsrc_t)
  constexpr int BN = 16 * NTW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int fr = lane & 15, fq = lane >> 4;
  // the channel blocks of one tile run on one XCD (blocks b and b + 8 share an XCD): they read the same patch out of its L2
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int nblk = seq % a.NB;
  const int tile = (seq / a.NB) * 8 + xcd;
  if (tile >= a.ntiles) return;
  clk_mark(p, 0);
  const int b = tile / a.tiles_per_img;
  const int y0 = (tile - b * a.tiles_per_img) * a.TR;
  const int H = p.IH, W = p.IW, Wp = a.Wp, P = a.P;

  const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.in), (short)0, (int)a.in_records, 0x00020000);
  const half_t* wblk = a.wpk + (long long)nblk * a.NSTEPS * (BN * 32);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(wblk), (short)0, a.NSTEPS * BN * 64, 0x00020000);

  // ---- halo patch -> LDS: every wave issues PPW pieces (the surplus ones repeat the last piece: same bytes to the same place) ----
  {
    const unsigned img_off = (unsigned)((long long)b * p.in_bstride * 2);
    for (int k = 0; k < a.PPW; ++k) {
      int piece = k * kTileWaves + wave;
      piece = piece < a.PP ? piece : a.PP - 1;
      const unsigned u = (unsigned)(piece * 64 + lane);
      const unsigned q = __umulhi(u, a.magic_p);  // (no pow2 / non-pow2 branches anywhere: a taken scalar branch costs tens of cycles)
      const unsigned phys = u - q * (unsigned)P;
      const unsigned pr = __umulhi(q, a.magic_wp);
      const unsigned pc = q - pr * (unsigned)Wp;
      const unsigned c = phys ^ ((pc >> a.sh) & (unsigned)a.swmask);  // swmask = 0 when P is not a power of two
      const int y = y0 - 1 + (int)pr, x = (int)pc - 1;
      const bool ok = u < (unsigned)a.units && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
      const unsigned vo = ok ? img_off + (unsigned)(((y * W + x) * p.in_ld + (int)c * 8) * 2) : 0xffffffffu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (lds_ptr_t)(smem + piece * 1024), 16, vo, 0, 0, 0);
    }
  }
  // ---- weight chunks: contiguous, pre-packed ----
  const unsigned vb = (unsigned)(lane * 16);
  auto issue_chunk = [&](int c, int slot) __attribute__((always_inline)) {
    unsigned char* sb = smem + a.wring_off + slot * a.CB;
    const int base = c * a.CB;
    for (int k = 0; k < a.PW; ++k) {
      int piece = k * kTileWaves + wave;
      piece = piece < a.CP ? piece : a.CP - 1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(sb + piece * 1024), 16, vb, base + piece * 1024, 0, 0);
    }
  };
  // ring of R slots: chunks 0 .. R - 2 go out now (all of them when every chunk has a slot of its own: the DMA latency of a chunk is
  // 1-2 us with every CU loading at once -- longer than the whole K loop of a small layer, so whatever fits is requested up front)
  const int npre = a.NCH < a.R - 1 ? a.NCH : a.R - 1;
  for (int c = 0; c < npre; ++c) issue_chunk(c, c);
  clk_mark(p, 1);

  // ---- per-lane fragment addresses (bytes).  Pixel m = (w * MT + i) * 16 + fr of the tile, row-major; patch pixel (pr + 1 + dh,
  // pc + 1 + dw); unit (row * Wp + col) * P + (chunk ^ f(col)).  S[dwi][i]: the column part for dw = dwi - 1 plus the row part for
  // dh = 0, chunk = fq; a K-step adds dh * rowpitch and XORs (s << 6) (chunk = 4 s + fq; power-of-two P) or adds the chunk offset. ----
  const int rows_here = H - y0 < a.TR ? H - y0 : a.TR;
  const int npix = rows_here * W;
  unsigned S0[MT], S1[MT], S2[MT];  // (three NAMED arrays: a [3][MT] array indexed by the tap's column offset went to scratch memory)
  bool pvalid[MT];
  const unsigned rowpitch = (unsigned)(Wp * P * 16);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = (wave * MT + i) * 16 + fr;
    pvalid[i] = m < npix;
    const unsigned mm = pvalid[i] ? (unsigned)m : 0u;
    const unsigned pr = __umulhi(mm, a.magic_w);
    const unsigned pc = mm - pr * (unsigned)W;
    auto s_of = [&](unsigned col) -> unsigned {  // col = pc + 1 + dw
      const unsigned sw = (col >> a.sh) & (unsigned)a.swmask;
      const unsigned ph = ((unsigned)fq ^ sw) & (unsigned)a.phmask;
      return (pr + 1u) * rowpitch + ((col * (unsigned)P + ph) << 4);
    };
    S0[i] = s_of(pc);
    S1[i] = s_of(pc + 1u);
    S2[i] = s_of(pc + 2u);
  }

  f4 acc[MT][NTW];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  // K-step addressing: xk (XOR, power-of-two P) / ak (ADD: tap row offset; non-power-of-two P: the chunk offset, clamped to the pixel's
  // last chunk where the step's K range passes Cin -- the packed weights are zero there) of step (tap code, s)
  const unsigned long long pos_pack = p.halo_pos;
  const unsigned xk_mask = a.pow2 ? 0xffffffffu : 0u, np2_mask = ~xk_mask;
  auto step_xk = [&](int s) -> unsigned { return (unsigned)(s << 6) & xk_mask; };
  auto step_ak = [&](int dh, int s) -> unsigned {
    int c = 4 * s + fq;
    c = c < P ? c : P - 1;
    return (unsigned)(dh * (int)rowpitch) + ((unsigned)((c - fq) << 4) & np2_mask);  // dh in {-1, 0, 1}: wraps as intended
  };

  // everything above is needed only after the wait below: keep it ABOVE the wait (the compiler sank these ~300 instructions behind the
  // barrier, where nothing overlapped them: 2.8 us of a 6.7-us "K loop" was this address arithmetic)
#pragma unroll
  for (int i = 0; i < MT; ++i) asm volatile("" ::"v"(S0[i]), "v"(S1[i]), "v"(S2[i]));
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) asm volatile("" ::"v"(acc[i][j]));
  // ---- wait: patch + chunk 0 (the other requested chunks may stay in flight) ----
  if (npre <= 1 || (a.dbg & 1)) wait_vmcnt<0>();
  else wait_vmcnt_dyn((npre - 1) * a.PW);
  workgroup_barrier();
  clk_mark(p, 2);
#ifdef CVX_TUNING
  if ((a.dbg & 8) && p.clk) {  // dump the LDS image (patch | ring) of block 0 after the first barrier (and after a full drain) and stop
    wait_vmcnt<0>();
    workgroup_barrier();
    if (blockIdx.x == 0)
      for (int o = tid * 4; o < a.stat_off; o += 1024) reinterpret_cast<unsigned*>(p.clk)[o >> 2] = *reinterpret_cast<const unsigned*>(smem + o);
    return;
  }
#endif

  // ---- K loop: ROLLED, three K-steps per trip (NSTEPS = 9 * SPT is a multiple of 3), ~3 KB of code: these launches are too short to
  // stream 30 KB of unrolled instructions through a cold instruction cache.  THREE fragment sets rotate: the body of step n requests the
  // fragments of step n + 2 (weights first, the pixel fragment of group i right behind the MFMAs of group i: address arithmetic and LDS
  // issue ride in the MFMA shadows -- one wave per SIMD, in order), runs the MT * NTW MFMAs of step n, and waits only for step n + 1's
  // fragments (requested a whole step earlier: LDS returns in order, so lgkmcnt(MT + NTW) leaves this step's requests in flight).  With
  // one step of look-ahead and lgkmcnt(0) every step, one LDS latency (~250 cycles with four waves' requests queued) stood exposed
  // against 224 cycles of MFMA per step.  Inline-asm ds_read: compiler-visible LDS loads would each wait for the ring's in-flight
  // LDS-DMA (vmcnt), which they may alias.  Every trip ends with lgkmcnt(0): nothing is in flight across the back-edge, where the
  // compiler may copy fragment registers (it did: stale operands). ----
  h8 xs[3][MT], ws[3][NTW];
  unsigned SC[MT];  // S row of the tap the load cursor is in
  const unsigned lds0 = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)smem;
  const int NSTEPS = a.NSTEPS, SPT = a.SPT, KSC = a.KSC, NCH = a.NCH, CB = a.CB, R = a.R, PWc = a.PW;
  const unsigned wfrag0 = lds0 + (unsigned)a.wring_off + (unsigned)(lds_row_off(fr, fq) * 2);
  // load cursor: the step whose fragments are requested next
  int c_tap = 0, c_st = 0, c_inc = 0, c_chunk = 0, c_slot = 0, c_n = 0;
  int dh_cur;
  unsigned xk_c, ak_c, w_c;
  auto set_tap = [&](int t) __attribute__((always_inline)) -> int {  // SC <- S[dw of tap t]; returns dh
    const int code = (int)(pos_pack >> (4 * t)) & 15;
    const int dwi = code & 3;
    // masks, not selects: hipcc turned the select chain into ~4 scalar branches per element, ~1,500 cycles per tap
    const unsigned m0 = dwi == 0 ? 0xffffffffu : 0u, m1 = dwi == 1 ? 0xffffffffu : 0u, m2 = dwi == 2 ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < MT; ++i) SC[i] = (S0[i] & m0) | (S1[i] & m1) | (S2[i] & m2);
    return (code >> 2) - 1;
  };
  dh_cur = set_tap(0);
  xk_c = step_xk(0);
  ak_c = step_ak(dh_cur, 0) + lds0;
  w_c = wfrag0;
  // moves the cursor to the next step (stays on the last one: the requests past the end re-read valid fragments, unused)
  auto advance = [&]() __attribute__((always_inline)) {
    if (c_n + 1 < NSTEPS) {
      ++c_n;
      if (++c_st == SPT) {
        c_st = 0;
        ++c_tap;
        dh_cur = set_tap(c_tap);
      }
      xk_c = step_xk(c_st);
      ak_c = step_ak(dh_cur, c_st) + lds0;
      if (++c_inc == KSC) {  // the cursor enters chunk + 1: published here; chunk - 1 is retired, its slot takes chunk + R - 1
        c_inc = 0;
        const int issued = NCH < c_chunk + R - 1 ? NCH : c_chunk + R - 1;
        if (issued - c_chunk - 2 <= 0) wait_vmcnt<0>();
        else wait_vmcnt_dyn((issued - c_chunk - 2) * PWc);  // the chunks requested behind chunk + 1 may stay in flight
        workgroup_barrier();
        if (c_chunk + R - 1 < NCH) issue_chunk(c_chunk + R - 1, c_slot == 0 ? R - 1 : c_slot - 1);
        ++c_chunk;
        c_slot = c_slot == R - 1 ? 0 : c_slot + 1;
      }
      w_c = wfrag0 + (unsigned)(c_slot * CB + c_inc * (NTW * 1024));
    }
  };
  // fragments of steps 0 and 1
  frag_load_w<NTW>(ws[0], w_c);
#pragma unroll
  for (int i = 0; i < MT; ++i) xs[0][i] = lds_frag<0>((SC[i] ^ xk_c) + ak_c);
  advance();
  frag_load_w<NTW>(ws[1], w_c);
#pragma unroll
  for (int i = 0; i < MT; ++i) xs[1][i] = lds_frag<0>((SC[i] ^ xk_c) + ak_c);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  constexpr int LW = (MT + NTW) > 15 ? 15 : (MT + NTW);
#if defined(CVX_TILE_ABL)  // compile-time ablation (one-off experiment builds): 64 no pixel-fragment reads, 128 no weight-fragment reads; results are WRONG
#define CVX_TILE_DBG_BIT(b) (((CVX_TILE_ABL) & (b)) != 0)
#else
#define CVX_TILE_DBG_BIT(b) false
#endif
  // body K: compute step n from set K, request step n + 2 into set (K + 2) % 3
#define CVX_TILE_BODY(K, LAST)                                                                                          \
  {                                                                                                                     \
    advance();                                                                                                          \
    if (!CVX_TILE_DBG_BIT(128)) frag_load_w<NTW>(ws[((K) + 2) % 3], w_c);                                                \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                                \
      _Pragma("unroll") for (int j = 0; j < NTW; ++j)                                                                   \
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ws[K][j], xs[K][i], acc[i][j], 0, 0, 0);                   \
      if (!CVX_TILE_DBG_BIT(64)) xs[((K) + 2) % 3][i] = lds_frag<0>((SC[i] ^ xk_c) + ak_c);                              \
    }                                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    if (LAST) {                                                                                                         \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                \
    } else {                                                                                                            \
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(LW) : "memory");                                                       \
    }                                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
  }
  clk_mark(p, 7);
  for (int n = 0; n < NSTEPS; n += 3) {
    CVX_TILE_BODY(0, false)
    CVX_TILE_BODY(1, false)
    CVX_TILE_BODY(2, true)
  }
#undef CVX_TILE_BODY
  clk_mark(p, 3);

  // ---- epilogue.  The accumulators (lane = pixel fr of a group, 4 channels per 16-channel tile) pass through the now idle LDS, wave by
  // wave (a private region: no barrier but the one below), and come back as ROWS: a lane takes 8 consecutive channels of one pixel, so
  // every output element leaves in a 16-byte store whose neighbours in the wave are contiguous (straight from the accumulators a
  // lane stored 8 bytes per pixel pitch: 16 cache lines per instruction), and the activation / residual / accumulate arithmetic is ONE
  // rolled loop (the shared epilogue's kinds sat inside MT x NTW unrolled bodies: 3-6 us of mostly instruction fetch per launch). ----
  constexpr int RS = BN * 4 + 16;                  // bytes per staged pixel row (fp32 channels + 16: conflict-free 16-byte columns)
  constexpr int CPR = BN / 8;                      // 8-channel chunks per row
  constexpr int RPI = 64 / CPR;                    // rows per loop trip; lanes >= RPI * CPR idle (channel counts that do not divide 64)
  constexpr int ROWS = MT * 16;                    // pixel rows of this wave
  // training: per-channel sums straight from the accumulators (one lane group = 16 pixels of 4 channels), folded below
  const int epi = p.epi;
  f4 st1[NTW], st2[NTW];
  if (epi == CVX_EPI_RAW_STATS) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      st1[j] = st2[j] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < MT; ++i)
        if (pvalid[i]) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            st1[j][r] += acc[i][j][r];
            st2[j][r] += acc[i][j][r] * acc[i][j][r];
          }
        }
    }
  }
  if (epi == CVX_EPI_RAW_STATS) {
    // training: raw fp32 rows straight from the accumulators (16-byte stores, 64-byte runs per pixel: measured ahead of the staged
    // form here -- 2.8 vs 4.2 us on 40 x 40, 64 -> 64 -- the fp32 rows are twice the bytes to stage)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (!pvalid[i]) continue;
      const unsigned m = (unsigned)((wave * MT + i) * 16 + fr);
      const unsigned pr = __umulhi(m, a.magic_w);
      const unsigned pc = m - pr * (unsigned)W;
      const long long dst = (long long)b * p.out_bstride + ((long long)(y0 + (int)pr) * W + (int)pc) * p.out_ld + nblk * BN + fq * 4;
#pragma unroll
      for (int j = 0; j < NTW; ++j)
        if (nblk * BN + j * 16 + fq * 4 < p.Cout) cvx_store_raw4(p, dst + j * 16, acc[i][j]);
    }
  } else {
  __syncthreads();  // every wave is done with the patch and the ring
    unsigned char* wreg = smem + wave * (ROWS * RS);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) *reinterpret_cast<f4*>(wreg + (i * 16 + fr) * RS + (j * 16 + fq * 4) * 4) = acc[i][j];
    {
      const int q = lane % CPR, r0 = lane / CPR;
      const int n = nblk * BN + q * 8;            // first of this lane's 8 channels
      const bool lane_on = r0 < RPI && n < p.Cout;
      const int act_kind = p.act_kind, res_pre = p.res_pre, accumulate = p.accumulate;
      const half_t* res = p.res;
      float c0[8], c1[8];                         // scale / shift (AFFINE_SILU), bias (BIAS_F32)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        c0[k] = 1.f;
        c1[k] = 0.f;
      }
      if (lane_on && epi == CVX_EPI_AFFINE_SILU) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          c0[k] = p.scale[n + k];
          c1[k] = p.shift[n + k];
        }
      }
      if (lane_on && epi == CVX_EPI_BIAS_F32) {
#pragma unroll
        for (int k = 0; k < 8; ++k) c1[k] = p.bias[n + k];
      }
      const int m_wave = wave * ROWS;
      for (int r = r0; r < ROWS; r += RPI) {
        const int m = m_wave + r;
        if (!lane_on || m >= npix) continue;
        const unsigned pr = __umulhi((unsigned)m, a.magic_w);
        const unsigned pc = (unsigned)m - pr * (unsigned)W;
        const long long pix = (long long)(y0 + (int)pr) * W + (int)pc;
        const f4 lo = *reinterpret_cast<const f4*>(wreg + r * RS + q * 32);
        const f4 hi = *reinterpret_cast<const f4*>(wreg + r * RS + q * 32 + 16);
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (epi == CVX_EPI_BIAS_F32) {
          float* dst = p.out32 + (long long)b * p.out_bstride + pix * p.out_ld + n;
          *reinterpret_cast<f4*>(dst) = f4{v[0] + c1[0], v[1] + c1[1], v[2] + c1[2], v[3] + c1[3]};
          *reinterpret_cast<f4*>(dst + 4) = f4{v[4] + c1[4], v[5] + c1[5], v[6] + c1[6], v[7] + c1[7]};
          continue;
        }
        half_t* dst = p.out16 + (long long)b * p.out_bstride + pix * p.out_ld + n;
        if (epi == CVX_EPI_AFFINE_SILU) {
          float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (res) {
            const h8 rr = *reinterpret_cast<const h8*>(res + (long long)b * p.res_bstride + pix * p.res_ld + n);
#pragma unroll
            for (int k = 0; k < 8; ++k) rv[k] = (float)rr[k];
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float t = v[k] * c0[k] + c1[k];
            if (res_pre) t += rv[k];
            if (act_kind == 0) t = cvx_silu(t);
            else if (act_kind == 1) t = fmaxf(t, 0.f);
            if (!res_pre) t += rv[k];
            v[k] = t;
          }
        } else if (accumulate) {  // PLAIN: data gradients that add to what another consumer left there
          const h8 old = *reinterpret_cast<const h8*>(dst);
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += (float)old[k];
        }
        *reinterpret_cast<h8*>(dst) = h8{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3], (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
      }
    }
  }
  if (epi == CVX_EPI_RAW_STATS) {
    __syncthreads();  // the staged rows have been read: the scratch below may overlap them
    float* sStat = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = cvx_wave_sum16(st1[j][r]), s2 = cvx_wave_sum16(st2[j][r]);
        if (fr == 0) {
          sStat[(wave * BN + j * 16 + fq * 4 + r) * 2 + 0] = s1;
          sStat[(wave * BN + j * 16 + fq * 4 + r) * 2 + 1] = s2;
        }
      }
    __syncthreads();
    for (int t = tid; t < BN * 2; t += 64 * kTileWaves) {
      const int ch = t >> 1, which = t & 1, nn = nblk * BN + ch;
      if (nn < p.Cout) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kTileWaves; ++w) v += sStat[(w * BN + ch) * 2 + which];
        cvx_fix_atomic_add(p.stats, ((long long)(blockIdx.x % p.stats_replicas) * p.Cout + nn) * 2 + which, v);
      }
    }
  }
  clk_mark(p, 4);
#endif
}


template <int MT, int NTW>
int launch_tile(const ConvParams& p, const TileArgs& a, int lds, hipStream_t stream) {
  static unsigned long long optin_mask = 0;
  CVX_TRY(cvx_lds_optin((const void*)conv_tile_kernel<MT, NTW>, 160 * 1024, &optin_mask));
  const int blocks = ((a.ntiles + 7) / 8) * 8 * a.NB;
  hipLaunchKernelGGL((conv_tile_kernel<MT, NTW>), dim3(blocks), dim3(64 * kTileWaves), lds, stream, p, a);
  return 0;
}
template <int MT>
int launch_tile_n(int NTW, const ConvParams& p, const TileArgs& a, int lds, hipStream_t st) {
  switch (NTW) {
    case 1: return launch_tile<MT, 1>(p, a, lds, st);
    case 2: return launch_tile<MT, 2>(p, a, lds, st);
    case 3: return launch_tile<MT, 3>(p, a, lds, st);
    case 4: return launch_tile<MT, 4>(p, a, lds, st);
    case 5: return launch_tile<MT, 5>(p, a, lds, st);
    case 6: return launch_tile<MT, 6>(p, a, lds, st);
    default: return launch_tile<MT, 9>(p, a, lds, st);
  }
}
}  // namespace

int CVX_TILE_LAUNCH_FN(int MT, int NTW, const ConvParams& p, const cvx_tile_k::TileArgs& a, int lds, hipStream_t st) {
  if (MT == CVX_TILE_MT_A) return launch_tile_n<CVX_TILE_MT_A>(NTW, p, a, lds, st);
  return launch_tile_n<CVX_TILE_MT_B>(NTW, p, a, lds, st);
}
