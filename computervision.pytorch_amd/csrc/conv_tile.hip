// 3x3 / stride-1 convolution for the SMALL maps (<= 80 x 80 at batch 32: the launches whose cost is their fixed per-launch floor, not
// their bytes or FLOPs) -- round 4.  One full-width row band of one image per workgroup, one round of workgroups on the 256 CUs:
//
//   * tile = TR full rows x W columns of one image; the pixels of the tile are enumerated row-major and cut into groups of 16 (the MFMA
//     pixel dimension), so a 40- or 20-wide map wastes nothing on 16-column tiles (conv_halo.hip: 48 / 32 columns computed for 40 / 20);
//     4 waves, wave w owns groups [w * MT, (w + 1) * MT) and ALL 16 * NTW output channels of the workgroup; the launcher picks
//     (TR, channel block) so that tiles x channel blocks fills the chip in ONE round where the layer is small enough (cvx_conv_tile_plan);
//   * the (TR + 2) x (W + 2) x Cin halo patch is DMA'd into LDS once (`buffer_load ... lds`, hardware zero fill outside the image);
//     16-byte units [pixel][Cin / 8], the unit index XOR-swizzled by the patch COLUMN (applied on the global side) so that the 16 pixels
//     of a fragment read fall on 16 different 16-byte slots;
//   * the weights arrive PRE-PACKED in LDS image order ([channel block][K-step][16 * NTW rows][32 k], cvx_conv_tile_pack_jobs: one
//     launch per forward for all layers), so a K-chunk is ONE contiguous block: a wave issues 1-KiB pieces at base + lane * 16 with a
//     scalar offset -- no per-lane address arithmetic (conv_halo.hip spent 1.7 us of a 12-us launch issuing its weight DMAs);
//     they stream through a 3-slot ring of chunks (chunk c + 2 is issued when chunk c's last K-step begins), so the first MFMA waits
//     for the patch and ONE chunk, and layers whose weights exceed the LDS (128 -> 144 at 40 x 40: 332 KB) need no channel split that
//     re-reads the patch;
//   * K loop: ROLLED, one K-step (tap, 32 channels) per iteration, everything but (MT, NTW) at run time (any Cin % 8 == 0, >= 32);
//     v_mfma_f32_16x16x32_f16, weights as the A operand; ROLLING fragment refill: the pixel fragment of group i is re-requested for the
//     NEXT K-step right behind the MFMAs that consumed it, the next step's weight fragments at the top of the step, so every LDS latency
//     hides behind MT * NTW MFMAs of the wave itself (one wave per SIMD: nobody else would hide it);
//   * epilogues: the shared ones of conv_tile_common.h (raw fp32 + statistics / folded BN + activation / bias / plain-accumulate).
// Roofline: latency (dispatch, DMA landing, K loop, stores); measured phases: profiles/r04_conv_floor.txt.  Reference: the arithmetic of
// nn.Conv2d in core/models/yolov8/modules.py:19-33 (Conv), :124-135 (Bottleneck), :407-455 (Detect) and its autograd data gradient.
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "conv_tile.h"

namespace {
using namespace cvx_tile_k;

// [rows][taps * Cin] fp16 (row pitch src_ld, tap t at wtap[t] * Cin) -> [channel block][K-step][BN rows][32 k] in LDS image order
// (lds_row_off); K-step n = tap * SPT + s covers channels [32 s, 32 s + 32) of the tap, zero beyond Cin and beyond the last row.
struct TilePackJob {
  const half_t* src;
  half_t* dst;
  unsigned long long wt_pack;  // weight tap index of K-order tap t: 4 bits each (cvx_halo_pack_taps)
  int src_ld, rows, Cin, BN, NB, SPT;
  int blk0, nblk;
};
constexpr int TPACK_UNITS = 256 * 8;
__device__ __forceinline__ void tile_pack_units(const TilePackJob& a, long long u0, long long u1) {
  const int nsteps = 9 * a.SPT;
  for (long long u = u0 + threadIdx.x; u < u1; u += blockDim.x) {
    const int sp = (int)(u & 3);
    const long long r0 = u >> 2;
    const int r = (int)(r0 % a.BN);
    const long long r1 = r0 / a.BN;
    const int n = (int)(r1 % nsteps);
    const int nb = (int)(r1 / nsteps);
    const int g = sp ^ ((r >> 1) & 3);
    const int tap = n / a.SPT, s = n - tap * a.SPT;
    const int k = s * 32 + g * 8;
    const int row = nb * a.BN + r;
    h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < a.rows && k < a.Cin) v = *reinterpret_cast<const h8*>(a.src + (long long)row * a.src_ld + (int)((a.wt_pack >> (4 * tap)) & 15) * a.Cin + k);
    *reinterpret_cast<h8*>(a.dst + u * 8) = v;
  }
}
__global__ __launch_bounds__(256) void tile_pack_kernel(const TilePackJob a) {
  const long long total = (long long)a.NB * 9 * a.SPT * a.BN * 4;
  const long long u0 = (long long)blockIdx.x * TPACK_UNITS;
  if (u0 < total) tile_pack_units(a, u0, u0 + TPACK_UNITS < total ? u0 + TPACK_UNITS : total);
}
__global__ __launch_bounds__(256) void tile_pack_jobs_kernel(const TilePackJob* jobs, int njobs) {
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].blk0 <= (int)blockIdx.x)
      lo = mid;
    else
      hi = mid - 1;
  }
  const TilePackJob a = jobs[lo];
  const long long total = (long long)a.NB * 9 * a.SPT * a.BN * 4;
  const long long u0 = (long long)((int)blockIdx.x - a.blk0) * TPACK_UNITS;
  if (u0 < total) tile_pack_units(a, u0, u0 + TPACK_UNITS < total ? u0 + TPACK_UNITS : total);
}

// ---- host side: tiling choice, packed-weight cache of the stand-alone entry points, launch ----
constexpr int kMaxMT = 4;
const int kNtwMenu[] = {1, 2, 3, 4, 5, 6, 9};  // (8 channel tiles: the compiler's register allocation collapses -- 256 VGPRs + spills at any MT; 128 channels go out as 2 x 4)

unsigned magic_of(int d) { return (unsigned)((0x100000000ULL + (unsigned long long)d - 1) / (unsigned long long)d); }  // d >= 2: floor(q / d) = umulhi(q, magic) for q * d < 2^32

struct Plan {
  int MT, NTW, TR, NB, KSC, NCH, R, SPT, lds, ntiles, tpi, PP, units, CB, wring_off, stat_off;
  double cost;
};

// estimated launch time (us) of one candidate; calibrated on the phase stamps of tools/conv_clock.py (profiles/r04_conv_floor.txt)
bool plan_tile(const ConvParams& p, Plan* best) {
  const int H = p.IH, W = p.IW, Cin = p.Cin;
  const int P = Cin / 8, SPT = (Cin + 31) / 32, NSTEPS = 9 * SPT;
  const int ctiles = (p.Cout + 15) / 16;
  static const int force_tr = cvx_tune_int("CVX_TILE_TR", 0), force_ntw = cvx_tune_int("CVX_TILE_NTW", 0);
  bool found = false;
  for (int NTW : kNtwMenu) {
    if (force_ntw && NTW != force_ntw) continue;
    if (NTW > ctiles && NTW != kNtwMenu[0]) {
      bool smaller_fits = false;
      for (int q : kNtwMenu) smaller_fits |= q >= ctiles && q < NTW;
      if (smaller_fits) continue;  // a smaller block already holds every channel
    }
    const int NB = (ctiles + NTW - 1) / NTW;
    if ((NB - 1) * NTW >= ctiles) continue;
    for (int TR = 1; TR <= H; ++TR) {
      if (force_tr && TR != force_tr) continue;
      const int G = (TR * W + 15) / 16;
      const int MT = (G + kTileWaves - 1) / kTileWaves;
      if (MT > kMaxMT) break;
      if (MT > 4 || MT * NTW * 4 + 3 * (MT + NTW) * 4 > 216) continue;  // two waves per SIMD: 256 registers per wave (accumulators + three fragment sets + ~40)
      const int tpi = (H + TR - 1) / TR;
      const int units = (TR + 2) * (W + 2) * P;
      const int PP = (units + 63) / 64;
      const int patch_bytes = PP * 1024;
      const int stat_bytes = kTileWaves * 16 * NTW * 2 * 4;
      const int budget = 160 * 1024 - patch_bytes - stat_bytes;
      // weights: ONE chunk (everything requested up front, one drain before the K loop: it lands within 0.3 us of the last request,
      // and the loop then runs without a barrier) where they fit beside the patch; else a ring of 4 (3) slots of whole taps / the largest
      // divisor of a tap's steps that fits, chunk c + 3 (c + 2) requested when chunk c's last K-step begins
      int KSC = 0, R = 0;
      if (NSTEPS * NTW * 1024 <= budget && (NSTEPS * NTW + kTileWaves - 1) / kTileWaves + (PP + kTileWaves - 1) / kTileWaves <= 60) {
        KSC = NSTEPS;
        R = 2;
      } else {
        for (int k = SPT; k >= 1 && !KSC; --k) {
          if (SPT % k) continue;
          const int r = budget / (k * NTW * 1024);
          if (r >= 3 && (k * NTW + kTileWaves - 1) / kTileWaves <= 20) {
            KSC = k;
            R = r > 4 ? 4 : r;
          }
        }
      }
      if (!KSC) continue;
      const int CB = KSC * NTW * 1024;
      const int nslots = R > NSTEPS / KSC ? NSTEPS / KSC : R;
      const int stage_bytes = kTileWaves * MT * 16 * (16 * NTW * 4 + 16);  // the epilogue's staged rows (reuses the whole allocation)
      const int lds = std::max(patch_bytes + nslots * CB + stat_bytes, stage_bytes);
      if (lds > 160 * 1024) continue;
      const long long wgs = (long long)tpi * p.B * NB;
      const double rounds = (double)((wgs + 255) / 256);
      // per workgroup: fixed 2.0 us + DMA of (patch + first chunk) at 60 GB/s before the first MFMA + K loop + stores
      const double mfma_cyc = (double)NSTEPS * MT * NTW * 16.0 * (kTileWaves / 4);  // per SIMD
      const double lds_cyc = (double)NSTEPS * kTileWaves * (MT + NTW) * 6.0;  // every wave: (MT + NTW) ds_read_b128 of ~6 LDS cycles (measured)
      const double valu_cyc = (double)NSTEPS * (MT * 3 + NTW * 4 + 8) * 4.0;
      const double loop_us = std::max(std::max(mfma_cyc * 1.15, lds_cyc), valu_cyc + mfma_cyc * 0.5) / 1900.0;
      const double wbytes = (double)NSTEPS * NTW * 1024.0;
      const double load_us = (patch_bytes + CB) / 60e3 + std::max(0.0, (wbytes - CB) / 60e3 - loop_us);
      const double epi_us = 0.6 + MT * NTW * 0.02;
      const double t = rounds * (1.2 + load_us + loop_us + epi_us) + 1.0;
      if (!found || t < best->cost) {
        found = true;
        *best = Plan{MT, NTW, TR, NB, KSC, NSTEPS / KSC, R, SPT, lds, tpi * p.B, tpi, PP, units, CB, patch_bytes, patch_bytes + nslots * CB, t};
      }
    }
  }
  return found;
}

struct PackSlot {
  half_t* buf = nullptr;
  size_t bytes = 0;
};
std::mutex g_pack_mu;
std::map<std::pair<const void*, unsigned long long>, PackSlot> g_pack;  // stand-alone launches only (unit tests, cvx_conv2d_nhwc)

}  // namespace

bool cvx_conv_tile_shape_ok(const ConvParams& p) {
  if (!p.zeros || !p.halo_taps_ok || p.ntaps != 9) return false;
  if (p.IS != 1 || p.OS != 1 || p.oph != 0 || p.opw != 0) return false;
  if (p.OH2 != p.IH || p.OW2 != p.IW || p.OWr != p.IW) return false;
  if (p.Cin % 8 != 0 || p.Cin < 32 || p.Cin > 512 || p.in_ld % 8 != 0) return false;
  // the staged epilogue hands a lane 8 consecutive channels of a pixel (8 scale / shift / bias reads, one 16-byte store): output channel
  // counts and pitches that are multiples of 4 only (36, 68: legal for cvx_conv_igemm_launch) stay on the kernels with 4-channel epilogues
  if (p.Cout % 8 != 0 || p.out_ld % 8 != 0 || (p.res && p.res_ld % 8 != 0)) return false;
  if (p.IW > 4000 || p.IH > 4000 || p.IW < 2) return false;  // (a 1-wide map: the magic-number division needs a divisor >= 2)
  const long long extent = ((long long)(p.B - 1) * p.in_bstride + ((long long)p.IH * p.IW - 1) * p.in_ld + p.Cin) * 2;
  if (extent >= (1LL << 32) - 65536) return false;  // 32-bit buffer offsets
  Plan pl;
  return plan_tile(p, &pl);
}

// where the dispatcher prefers this kernel: maps of at most 80 x 80 (larger ones are bandwidth-bound: conv_halo.hip's persistent
// double-buffered tiles), at most ~8 K pixels per image
bool cvx_conv_tile_supported(const ConvParams& p) {
  static const bool off = cvx_tune_set("CVX_NO_TILE");
  if (off || p.no_tile) return false;
  // measured against conv_halo / conv_gemm / the DMA-ring kernel (profiles/r04_conv_floor.txt): ahead on maps up to 40 x 40 with up to 144
  // gathered channels; 80 x 80 needs two rounds of workgroups (one workgroup per CU) and loses to conv_halo's persistent tiles
  static const int max_w = cvx_tune_int("CVX_TILE_MAXW", 40), max_c = cvx_tune_int("CVX_TILE_MAXC", 144);
  if (p.IW > max_w || p.IH > max_w || p.Cin > max_c) return false;
  return cvx_conv_tile_shape_ok(p);
}

bool cvx_conv_tile_plan(const ConvParams& p, TilePackPlan* out) {
  if (!cvx_conv_tile_shape_ok(p)) return false;
  Plan pl;
  if (!plan_tile(p, &pl)) return false;
  out->BN = 16 * pl.NTW;
  out->NB = pl.NB;
  out->SPT = pl.SPT;
  out->bytes = (size_t)pl.NB * 9 * pl.SPT * pl.NTW * 1024;
  out->cost_us = pl.cost;
  return true;
}

int cvx_conv_tile_pack_jobs(const void* d_jobs, int njobs, int nblocks, hipStream_t stream) {
  if (njobs <= 0 || nblocks <= 0) return 0;
  hipLaunchKernelGGL(tile_pack_jobs_kernel, dim3(nblocks), dim3(256), 0, stream, (const TilePackJob*)d_jobs, njobs);
  CVX_HIP(hipGetLastError());
  return 0;
}

// fills one job of the batched pack launch (the engine uploads the array); returns the number of pack blocks
int cvx_conv_tile_fill_job(const ConvParams& p, const TilePackPlan& tp, half_t* dst, int blk0, void* job_out) {
  TilePackJob j;
  memset(&j, 0, sizeof(j));
  j.src = p.wt;
  j.dst = dst;
  j.wt_pack = p.halo_wt;
  j.src_ld = p.wt_ld;
  j.rows = p.Cout;
  j.Cin = p.Cin;
  j.BN = tp.BN;
  j.NB = tp.NB;
  j.SPT = tp.SPT;
  j.blk0 = blk0;
  const long long units = (long long)tp.NB * 9 * tp.SPT * tp.BN * 4;
  j.nblk = (int)((units + TPACK_UNITS - 1) / TPACK_UNITS);
  memcpy(job_out, &j, sizeof(j));
  return j.nblk;
}
size_t cvx_conv_tile_job_bytes() { return sizeof(TilePackJob); }

int cvx_conv_tile_launch(const ConvParams& p, hipStream_t stream) {
  Plan pl;
  CVX_CHECK(plan_tile(p, &pl), "conv_tile: no tiling for this shape");
  const half_t* packed = (p.tile_packed && p.tile_packed_bn == 16 * pl.NTW) ? p.tile_packed : nullptr;
  if (!packed) {  // stand-alone launch: pack here (cached buffer, re-packed at every launch: the weights may have changed)
    TilePackPlan tp{16 * pl.NTW, pl.NB, pl.SPT, (size_t)pl.NB * 9 * pl.SPT * pl.NTW * 1024, 0.0};
    half_t* buf = nullptr;
    {
      std::lock_guard<std::mutex> lk(g_pack_mu);
      PackSlot& ps = g_pack[{p.wt, p.halo_wt ^ ((unsigned long long)tp.BN << 52)}];
      if (ps.bytes < tp.bytes) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(stream, &cs);
        CVX_CHECK(cs == hipStreamCaptureStatusNone, "conv_tile: the packed-weight buffer cannot be allocated while the stream is capturing");
        if (ps.buf) CVX_HIP(hipFree(ps.buf));
        CVX_HIP(hipMalloc((void**)&ps.buf, tp.bytes));
        ps.bytes = tp.bytes;
      }
      buf = ps.buf;
    }
    TilePackJob j;
    const int nblk = cvx_conv_tile_fill_job(p, tp, buf, 0, &j);
    hipLaunchKernelGGL(tile_pack_kernel, dim3(nblk), dim3(256), 0, stream, j);
    packed = buf;
  }
  TileArgs a;
  memset(&a, 0, sizeof(a));
  a.TR = pl.TR;
  a.tiles_per_img = pl.tpi;
  a.ntiles = pl.ntiles;
  a.NB = pl.NB;
  a.P = p.Cin / 8;
  a.pow2 = (a.P & (a.P - 1)) == 0 ? 1 : 0;
  a.logP = 0;
  while ((1 << a.logP) < a.P) ++a.logP;
  a.sh = a.P == 4 ? 2 : (a.P == 8 ? 1 : 0);
  a.swmask = a.pow2 ? (a.P >= 16 ? 15 : a.P - 1) : 0;
  a.phmask = a.pow2 ? a.P - 1 : -1;
  a.SPT = pl.SPT;
  a.NSTEPS = 9 * pl.SPT;
  a.KSC = pl.KSC;
  a.NCH = pl.NCH;
  a.R = pl.R;
  a.CP = pl.KSC * pl.NTW;
  a.PW = (a.CP + kTileWaves - 1) / kTileWaves;
  a.PP = pl.PP;
  a.PPW = (pl.PP + kTileWaves - 1) / kTileWaves;
  a.units = pl.units;
  a.Wp = p.IW + 2;
  a.magic_wp = magic_of(a.Wp);
  a.magic_w = magic_of(p.IW);
  a.magic_p = magic_of(a.P);
  a.wring_off = pl.wring_off;
  a.CB = pl.CB;
  a.stat_off = pl.stat_off;
  a.in_records = (unsigned)(((long long)(p.B - 1) * p.in_bstride + ((long long)p.IH * p.IW - 1) * p.in_ld + p.Cin) * 2);
  a.wpk = packed;
  static const int dbg = cvx_tune_int("CVX_TILE_DBG", 0);
  a.dbg = dbg;
  int rc;
  if (pl.MT <= 2)
    rc = cvx_conv_tile_launch_k1(pl.MT, pl.NTW, p, a, pl.lds, stream);
  else
    rc = cvx_conv_tile_launch_k2(pl.MT, pl.NTW, p, a, pl.lds, stream);
  CVX_TRY(rc);
  CVX_HIP(hipGetLastError());
  return 0;
}

extern "C" int cvx_debug_conv_tile_plan(int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t* out8) {
  CVX_CHECK(out8 && batch > 0 && h > 0 && w > 0, "bad arguments");
  ConvParams p;
  memset(&p, 0, sizeof(p));
  static half_t dummy_zero[8];
  p.zeros = dummy_zero;  // (only tested for presence)
  p.halo_taps_ok = 1;
  p.ntaps = 9;
  p.IS = p.OS = 1;
  p.B = batch;
  p.IH = p.OH2 = h;
  p.IW = p.OW2 = p.OWr = w;
  p.Cin = p.in_ld = cin;
  p.in_bstride = (long long)h * w * cin;
  p.Cout = cout;
  if (!cvx_conv_tile_shape_ok(p)) return -1;
  Plan pl;
  if (!plan_tile(p, &pl)) return -1;
  out8[0] = pl.TR;
  out8[1] = pl.MT;
  out8[2] = pl.NTW;
  out8[3] = pl.NB;
  out8[4] = pl.ntiles * pl.NB;
  out8[5] = pl.KSC;
  out8[6] = pl.lds;
  out8[7] = (int)(pl.cost * 100.0);
  return 0;
}

void cvx_conv_tile_release() {
  std::lock_guard<std::mutex> lk(g_pack_mu);
  for (auto& kv : g_pack)
    if (kv.second.buf) (void)hipFree(kv.second.buf);
  g_pack.clear();
}
