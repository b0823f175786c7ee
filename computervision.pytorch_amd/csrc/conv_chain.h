// Tile-resident convolution CHAINS (gfx950): descriptor shared by the kernel (conv_chain.hip) and the host-side planner.
//
// One workgroup owns one spatial tile of one image and runs a short program on it:
//   LOAD   global NHWC view (+halo, zero outside the image, optional folded nearest-2x upsample) -> LDS plane  (LDS-DMA)
//   PASS   conv (1x1 / 3x3, stride 1 / 2, or a 1x1 over several planes = a fused concat) reading an LDS plane as the MFMA
//          pixel operand, weights streamed through ONE continuous LDS-DMA ring that runs ahead across pass boundaries;
//          epilogue: folded BN + SiLU / ReLU (+ residual from a plane) -> fp16 into another LDS plane (zero outside the
//          image, so the next 3x3 sees the reference's zero padding), or + bias -> fp32 rows in global memory (Detect output)
//   STORE  LDS plane region -> global NHWC view in whole pixel rows (coalesced)
// so a Bottleneck (3x3 -> 3x3 + shortcut), a Bottleneck + the C2f's closing 1x1, or a whole Detect level
// (3x3 -> 3x3|3x3 -> 1x1|1x1 + bias) is one launch and the intermediates never leave the CU.
// Replaces, in eval mode: core/models/yolov8/modules.py:124-135 (Bottleneck), :189-207 (C2f tail), :407-455 (Detect branches).
#pragma once
#include "conv_igemm.h"

#define CHAIN_MAX_PASSES 10
#define CHAIN_MAX_LOADS 8
#define CHAIN_MAX_STORES 3
#define CHAIN_MAX_SEGS 4
#define CHAIN_RING_SLOTS 4

// LDS addresses are in 16-byte UNITS from the start of the plane region; a plane pixel is `ps` units apart (ps odd:
// conflict-free fragment reads, see cvx_chain_pixel_units), of which the first P hold channels.
struct ChainLoad {
  const void* src;    // kind 0: fp16 NHWC view (base already at the first channel); kind 1: raw 16-byte units
  long long bstride;  // elements per image
  int ld;             // elements per pixel
  int kind;
  int IH, IW;         // image size at the plane's resolution (bounds of the zero fill)
  int up;             // 1: folded nearest-2x upsample -- source pixel (iy >> 1, ix >> 1), source row pitch IW >> 1
  int base, ps, P;    // plane: first unit, pixel stride, data units per pixel
  int PW;             // plane width in pixels
  int nunits;         // units to fill (npix * ps; raw: units)
  int scale, y0, x0;  // image coordinates of plane pixel (0, 0): tile_y0 * scale + y0, tile_x0 * scale + x0
  int per_wave;       // DMA instructions per wave (ceil(ceil(nunits / 64) / 8))
  int at_pass;        // issued at the first barrier of this pass (0: before the ring prologue)
};

struct ChainPass {
  // weights: `rows` rows x K = nseg * P * 8, PRE-PACKED chunk by chunk in the ring's LDS image order (cvx_chain_pack)
  const half_t* wpk;
  int K, rows, nchunks;
  // pixel operand: segment s (tap or concat part) reads unit in_off(pixel) + segoff(s) + chunk
  int in_base, in_ps, in_W;  // plane of segment 0 / the taps
  int P;                     // units (8 channels) per segment
  int nseg, ksz;             // ksz 3: segment s = tap (s / 3, s % 3); ksz 1: segoff[] below
  int segoff[CHAIN_MAX_SEGS];  // ksz 1, nseg > 1: unit offset of segment s relative to segment 0's unit of the same pixel
  int cy, cx, s;             // input plane pixel of output region pixel (qy, qx), tap (dh, dw): (cy + qy*s + dh, cx + qx*s + dw)
  int RW, npix;              // output region: width, pixel count (flat enumeration q = qy * RW + qx)
  int drain;                 // 1: full vmcnt(0) before the first chunk (global stores or late loads precede)
  // epilogue
  int par_off, par_off2;     // float offsets of the pass's scale rows and shift rows (fp16 out) or bias rows (fp32 out, par_off) in the LDS parameter table
  int act;                   // 0 SiLU, 1 ReLU, 2 none
  int out_kind;              // 0: fp16 LDS plane, 1: fp32 global rows (+ bias)
  int out_base, out_ps, out_W, oy, ox, out_c0;  // destination plane, pixel (oy + qy, ox + qx), first channel (halves)
  int has_res, res_base, res_ps, res_W, ry, rx, res_c0;
  int img_scale, img_y0, img_x0, IH, IW;  // image coordinates of region pixel (0, 0) and the image size there: out-of-image pixels are stored as zero
  float* out32;              // out_kind 1: element (b, iy * IW + ix, out_c0 + n) of out32 (or, when NULL, of the launch's base pointer) + out32_off
  long long out32_off;
  long long out_bstride;
  int out_ld;
};

struct ChainStore {
  half_t* dst;
  long long bstride;
  int ld;
  int OH, OW;           // image size (bounds)
  int base, ps, PW, P;  // plane, units per pixel to copy
  int py0, px0, RW, RH;  // region inside the plane
  int scale, y0, x0;    // image coordinates of region pixel (0, 0)
};

struct ChainDesc {
  int npass, nload, nstore;
  int tiles_x, tiles_y, TW, TH;  // tile grid per image, tile size at scale 1
  int par_base;                  // unit offset of the fp32 parameter table inside the plane region
  const half_t* zeros;
  int dbg;                       // timing experiments (CVX_TUNING builds, CVX_CHAIN_DBG): 1 no MFMA / fragment reads, 2 no weight DMA, 4 no epilogue; results are WRONG
  unsigned long long* clk;       // tuning aid (cvx_debug_clock_buffer): thread 0 of every workgroup stores 100 MHz stamps, 32 slots per workgroup
  ChainLoad load[CHAIN_MAX_LOADS];
  ChainPass pass[CHAIN_MAX_PASSES];
  ChainStore store[CHAIN_MAX_STORES];
};

// ---- host side: a planned chain (descriptor in device memory + launch geometry) ----
// weight pre-pack job: [rows][K] fp16 (row pitch src_ld) -> ring chunk images (conv_chain.hip: chain_pack_kernel)
struct ChainPackJob {
  const half_t* src;
  half_t* dst;
  int src_ld, rows, K, RR, units;  // units: 16-byte units of the destination
};
#define CHAIN_MAX_JOBS CHAIN_MAX_PASSES
struct ChainPlan {
  ChainDesc* d_desc = nullptr;
  void* d_jobs = nullptr;  // device array of the weight pre-pack jobs (one per pass)
  int njobs = 0, max_job_units = 0;
  ChainPackJob jobs[CHAIN_MAX_JOBS];  // host copy (the engine packs the weights of all its chains in one launch)
  int cfg = -1;        // index into the compiled (MT, NT, KSUB) table
  int lds_bytes = 0;
  int blocks = 0;
  double flops = 0, bytes = 0;  // algorithmic work of the whole chain (profiling)
};

// one conv stage of a chain as the planner describes it (conv_chain.hip: cvx_chain_plan)
struct ChainStageSpec {
  int in_plane, in_c0, cin;  // input plane index and channel slice
  int nextra;                // 1x1 only: further K segments (same cin each), from other planes
  int extra_plane[CHAIN_MAX_SEGS - 1], extra_c0[CHAIN_MAX_SEGS - 1];
  int k, stride;
  const half_t* wt;          // [cout][k*k*(1+nextra)*cin]
  int wt_ld, cout;
  const float *scale, *shift;  // folded BN (fp16 out) ...
  const float* bias;           // ... or bias (fp32 out)
  int live_params;             // 1: the kernel reads scale / shift / bias from these arrays at every launch (shift == scale + cout required);
                               // 0: they are copied into the plan when it is built
  int act;
  int out_plane, out_c0;     // out_plane < 0: fp32 global rows
  int res_plane, res_c0;     // res_plane < 0: none
  int ry0, rx0, RH, RW;      // output region in OUT-plane pixel coordinates (fp32 out: in tile coordinates, scale 1)
  float* out32;              // NULL: the base pointer comes with the launch (cvx_chain_launch's out32_base)
  long long out32_off;       // elements added to the base
  long long out_bstride;
  int out_ld, out32_c0;
};
struct ChainPlaneSpec {
  int scale, y0, x0, PH, PW, C;  // covers image rows tile_y0*scale + y0 .. + PH at resolution `scale`; C channels (multiple of 8)
  int alias;                     // >= 0: lives inside the LDS space of that (dead by then) plane ...
  int alias_off;                 // ... this many 16-byte units from its start
};
struct ChainLoadSpec {
  int plane;
  const half_t* src;
  long long bstride;
  int ld, IH, IW, up, at_pass;
  int c0, c;  // channel slice of the plane that is filled (whole pixels are DMA'd: c0 must be 0 and c the plane's C)
};
struct ChainStoreSpec {
  int plane, c0, c;
  half_t* dst;
  long long bstride;
  int ld, OH, OW;
  int py0, px0, RH, RW;
};
struct ChainSpec {
  int B, TH, TW, OH, OW;  // batch, tile size and image size at scale 1
  int nplanes, nloads, nstages, nstores;
  ChainPlaneSpec planes[8];
  ChainLoadSpec loads[CHAIN_MAX_LOADS];
  ChainStageSpec stages[CHAIN_MAX_PASSES];
  ChainStoreSpec stores[CHAIN_MAX_STORES];
  const half_t* zeros;
};

// Builds the device descriptor of `spec`.  *d_alloc receives the one hipMalloc'd block behind the plan (the caller frees it).
// dry_run: only checks feasibility (LDS budget, geometry) and fills out->lds_bytes / cfg / blocks, allocating nothing.
int cvx_chain_plan(const ChainSpec& spec, ChainPlan* out, void** d_alloc, bool dry_run = false);
// LDS units a plane of C channels needs per pixel: an ODD pixel stride.  The MFMA 32x32x16 pixel operand is read as lane = (pixel
// lane & 31, k-half lane >> 5), 16 bytes each; a ds_read_b128 is served in groups of 16 lanes ({0-3, 12-15, 20-27}, ...) whose pixel
// indices cover all 16 residues mod 16, so with an odd stride they fall on 16 different 16-byte slots of the 256-byte bank row
// (tools/lds_conflicts.py enumerates the strides).
inline int cvx_chain_pixel_units(int C) { return (C / 8) | 1; }
// Re-packs the chain's weights from the fp16 shadows into the ring image order (call after the shadows changed, before the launch).
int cvx_chain_pack(const ChainPlan& plan, hipStream_t stream);
// the same for a device array of jobs gathered from several plans
int cvx_chain_pack_jobs(const ChainPackJob* d_jobs, int njobs, int max_job_units, hipStream_t stream);
int cvx_chain_launch(const ChainPlan& plan, hipStream_t stream, float* out32_base = nullptr);

// ---- specs of the fusion groups (conv_chain.hip) ----
struct ChainConvArgs {
  const half_t* wt;  // [cout][K] fp16 (engine shadow layout: [cout][tap][cin])
  int wt_ld, cout;
  const float *scale, *shift;  // folded BN (fp16 out)
  const float* bias;           // fp32 out
  int act;                     // 0 SiLU, 1 ReLU, 2 none
  int live;                    // ChainStageSpec::live_params
};
// estimated device time (us) of a planned chain on one MI355X: rounds of workgroups x (MFMA time of the per-wave register tiles at the
// measured ~45 % issue efficiency + fixed prologue / epilogue / store time) -- ranks tile shapes, nothing else
double cvx_chain_cost_us(const ChainSpec& spec, const ChainPlan& plan);
int cvx_chain_spec_pair(ChainSpec* sp, const half_t* x, long long x_bs, int x_ld, int B, int H, int W, int C, const ChainConvArgs& c1,
                        const ChainConvArgs& c2, bool shortcut, half_t* out, long long out_bs, int out_ld, int TH, int TW, const half_t* zeros);
int cvx_chain_spec_single(ChainSpec* sp, const half_t* x, long long x_bs, int x_ld, int B, int IH, int IW, int Cin, int k, int stride, int up,
                          const ChainConvArgs& c, half_t* out, long long out_bs, int out_ld, int TH, int TW, const half_t* zeros);
int cvx_chain_spec_detect(ChainSpec* sp, const half_t* x, long long x_bs, int x_ld, int B, int H, int W, int Cin, int cb, int cc, int ncp,
                          const ChainConvArgs& a, const ChainConvArgs& b1, const ChainConvArgs& b2, const ChainConvArgs& o1, const ChainConvArgs& o2,
                          float* pred, long long pred_bs, int pred_ld, int a_off, int TH, int TW, const half_t* zeros);
