// Implicit-GEMM convolution on MFMA (gfx950): shared parameter block for the forward,
// data-gradient and weight-gradient kernels.  All tensors are NHWC fp16 "views": a base pointer
// (already advanced by the channel offset), a per-pixel stride `ld` and a per-image stride
// `bstride` (both in elements), so channel slices of concat buffers are read and written in place.
#pragma once
#include "cvx_common.h"
#include "bn_act.h"  // cvx_stat_replicas

enum {
  CVX_EPI_RAW_STATS = 0,    // train fwd: raw conv output FP32 (out32) + per-channel (sum, sumsq) into the fixed-point replica slabs
  CVX_EPI_AFFINE_SILU = 1,  // eval fwd: act(y*scale+shift [+res]) [+res] -> fp16; act / residual order: ConvParams::act_kind, res_pre
  CVX_EPI_BIAS_F32 = 2,     // head output: +bias -> fp32
  CVX_EPI_PLAIN = 3,        // dgrad: fp16 store (optionally accumulate into the destination)
};

#define CVX_MAX_TAPS 64

// One tap of the gather: input pixel = (o2*IS + dh, o2w*IS + dw); weight block index wtap.
struct ConvTap {
  int dh, dw, wtap, pad_;
};

struct ConvParams {
  // ---- gathered operand (activations for fwd, output-gradients for dgrad) ----
  const half_t* in;
  long long in_bstride;
  int in_ld;
  int IH, IW;
  int Cin;  // multiple of 8
  // ---- weights: [Cout rows][wt_ld], tap block `wtap` starts at wtap*Cin ----
  const half_t* wt;
  int wt_ld;
  int Cout;
  // ---- output pixel enumeration: m = (b*OH2 + oh2)*OW2 + ow2 ----
  int B, OH2, OW2;
  int IS;            // input coordinate multiplier
  int OS, oph, opw;  // real output pixel = (oh2*OS+oph, ow2*OS+opw)
  int OWr;           // real output width
  int ntaps;
  const ConvTap* taps;  // device memory, ntaps entries
  // ---- epilogue ----
  int epi;
  int accumulate;
  half_t* out16;
  float* out32;
  long long out_bstride;
  int out_ld;
  const half_t* res;  // optional residual (same pixel coordinates as the output)
  long long res_bstride;
  int res_ld;
  const float* scale;  // AFFINE_SILU
  int act_kind;        // AFFINE_SILU: 0 SiLU (YOLO), 1 ReLU, 2 none (DLA / heads)
  int res_pre;         // AFFINE_SILU: 1 = the residual is added BEFORE the activation (DLA BasicBlock), 0 = after (YOLO Bottleneck)
  const float* shift;  // AFFINE_SILU
  const float* bias;   // BIAS_F32
  long long* stats;    // RAW_STATS: [stats_replicas][Cout][2] fixed-point values of CVX_FIX_WORDS words (cvx_fix_atomic_add), zero on entry
  int stats_replicas;
  const half_t* zeros; // >= 16 zero bytes in device memory, 16-byte aligned: DMA source for padding
  int dbg;             // timing experiments only (CVX_DBG): 1 = halo kernel streams the weights once, 2 = no MFMA; results are WRONG
  unsigned long long* clk;  // tuning aid (cvx_debug_clock_buffer): thread 0 of every block stores 100 MHz timestamps, 8 slots per block
  int halo_taps_ok;    // 1 when the tap table is a 3x3 neighbourhood (all |dh|,|dw| <= 1): the LDS halo-tile kernel may be used
  // ---- optional: several launches that differ only in their tap subset and output phase (the 4 phases of a stride-2
  // data gradient) as ONE launch of the DMA-ring kernel: blockIdx.z selects the phase, gridDim.x covers the largest ----
  int nphase;  // 0 / 1: the fields above describe the launch
  struct Phase {
    const ConvTap* taps;
    int ntaps, OH2, OW2, oph, opw;
  } phase[4];
  int pointwise;       // 1 when the table is the single tap (0, 0, weight tap 0): a 1x1 convolution (conv_pw.hip)
  unsigned long long halo_pos, halo_wt;  // cvx_halo_pack_taps of the table (valid when halo_taps_ok): 4 bits per tap
  // GEMM-shaped kernel: the weights already in its ring image order for channel tiles of wt_packed_bn rows (cvx_conv_gemm_pack_jobs, once
  // per forward for all layers); null: the launch packs them itself
  const half_t* wt_packed;
  int wt_packed_bn, wt_packed_kc;  // channel-tile rows and K-values per chunk of that image
  int std7x7;                      // 1 when the table is the 7x7 / pad 3 / dilation 1 neighbourhood in row-major order (conv_stem7.hip)
  int std3x3;                      // likewise 3x3 / pad 1 (cvx_taps_std3x3)
  int gemm_variant;                // unit tests: run this variant of the GEMM-shaped kernel (conv_gemm.hip: kVariants index + 1), 0 = the cost model's
  // row-band kernel (conv_tile.hip): the weights in its LDS image order for channel blocks of tile_packed_bn rows (cvx_conv_tile_pack_jobs,
  // once per forward for all layers); null: the launch packs them itself
  const half_t* tile_packed;
  int tile_packed_bn;
  int no_tile;                     // unit tests / A-B timing: keep this launch off the row-band kernel
  // Data gradient of a stride-2 convolution as ONE stride-1 GEMM (engine.hip: plan_ps_dgrads): the four output phases are channel blocks
  // of ps_cin channels each -- output channel n = (2 * ph + pw) * ps_cin + c goes to pixel (oh2 * OS + ph, ow2 * OS + pw), channel c
  // (OS = 2, oph = opw = 0).  0: off.  Served by the GEMM-shaped kernel only (plain epilogue).
  int ps_cin;
  // RAW_STATS: 1 = the raw output goes out rounded to fp16 (out16 / out_ld / out_bstride) instead of fp32 (out32): CVX_OPF_RAW_F16 layers.
  // The statistics are summed from the fp32 accumulators either way.
  int raw16;
};

// RAW_STATS store of 4 consecutive channels of one pixel at element offset `off` of the raw-output tensor
__device__ __forceinline__ void cvx_store_raw4(const ConvParams& p, long long off, const f4& v) {
  if (p.raw16) *reinterpret_cast<h4*>(p.out16 + off) = h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
  else *reinterpret_cast<f4*>(p.out32 + off) = v;
}

// Packs a 9-entry tap table whose offsets all lie in the 3x3 neighbourhood into two 64-bit words, 4 bits per tap:
// pos = (dh+1)*4 + (dw+1), wt = weight tap index.  Kernel arguments instead of a device table: the halo kernel reads
// no tap memory at all.  Returns false (and the generic kernels are used) for any other table.
inline bool cvx_halo_pack_taps(const ConvTap* t, int n, unsigned long long* pos, unsigned long long* wt) {
  *pos = *wt = 0;
  if (n != 9) return false;
  for (int i = 0; i < 9; ++i) {
    if (t[i].dh < -1 || t[i].dh > 1 || t[i].dw < -1 || t[i].dw > 1 || t[i].wtap < 0 || t[i].wtap > 15) return false;
    *pos |= (unsigned long long)(((t[i].dh + 1) << 2) | (t[i].dw + 1)) << (4 * i);
    *wt |= (unsigned long long)t[i].wtap << (4 * i);
  }
  return true;
}

// tuning aid: when set (cvx_debug_clock_buffer), the DMA-ring and halo kernels store per-block phase timestamps there
extern unsigned long long* g_cvx_clk;
// share of the chip a persistent conv launch should size its grid for: 1 = all CUs, n = 1/n of them (set by the engine
// around launches on concurrent lanes, so that side-by-side persistent kernels do not queue behind each other)
extern int g_cvx_grid_div;
// Validates and dispatches to one of the kernels below (m_blocks: legacy out-parameter, always 0).
int cvx_conv_igemm_launch(const ConvParams& p, hipStream_t stream, int* m_blocks);
// LDS-DMA ring kernel (conv_igemm_dma.hip): every shape the two persistent kernels do not take
int cvx_conv_igemm_dma_launch(const ConvParams& p, hipStream_t stream);
// 3x3 stride-1 halo-tile kernel (conv_halo.hip)
bool cvx_conv_halo_supported(const ConvParams& p);
int cvx_conv_halo_launch(const ConvParams& p, hipStream_t stream);
// 7x7 first layer on the 8-channel-padded image (conv_stem7.hip)
bool cvx_conv_stem7_supported(const ConvParams& p);
int cvx_conv_stem7_launch(const ConvParams& p, hipStream_t stream);
inline int cvx_taps_std7x7(const ConvTap* t, int n) {
  if (n != 49) return 0;
  for (int i = 0; i < 49; ++i)
    if (t[i].dh != i / 7 - 3 || t[i].dw != i % 7 - 3 || t[i].wtap != i) return 0;
  return 1;
}
// GEMM-shaped kernel for the big-channel layers (conv_gemm.hip)
// one layer's weights -> ring image order ([channel tile][chunk][K-step][k-half][BN rows][8]); blocks [blk0, blk0 + nblk) of the batched launch
struct GemmPackJob {
  const half_t* src;
  half_t* dst;
  const ConvTap* taps;  // device table: chunk block `tap` reads weight tap taps[tap].wtap
  int src_ld, rows, Cin, ntaps, BN, kc, nblocks, chunks;  // kc: K-values per chunk (32 or 64)
  int blk0, nblk;
};
// Fills `job` (all but dst, blk0) and the bytes of its image when the dispatcher will take this launch to the GEMM-shaped kernel
bool cvx_conv_gemm_plan(const ConvParams& p, GemmPackJob* job, size_t* bytes);
int cvx_conv_gemm_pack_jobs(const GemmPackJob* d_jobs, int njobs, int nblocks, hipStream_t stream);
bool cvx_conv_gemm_shape_ok(const ConvParams& p);   // what the kernel can run at all
bool cvx_conv_gemm_supported(const ConvParams& p);  // ... and where the dispatcher prefers it
int cvx_conv_gemm_launch(const ConvParams& p, hipStream_t stream);
void cvx_conv_gemm_release();  // packed-weight buffers of the stand-alone launches
// row-band kernel for the small 3x3 stride-1 maps (conv_tile.hip)
struct TilePackPlan {
  int BN, NB, SPT;   // channel-block rows, channel blocks, K-steps per tap of the packed image
  size_t bytes;      // size of the image
  double cost_us;    // the launcher's estimate for the launch (dispatcher: compared against the other kernels' measured floors)
};
bool cvx_conv_tile_shape_ok(const ConvParams& p);                    // what the kernel can run at all
bool cvx_conv_tile_supported(const ConvParams& p);                   // ... and where the dispatcher prefers it
bool cvx_conv_tile_plan(const ConvParams& p, TilePackPlan* out);     // packed-image geometry of the launch cvx_conv_tile_launch will make
int cvx_conv_tile_fill_job(const ConvParams& p, const TilePackPlan& tp, half_t* dst, int blk0, void* job_out);  // -> pack blocks of the job
size_t cvx_conv_tile_job_bytes();
int cvx_conv_tile_pack_jobs(const void* d_jobs, int njobs, int nblocks, hipStream_t stream);
int cvx_conv_tile_launch(const ConvParams& p, hipStream_t stream);
void cvx_conv_tile_release();
// pointwise (1x1 stride-1) persistent GEMM kernel (conv_pw.hip)
bool cvx_conv_pw_supported(const ConvParams& p);
int cvx_conv_pw_launch(const ConvParams& p, hipStream_t stream);
inline int cvx_taps_pointwise(const ConvTap* t, int n) { return (n == 1 && t[0].dh == 0 && t[0].dw == 0 && t[0].wtap == 0) ? 1 : 0; }

// Weight gradient: dW[co][tap][ci] partial sums over a slice of the pixels, written as fp32 slabs.
struct WgradParams {
  const half_t* x;  // forward input view
  long long x_bstride;
  int x_ld;
  int IH, IW, Cin;  // Cin multiple of 8
  const half_t* dy;  // gradient of the raw conv output: element (b, pix, co) at dy[b*dy_bstride + pix*dy_ld + co]
  long long dy_bstride;
  int dy_ld;
  int Cout;
  int B, OH, OW;  // M = B*OH*OW
  int stride;
  int ntaps;
  const ConvTap* taps;  // dh/dw relative to oh*stride, wtap = tap index in the weight layout
  float* slabs;         // [nsplit][Cout][ntaps*Cin]
  int nsplit;
  int cin_pad16;  // Cin rounded up to 16 (column tiling unit)
  int std3x3;     // 1 when taps[t] == (t/3-1, t%3-1, wtap t), t = 0..8: the register-tile kernel (conv_wgrad_halo.hip) applies
};
int cvx_conv_wgrad_launch(const WgradParams& p, hipStream_t stream);
void cvx_conv_wgrad_tile(int cout, int jtot, int* co_b, int* j_b);  // (co, j) tile the generic kernel takes for a layer
// GEMM-shaped weight gradient for the big-channel layers (conv_wgrad_gemm.hip); `supported` needs the geometry and strides filled in
bool cvx_conv_wgrad_gemm_supported(const WgradParams& p);
void cvx_conv_wgrad_gemm_tile(int cout, int jtot, int* co_b, int* j_b);
int cvx_conv_wgrad_gemm_launch(const WgradParams& p, hipStream_t stream);
// 3x3 stride-1 kernel with the whole (co block x 9 taps x ci block) tile in registers (conv_wgrad_halo.hip)
bool cvx_conv_wgrad_halo_supported(const WgradParams& p);
void cvx_conv_wgrad_halo_grid(int cout, int cin, int* gx, int* gy);
int cvx_conv_wgrad_halo_tiles(int B, int OH, int OW);
int cvx_conv_wgrad_halo_launch(const WgradParams& p, hipStream_t stream);
// streaming kernel for the narrow 1x1 / 3x3 stride-1 layers (conv_wgrad_stream.hip): LDS-DMA ring, padded-linear pixel space
bool cvx_conv_wgrad_stream_supported(const WgradParams& p);  // needs geometry, channel counts, strides and cin_pad16 filled in
int cvx_conv_wgrad_stream_nsplit(const WgradParams& p);      // the planner's pixel-split count (the engine sizes the slabs with it)
int cvx_conv_wgrad_stream_launch(const WgradParams& p, hipStream_t stream);
// fat-workgroup kernel for the 3x3 stride-1 layers with 16 ... 144 channels (conv_wgrad_k3.hip): 8 waves, a CU's whole LDS, few workgroups
bool cvx_conv_wgrad_k3_supported(const WgradParams& p);
int cvx_conv_wgrad_k3_nsplit(const WgradParams& p, bool wide = false);  // wide: the tail of the backward pass -- the whole chip
int cvx_conv_wgrad_k3_launch(const WgradParams& p, hipStream_t stream);
inline int cvx_taps_std3x3(const ConvTap* t, int n) {
  if (n != 9) return 0;
  for (int i = 0; i < 9; ++i)
    if (t[i].dh != i / 3 - 1 || t[i].dw != i % 3 - 1 || t[i].wtap != i) return 0;
  return 1;
}
