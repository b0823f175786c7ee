// Layout, pooling, resampling, weight-packing, gradient-slab reduction and Adam kernels.
#pragma once
#include "bn_act.h"

int cvx_image_to_nhwc8(const float* img_nchw, int B, int H, int W, half_t* out, hipStream_t st);
// 2x2 / stride 2 max pool, depthwise ConvTranspose2d (kernel 2f, stride f, padding f/2; w fp32 [C][2f][2f]), channel-slice copy
int cvx_maxpool2(const ViewDesc& in, const ViewDesc& out, int B, int IH, int IW, int OH, int OW, int C, hipStream_t st);  // floor or ceil mode
// L2Normalize over channels with a learned per-channel scale (SSD conv4_3)
int cvx_l2norm(const ViewDesc& in, const ViewDesc& out, const float* weight, int B, int HW, int C, hipStream_t st);
int cvx_dwconvt(const ViewDesc& in, const ViewDesc& out, const float* w, int B, int IH, int IW, int C, int f, hipStream_t st);
int cvx_copy_slice(const ViewDesc& in, const ViewDesc& out, int B, int HW, int C, hipStream_t st);
// DeepLabv3+ (inference): 3x3 / stride 2 / pad 1 max pool, global average pool -> (B, 1, 1, C), bilinear resize with
// align_corners = False on fp16 NHWC views, and fp32 rows (B, IH*IW, ld) -> NCHW fp32 (B, C, OH, OW)
// 3x3, pad 1, stride 1 | 2; idx (optional): argmax byte per output element [B][OH*OW][C], the operand of the backward gather
int cvx_maxpool3(const ViewDesc& in, const ViewDesc& out, int B, int IH, int IW, int C, int stride, uint8_t* idx, hipStream_t st);
int cvx_maxpool3_bwd(const ViewDesc& gout, const ViewDesc& gin, int B, int IH, int IW, int C, int stride, const uint8_t* idx, int accumulate,
                     hipStream_t st);
// gradient of the 2x2 stride-2 max pool (floor or ceil mode); `in` = the forward input (the argmax is re-derived from it)
int cvx_maxpool2_bwd(const ViewDesc& in, const ViewDesc& gout, const ViewDesc& gin, int B, int IH, int IW, int OH, int OW, int C, int accumulate,
                     hipStream_t st);
// L2Normalize backward: gin (+)= dx, dweight += inv_scale * dw; partial: cvx_l2norm_bwd_blocks(B*HW) * C floats of scratch
int cvx_l2norm_bwd_blocks(long long npix);
int cvx_l2norm_bwd(const ViewDesc& in, const ViewDesc& gout, const ViewDesc& gin, const float* weight, float* dweight, float inv_scale, int B, int HW,
                   int C, int accumulate, float* partial, hipStream_t st);
int cvx_nchw_cols_grad_to_pred_launch(const float* g, long long g_bstride, long long g_off, int C, int B, int A, int a_off, int HW, float scale,
                                      half_t* dpred, int ld, int col0, hipStream_t st);
int cvx_add_slice(const ViewDesc& in, const ViewDesc& out, int B, int HW, int C, hipStream_t st);  // out += in
// depthwise transposed conv (kernel 2f, stride f, padding f/2): data gradient into gin, weight gradient ACCUMULATED (x inv_scale) into dw fp32 [C][2f][2f];
// part: cvx_dwconvt_bwd_scratch_floats(B*IH*IW, C, f) floats of scratch (per-workgroup partial sums, summed in index order)
long long cvx_dwconvt_bwd_scratch_floats(long long npix, int C, int f);
int cvx_dwconvt_bwd(const ViewDesc& in, const ViewDesc& gout, const ViewDesc& gin, const float* w, float* dw, float inv_scale, int B, int IH, int IW,
                    int C, int f, int accumulate, float* part, hipStream_t st);
// conv + bias (+ ReLU) without BatchNorm: dy = gout (* [fout > 0]) dense fp16 [M][C]; dbias += inv_scale * column sums (part: zeroed replica slabs)
int cvx_bias_act_bwd(const ViewDesc& gout, const ViewDesc& fout, int relu, long long M, int C, int hw, half_t* dy, long long* part, float inv_scale,
                     float* dbias, hipStream_t st);
int cvx_zero_slice(const ViewDesc& v, int B, int HW, int C, hipStream_t st);  // zero fill of a channel slice
int cvx_avgpool_global_bwd(const ViewDesc& gout, const ViewDesc& gin, int B, int HW, int C, int accumulate, hipStream_t st);
int cvx_resize_bilinear_bwd(const ViewDesc& gout, const ViewDesc& gin, int B, int IH, int IW, int OH, int OW, int C, int accumulate,
                            hipStream_t st);
// inverted dropout with a counter-based mask of (seed, element index): forward and backward are the same kernel (backward: in = gout,
// out = gin, same seed); accumulate adds onto `out`
int cvx_dropout(const ViewDesc& in, const ViewDesc& out, int B, int HW, int C, float p, unsigned long long seed, int accumulate, hipStream_t st);
int cvx_avgpool_global(const ViewDesc& in, const ViewDesc& out, int B, int HW, int C, hipStream_t st);
int cvx_resize_bilinear(const ViewDesc& in, const ViewDesc& out, int B, int IH, int IW, int OH, int OW, int C, hipStream_t st);
int cvx_resize_bilinear_f32_nchw(const float* in, int ld, int B, int C, int IH, int IW, int OH, int OW, float* out, hipStream_t st);

// 5x5 / stride 1 / pad 2 max pool on channel-slice views; idx (optional, train) records the argmax
// tap (0..24, first max in row-major window order as torch does) per output element: [B*H*W][C] bytes.
int cvx_maxpool5_fwd(const ViewDesc& in, const ViewDesc& out, int B, int H, int W, int C, uint8_t* idx, hipStream_t st);
// gin (+)= scatter of gout through idx, written gather-style (no atomics)
int cvx_maxpool5_bwd(const ViewDesc& gout, const ViewDesc& gin, int B, int H, int W, int C, const uint8_t* idx, int accumulate, hipStream_t st);
// SPPF's three chained 5x5 pools (y1 = m(x), y2 = m(y1), y3 = m(y2)) and their backward chain as one launch each
bool cvx_sppf_pool3_fits(int H, int W);
int cvx_sppf_pool3_fwd(const ViewDesc& in, const ViewDesc& o1, const ViewDesc& o2, const ViewDesc& o3, int B, int H, int W, int C, uint8_t* i1, uint8_t* i2,
                       uint8_t* i3, hipStream_t st);
int cvx_sppf_pool3_bwd(const ViewDesc& g3, const ViewDesc& g2, const ViewDesc& g1, const ViewDesc& g0, int B, int H, int W, int C, const uint8_t* i1,
                       const uint8_t* i2, const uint8_t* i3, int acc_mask /* bit s: stage s accumulates into its input gradient */, hipStream_t st);

int cvx_upsample2_fwd(const ViewDesc& in, const ViewDesc& out, int B, int H, int W, int C, hipStream_t st);  // (H,W) -> (2H,2W)
int cvx_upsample2_bwd(const ViewDesc& gout, const ViewDesc& gin, int B, int H, int W, int C, int accumulate, hipStream_t st);

// copy pred (B, A, no) fp32 level slice -> NCHW fp32 (B, no, H, W)   (API-compat outputs)
int cvx_pred_to_nchw(const float* pred, int B, int A, int no, int a_off, int H, int W, float* out, hipStream_t st);
int cvx_pred_cols_to_nchw_launch(const float* rows, int ld, int col0, int C, int B, int A, int a_off, int HW, float* out, long long out_bstride,
                                 long long out_off, hipStream_t st);
int cvx_nchw_to_pred_f16(const float* g_nchw, int B, int A, int no, int a_off, int H, int W, float scale, half_t* dpred, hipStream_t st);

// ---- table-driven multi-tensor kernels ---------------------------------------------------------
struct PackDesc {        // one conv weight tensor
  long long src_off;     // fp32 master, [Cout][T][Cin]
  long long fwd_off;     // fp16 [Cout][T][Cin_pad]
  long long dg_off;      // fp16 [Cin][T][Cout]  (-1: no dgrad copy)
  int Cout, T, Cin, Cin_pad;
};
struct BlockRef {
  int desc;
  int start;  // first element (in units of the kernel's index space) this block handles
};
int cvx_pack_weights(const float* master, half_t* shadow, const PackDesc* descs, const BlockRef* blocks, int nblocks, hipStream_t st);

// Weights of a stride-2 data gradient as ONE stride-1 GEMM over the 2 x 2 window of dy (engine.hip: pixel-shuffle data gradient): fp16
// [4 phases x cin_pad rows][4 window taps x C columns], block (phase, tau) = tap wtap[phase * 4 + tau] of the transposed data-gradient
// shadow [cin_pad][T][C] (-1: the phase does not use that window position -- zeros).
struct PsPackDesc {
  long long dg_off, ps_off;  // element offsets from dg_base / ps_base (the engine: both the fp16 shadow arena)
  int cin_pad, T, C;
  int wtap[16];
};
int cvx_pack_ps_weights(const half_t* dg_base, half_t* ps_base, const PsPackDesc& d, hipStream_t st);

struct SlabDesc {        // one weight-gradient tensor
  long long slab_off;    // fp32 [nsplit][Cout*T][Cin_pad]
  long long dst_off;     // fp32 grad arena [Cout*T][Cin]
  int nsplit, rows, Cin, Cin_pad;
  int lanes;             // split-lanes per block (power of two, 1..64); a block covers 256/lanes elements
};
static inline int cvx_slab_lanes(int nsplit) {
  int l = 1;
  while (l < 64 && l * 8 < nsplit) l <<= 1;
  return l;
}
int cvx_reduce_slabs(const float* slabs, float* grads, float inv_scale, const SlabDesc* descs, const BlockRef* blocks, int nblocks,
                     hipStream_t st);

// Adam (torch.optim.Adam defaults semantics, no weight decay / amsgrad), fp32 state; optionally skipped
// when *found_inf != 0; zeroes the gradient arena afterwards when zero_grad != 0.
int cvx_adam(float* p, float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, int step, const int* found_inf,
             int zero_grad, hipStream_t st);
// same update with the step state on the device (state[0]=lr, [1]=step, [2],[3] derived): replayable from a hipGraph
int cvx_adam_dev(float* p, float* g, float* m, float* v, long long n, float b1, float b2, float eps, float* state, const int* found_inf,
                 int zero_grad, float grad_scale, hipStream_t st);
// sets *found_inf = 1 if any gradient is non-finite
int cvx_check_finite_launch(const float* g, long long n, int* found_inf, hipStream_t st);
