// BatchNorm + SiLU streaming passes (see bn_act.hip).
#pragma once
#include "cvx_common.h"
#define CVX_BN_MAX_C 2048    // widest BatchNorm the streaming passes hold coefficients for (one channel group of 8 per thread: ResNet-101 layer4)
// Replica slabs the reduction kernels scatter their atomics over, by channel count.  Every consumer block folds all R
// replicas of all C channels first, so R * C is held at ~512: 16 replicas up to 32 channels, 2 from 256 channels on (the
// fold used to be 131 KB per block at 256 channels -- more than the block's share of the tensor on the 20x20 layers).
__host__ __device__ constexpr int cvx_stat_replicas(int C) { return C <= 32 ? 16 : C <= 64 ? 8 : C <= 128 ? 4 : C <= 256 ? 2 : 1; }
#define CVX_STAT_REPLICAS_MAX 16
#define CVX_STAT_WORDS (2 * CVX_FIX_WORDS)  // 64-bit words per channel and replica: (value 0, value 1) x (coarse, fine)

// fp16 NHWC channel-slice view: element (b, pix, c) at p[b*bstride + pix*ld + c]
struct ViewDesc {
  half_t* p;
  long long bstride;
  int ld;
};

struct BnCoef {
  const float* invstd;  // batch 1/sqrt(var+eps) of the forward pass
  const float* gamma;
  const float* beta;
  const float* mean = nullptr;  // not null: the kept tensor is the RAW conv output y in fp16 (CVX_OPF_RAW_F16), xhat = (y - mean) * invstd on the fly
};

// training-mode forward: fixed-point statistics replicas [R][C][2][CVX_FIX_WORDS] filled by the conv epilogue (zero before it)
struct BnTrainArgs {
  const long long* stats;
  const float* gamma;
  const float* beta;
  float* mean;    // out: batch mean   (kept for backward)
  float* invstd;  // out: 1/sqrt(var+eps)
  float* rmean;   // running statistics, updated with `momentum` (unbiased variance)
  float* rvar;
  float eps, momentum;
  const float* cbias = nullptr;  // the convolution's own bias in front of the BatchNorm (VGG-BN, CVX_OPF_CONV_BIAS): the statistics are
                                 // taken on the bias-free output (the normalised value is the same); only the running mean needs it
};

int cvx_stream_rows_per_block(long long M, int C, int kb_per_block);

// every BN layer of the network in one launch (eval-mode forward): one workgroup per descriptor
struct BnFoldDesc {
  long long gamma_off, beta_off;  // element offsets into the parameter arena
  long long rmean_off, rvar_off;  // ... into the statistics arena
  float* scale;                   // outputs (C floats each)
  float* shift;
  int C;
  int bias_only;                  // 1: no BatchNorm -- scale = 1, shift = params[beta_off] (a conv bias feeding an activation)
  long long cbias_off;            // >= 0: the conv has a bias of its own in front of the BatchNorm (VGG-BN): shift += scale * bias
};
int cvx_bn_fold_all(const BnFoldDesc* descs, int n, const float* params, const float* stats, float eps, hipStream_t st);
int cvx_bn_fold(int n, const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps, float* scale, float* shift,
                hipStream_t st);
// y: raw conv output, FP32 [M][C]; writes xhat = (y-mean)*invstd as fp16 [M][C] (operand of the backward passes) and
// silu(gamma*xhat+beta) (+res) into the `out` view
int cvx_bn_silu_apply(const float* y, long long M, int C, int hw, const BnTrainArgs& a, const ViewDesc& out, const ViewDesc& res,
                      half_t* xhat, hipStream_t st);
// ... with the activation kind (0 SiLU, 1 ReLU, 2 none) and the position of the residual (res_pre: inside the activation,
// ResNet's relu(bn(conv) + identity); else added to the activation's output, YOLO's x + cv2(cv1(x)))
int cvx_bn_act_apply(const float* y, long long M, int C, int hw, const BnTrainArgs& a, const ViewDesc& out, const ViewDesc& res, int act,
                     int res_pre, half_t* xhat, hipStream_t st);
// ... of a CVX_OPF_RAW_F16 layer: y16 is the raw conv output rounded to fp16 (it stays, for the backward pass: no xhat is written); SiLU, no residual
// debug aid (cvx_engine_debug_copy): xhat = (y16 - mean) * invstd, rounded to fp16, of a raw-fp16 layer
// ... in fp32, exactly as the backward passes form it (mean == nullptr: `kept` IS xhat, widened)
int cvx_kept_to_xhat_f32(const half_t* kept, long long M, int C, const float* mean, const float* invstd, float* xhat, hipStream_t st);
int cvx_raw16_to_xhat(const half_t* y16, long long M, int C, const float* mean, const float* invstd, half_t* xhat, hipStream_t st);
int cvx_bn_silu_apply_raw16(const half_t* y16, long long M, int C, int hw, const BnTrainArgs& a, const ViewDesc& out, hipStream_t st);
struct BnActKind {
  int act, res_pre;
  ViewDesc fout;  // SiLU + res_pre: the residual's forward value; else unused (ReLU's mask rides in the lowest bit of xhat)
};
// part (zero on entry) receives per-channel (sum y, sum y^2) of an fp32 [M][C] tensor -- the conv epilogues' job in the engine
int cvx_bn_stats_f32(const float* y, long long M, int C, long long* part, hipStream_t st);
// part: replica slabs [R][C][2] fixed-point values, zero on entry; receives (sum dz, sum dz*xhat)
int cvx_bn_bwd_reduce(const half_t* xhat, long long M, int C, int hw, const BnCoef& k, const ViewDesc& gout, const BnActKind& ak, long long* part,
                      hipStream_t st);
// gres receives the incoming gradient (res_pre: the pre-activation gradient dz), accumulated when res_accumulate
int cvx_bn_bwd_apply(const half_t* xhat, long long M, int C, int hw, const BnCoef& k, const long long* part, float inv_scale, float* dgamma,
                     float* dbeta, const ViewDesc& gout, const BnActKind& ak, half_t* dy, const ViewDesc& gres, int res_accumulate, hipStream_t st);
// reduce + apply in ONE launch behind a grid-wide gate (bn_act.hip); 1 = the layer does not qualify, take the two passes.  gate: one 64-bit
// counter, zero on entry (the engine keeps it behind the layer's replica slabs).  Main stream only: see the kernel's comment.
#define CVX_STAT_GATE_WORDS 2  // 64-bit words behind a layer's slabs: (gate counter, spare) -- keeps the next slab 16-byte aligned
int cvx_bn_bwd_fused(const half_t* xhat, long long M, int C, int hw, const BnCoef& k, long long* part, unsigned long long* gate, float inv_scale,
                     float* dgamma, float* dbeta, const ViewDesc& gout, const BnActKind& ak, half_t* dy, const ViewDesc& gres, int res_accumulate,
                     hipStream_t st);
// part: replica slabs [R][C][2], zero on entry
// ... of several tensors that are views of one allocation, in two launches: descs / blocks live in device memory
struct ColsumDesc {
  long long off, bstride;  // element (b, pix, c) of the tensor at base[off + b*bstride + pix*ld + c]
  long long M;             // rows = batch * hw
  int ld, hw, C, rows_per_block;
  long long* part;         // replica slabs [R][C][2] fixed-point values, zero on entry
  long long dbias_off;     // grads[dbias_off + c] += column sum * inv_scale
};
struct ColsumBlock {
  int desc, block;         // workgroup -> (tensor, row chunk of rows_per_block rows)
};
int cvx_colsum_multi(const half_t* base, const ColsumDesc* descs, int ndesc, int max_c, const ColsumBlock* blocks, int nblocks, float inv_scale,
                     float* grads, hipStream_t st);
int cvx_colsum_finalize(const long long* part, int C, float inv_scale, float* dbias, hipStream_t st);  // dbias += inv_scale * folded column sums
int cvx_colsum(long long M, int C, int hw, const ViewDesc& g, long long* part, float inv_scale, float* dbias, hipStream_t st);
