// FP32 stem convolution (3 -> Cout, 3x3 / stride 2 / pad 1) straight from NCHW fp32 images: see stem.hip.
#pragma once
#include "bn_act.h"

struct StemParams {
  const float* img;  // (B, 3, H, W) fp32, NCHW, H and W even
  int B, H, W, OH, OW;
  const float* w;    // fp32 master weights [Cout][kh][kw][ci]
  int Cout;          // 16, 32, 48, 64 or 80
};

// pass 1 (train): per-channel (sum, sumsq) of the raw conv output into the fixed-point replica slabs (zero on entry)
int cvx_stem_stats(const StemParams& p, long long* stats, hipStream_t st);
// pass 2 (train): recompute, (y-mean)*invstd -> xhat fp16 [M][Cout], silu(gamma*xhat+beta) -> out view; block 0 publishes
// mean / invstd and updates the running statistics
int cvx_stem_apply_train(const StemParams& p, const BnTrainArgs& a, const ViewDesc& out, half_t* xhat, hipStream_t st);
// eval: silu(y*scale+shift) -> out view (scale / shift = folded running statistics)
int cvx_stem_apply_eval(const StemParams& p, const float* scale, const float* shift, const ViewDesc& out, hipStream_t st);
// weight gradient from dy fp16 [M][Cout] into fp32 slabs [nsplit][Cout][9 * 16]; nsplit = cvx_stem_wgrad_splits(M)
int cvx_stem_wgrad_splits(long long M);
int cvx_stem_wgrad(const StemParams& p, const half_t* dy, float* slabs, int nsplit, hipStream_t st);
// fused tail of the backward pass: BN backward "apply" (dy formed in registers from xhat / gout and the folded sums of
// bn_bwd_reduce in `part`) + weight gradient; accumulates dgamma / dbeta; dy itself is never stored
int cvx_stem_backward(const StemParams& p, const half_t* xhat, const ViewDesc& gout, const BnCoef& k, const long long* part, float inv_scale,
                      float* dgamma, float* dbeta, float* slabs, int nsplit, hipStream_t st);
// The whole backward of the stem, folded into the gradient arena: dw [Cout][27] (+=), dgamma / dbeta (+=).  Where the images and the two
// gradient-side tensors are 16-byte granular (cvx_stem_backward_onepass_ok) it is ONE pass over them -- the BatchNorm-backward sums come out of
// the weight-gradient kernel instead of going in, `part` is not read and bn_bwd_reduce need not have run; otherwise `part` must hold its sums.
bool cvx_stem_backward_onepass_ok(const StemParams& p, const half_t* xhat, const ViewDesc& gout, int nsplit);
int cvx_stem_backward_fold(const StemParams& p, const half_t* xhat, const ViewDesc& gout, const BnCoef& k, const long long* part, float inv_scale,
                           float* dgamma, float* dbeta, float* dw, float* slabs, int nsplit, hipStream_t st);
// false: cvx_stem_backward_fold (given BnCoef::mean) recomputes xhat from the images, and the training forward need not store it --
// cvx_stem_apply_train accepts xhat == nullptr.  Decided from the same conditions on both sides.
bool cvx_stem_keeps_xhat(const StemParams& p, const ViewDesc& gout, int nsplit);
