// BatchNorm + SiLU streaming passes (see bn_act.hip).
#pragma once
#include "cvx_common.h"
#define CVX_STAT_REPLICAS 32  // replica slabs the reduction kernels scatter their float atomics over

// fp16 NHWC channel-slice view: element (b, pix, c) at p[b*bstride + pix*ld + c]
struct ViewDesc {
  half_t* p;
  long long bstride;
  int ld;
};

struct BnCoef {
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
};

int cvx_stream_rows_per_block(long long M, int C);
int cvx_stream_blocks(long long M, int C);  // = number of partial slabs the reduce kernels write

int cvx_bn_finalize(float* part, int P, int C, long long count, float momentum, float eps, float* mean, float* invstd, float* rmean,
                    float* rvar, hipStream_t st);
int cvx_bn_fold(int n, const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps, float* scale, float* shift,
                hipStream_t st);
int cvx_bn_silu_apply(const half_t* y, long long M, int C, int hw, const BnCoef& k, const ViewDesc& out, const ViewDesc& res, hipStream_t st);
int cvx_bn_bwd_reduce(const half_t* y, long long M, int C, int hw, const BnCoef& k, const ViewDesc& gout, float* part, hipStream_t st);
int cvx_bn_bwd_finalize(float* part, int P, int C, long long count, float inv_scale, float* c1, float* c2, float* dgamma, float* dbeta,
                        hipStream_t st);
int cvx_bn_bwd_apply(const half_t* y, long long M, int C, int hw, const BnCoef& k, const float* c1, const float* c2, const ViewDesc& gout,
                     half_t* dy, const ViewDesc& gres, int res_accumulate, hipStream_t st);
int cvx_colsum(long long M, int C, int hw, const ViewDesc& g, float* part, float inv_scale, float* dbias, hipStream_t st);
