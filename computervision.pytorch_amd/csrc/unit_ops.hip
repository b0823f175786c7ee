// Single-op C-ABI entry points over the streaming kernels (BatchNorm+SiLU passes, SPPF pool, nearest upsample, fp32 stem):
// the same kernels the engine launches, on caller-owned dense NHWC tensors -- for unit parity tests and for host code
// that wants one op.  They allocate their small statistic slabs themselves and synchronise before returning.
#include <vector>

#include "../../include/cvx_engine.h"
#include "misc_ops.h"
#include "stem.h"

namespace {

struct Scratch {  // RAII device allocation (zeroed)
  void* p = nullptr;
  int alloc(size_t bytes) {
    CVX_HIP(hipMalloc(&p, bytes));
    CVX_HIP(hipMemset(p, 0, bytes));
    return 0;
  }
  ~Scratch() {
    if (p) (void)hipFree(p);
  }
};
size_t slab_bytes(int C) { return (size_t)cvx_stat_replicas(C) * C * CVX_STAT_WORDS * 8; }
ViewDesc dense(const void* p, int hw, int C) { return ViewDesc{(half_t*)p, (long long)hw * C, C}; }

}  // namespace

extern "C" int cvx_bn_silu_train_nhwc(const float* y_f32, int32_t batch, int32_t hw, int32_t c, const float* gamma, const float* beta, float eps,
                                      float momentum, float* running_mean, float* running_var, const void* res_f16, void* out_f16,
                                      void* xhat_f16, float* mean, float* invstd, void* hip_stream) {
  return cvx_bn_act_train_nhwc(y_f32, batch, hw, c, gamma, beta, eps, momentum, running_mean, running_var, res_f16, 0, 0, out_f16, xhat_f16, mean,
                               invstd, hip_stream);
}
extern "C" int cvx_bn_act_train_nhwc(const float* y_f32, int32_t batch, int32_t hw, int32_t c, const float* gamma, const float* beta, float eps,
                                     float momentum, float* running_mean, float* running_var, const void* res_f16, int32_t act, int32_t res_pre,
                                     void* out_f16, void* xhat_f16, float* mean, float* invstd, void* hip_stream) {
  CVX_CHECK(y_f32 && gamma && beta && running_mean && running_var && out_f16 && xhat_f16 && mean && invstd && batch > 0 && hw > 0, "bad arguments");
  hipStream_t st = (hipStream_t)hip_stream;
  const long long M = (long long)batch * hw;
  Scratch slab;
  CVX_TRY(slab.alloc(slab_bytes(c)));
  CVX_TRY(cvx_bn_stats_f32(y_f32, M, c, (long long*)slab.p, st));
  BnTrainArgs ta{(const long long*)slab.p, gamma, beta, mean, invstd, running_mean, running_var, eps, momentum};
  CVX_TRY(cvx_bn_act_apply(y_f32, M, c, hw, ta, dense(out_f16, hw, c), res_f16 ? dense(res_f16, hw, c) : ViewDesc{nullptr, 0, 0}, act, res_pre,
                           (half_t*)xhat_f16, st));
  CVX_HIP(hipStreamSynchronize(st));
  return 0;
}

extern "C" int cvx_bn_silu_bwd_nhwc(const void* xhat_f16, const void* gout_f16, int32_t batch, int32_t hw, int32_t c, const float* gamma,
                                    const float* beta, const float* invstd, float inv_scale, float* dgamma, float* dbeta, void* dy_f16,
                                    void* gres_f16, int32_t res_accumulate, void* hip_stream) {
  return cvx_bn_act_bwd_nhwc(xhat_f16, gout_f16, nullptr, batch, hw, c, gamma, beta, invstd, 0, 0, inv_scale, dgamma, dbeta, dy_f16, gres_f16,
                             res_accumulate, hip_stream);
}
extern "C" int cvx_bn_act_bwd_nhwc(const void* xhat_f16, const void* gout_f16, const void* out_f16, int32_t batch, int32_t hw, int32_t c,
                                   const float* gamma, const float* beta, const float* invstd, int32_t act, int32_t res_pre, float inv_scale,
                                   float* dgamma, float* dbeta, void* dy_f16, void* gres_f16, int32_t res_accumulate, void* hip_stream) {
  CVX_CHECK(xhat_f16 && gout_f16 && gamma && beta && invstd && dgamma && dbeta && dy_f16 && batch > 0 && hw > 0, "bad arguments");
  CVX_CHECK(!(act == 0 && res_pre) || out_f16, "SiLU with a pre-activation residual: the residual's forward value is needed (pass it as out_f16)");
  const BnActKind ak{act, res_pre, out_f16 ? dense(out_f16, hw, c) : ViewDesc{nullptr, 0, 0}};
  hipStream_t st = (hipStream_t)hip_stream;
  const long long M = (long long)batch * hw;
  Scratch slab;
  CVX_TRY(slab.alloc(slab_bytes(c) + CVX_STAT_GATE_WORDS * 8));
  BnCoef k{invstd, gamma, beta};
  const ViewDesc g = dense(gout_f16, hw, c);
  // the engine's choice: one gated launch where the layer qualifies, else the two passes
  const int one = cvx_bn_bwd_fused((const half_t*)xhat_f16, M, c, hw, k, (long long*)slab.p, (unsigned long long*)((char*)slab.p + slab_bytes(c)), inv_scale,
                                   dgamma, dbeta, g, ak, (half_t*)dy_f16, gres_f16 ? dense(gres_f16, hw, c) : ViewDesc{nullptr, 0, 0}, res_accumulate, st);
  if (one < 0) return one;
  if (one == 0) {
    CVX_HIP(hipStreamSynchronize(st));
    return 0;
  }
  CVX_TRY(cvx_bn_bwd_reduce((const half_t*)xhat_f16, M, c, hw, k, g, ak, (long long*)slab.p, st));
  CVX_TRY(cvx_bn_bwd_apply((const half_t*)xhat_f16, M, c, hw, k, (const long long*)slab.p, inv_scale, dgamma, dbeta, g, ak, (half_t*)dy_f16,
                           gres_f16 ? dense(gres_f16, hw, c) : ViewDesc{nullptr, 0, 0}, res_accumulate, st));
  CVX_HIP(hipStreamSynchronize(st));
  return 0;
}

extern "C" int cvx_maxpool5_nhwc(const void* x_f16, int32_t batch, int32_t h, int32_t w, int32_t c, void* out_f16, uint8_t* argmax,
                                 void* hip_stream) {
  CVX_CHECK(x_f16 && out_f16 && batch > 0 && c % 8 == 0, "bad arguments (channels in multiples of 8)");
  CVX_TRY(cvx_maxpool5_fwd(dense(x_f16, h * w, c), dense(out_f16, h * w, c), batch, h, w, c, argmax, (hipStream_t)hip_stream));
  CVX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  return 0;
}
extern "C" int cvx_maxpool5_bwd_nhwc(const void* gout_f16, const uint8_t* argmax, int32_t batch, int32_t h, int32_t w, int32_t c,
                                     void* gin_f16, int32_t accumulate, void* hip_stream) {
  CVX_CHECK(gout_f16 && argmax && gin_f16 && batch > 0 && c % 8 == 0, "bad arguments (channels in multiples of 8)");
  CVX_TRY(cvx_maxpool5_bwd(dense(gout_f16, h * w, c), dense(gin_f16, h * w, c), batch, h, w, c, argmax, accumulate, (hipStream_t)hip_stream));
  CVX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  return 0;
}
extern "C" int cvx_resize_bilinear_rows_to_nchw(const float* rows_f32, int32_t ld, int32_t batch, int32_t c, int32_t ih, int32_t iw, int32_t oh,
                                                int32_t ow, float* out_nchw, void* hip_stream) {
  CVX_CHECK(rows_f32 && out_nchw && batch > 0, "bad arguments");
  CVX_TRY(cvx_resize_bilinear_f32_nchw(rows_f32, ld, batch, c, ih, iw, oh, ow, out_nchw, (hipStream_t)hip_stream));
  return 0;  // asynchronous on the caller's stream, like the engine's forward it follows
}
extern "C" int cvx_pred_cols_to_nchw(const float* rows, int32_t ld, int32_t col0, int32_t c, int32_t batch, int32_t anchors, int32_t a_off,
                                     int32_t hw, float* out, int64_t out_bstride, int64_t out_off, void* hip_stream) {
  CVX_CHECK(rows && out && batch > 0 && c > 0 && col0 >= 0 && col0 + c <= ld && a_off >= 0 && a_off + hw <= anchors, "bad arguments");
  return cvx_pred_cols_to_nchw_launch(rows, ld, col0, c, batch, anchors, a_off, hw, out, out_bstride, out_off, (hipStream_t)hip_stream);
}
extern "C" int cvx_nchw_cols_grad_to_pred(const float* grad, int64_t grad_bstride, int64_t grad_off, int32_t c, int32_t batch, int32_t anchors,
                                          int32_t a_off, int32_t hw, float scale, void* dpred_f16, int32_t ld, int32_t col0, void* hip_stream) {
  CVX_CHECK(grad && dpred_f16 && batch > 0 && c > 0 && col0 >= 0 && col0 + c <= ld && a_off >= 0 && a_off + hw <= anchors, "bad arguments");
  return cvx_nchw_cols_grad_to_pred_launch(grad, grad_bstride, grad_off, c, batch, anchors, a_off, hw, scale, (half_t*)dpred_f16, ld, col0,
                                           (hipStream_t)hip_stream);
}
extern "C" int cvx_l2norm_bwd_nhwc(const void* x_f16, const void* gout_f16, const float* weight, int32_t batch, int32_t hw, int32_t c, float inv_scale,
                                   void* gin_f16, float* dweight, int32_t accumulate, void* hip_stream) {
  CVX_CHECK(x_f16 && gout_f16 && weight && gin_f16 && dweight && batch > 0, "bad arguments");
  Scratch part;
  CVX_TRY(part.alloc((size_t)cvx_l2norm_bwd_blocks((long long)batch * hw) * c * 4));
  CVX_TRY(cvx_l2norm_bwd(dense(x_f16, hw, c), dense(gout_f16, hw, c), dense(gin_f16, hw, c), weight, dweight, inv_scale, batch, hw, c, accumulate,
                         (float*)part.p, (hipStream_t)hip_stream));
  CVX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  return 0;
}
// ---- the inference-only pooling / resampling / normalisation ops of the DLA, ResNet / DeepLab and VGG / SSD graphs, one by one ----
extern "C" int cvx_maxpool_nhwc(const void* x_f16, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t kernel, int32_t stride,
                                int32_t ceil_mode, void* out_f16, void* hip_stream) {
  CVX_CHECK(x_f16 && out_f16 && batch > 0 && c % 8 == 0, "bad arguments (channels in multiples of 8)");
  hipStream_t st = (hipStream_t)hip_stream;
  if (kernel == 2 && stride == 2) {
    const int oh = ceil_mode ? (h + 1) / 2 : h / 2, ow = ceil_mode ? (w + 1) / 2 : w / 2;
    CVX_TRY(cvx_maxpool2(dense(x_f16, h * w, c), dense(out_f16, oh * ow, c), batch, h, w, oh, ow, c, st));
  } else if (kernel == 3 && (stride == 1 || stride == 2) && !ceil_mode) {
    const int oh = (h - 1) / stride + 1, ow = (w - 1) / stride + 1;
    CVX_TRY(cvx_maxpool3(dense(x_f16, h * w, c), dense(out_f16, oh * ow, c), batch, h, w, c, stride, nullptr, st));
  } else {
    CVX_CHECK(false, "max pools built: 2x2 / stride 2 (floor or ceil mode), 3x3 / pad 1 / stride 1 or 2");
  }
  CVX_HIP(hipStreamSynchronize(st));
  return 0;
}
extern "C" int cvx_avgpool_global_nhwc(const void* x_f16, int32_t batch, int32_t hw, int32_t c, void* out_f16, void* hip_stream) {
  CVX_CHECK(x_f16 && out_f16 && batch > 0 && c % 8 == 0, "bad arguments (channels in multiples of 8)");
  CVX_TRY(cvx_avgpool_global(dense(x_f16, hw, c), dense(out_f16, 1, c), batch, hw, c, (hipStream_t)hip_stream));
  CVX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  return 0;
}
extern "C" int cvx_resize_bilinear_nhwc(const void* x_f16, int32_t batch, int32_t ih, int32_t iw, int32_t c, int32_t oh, int32_t ow,
                                        void* out_f16, void* hip_stream) {
  CVX_CHECK(x_f16 && out_f16 && batch > 0 && c % 8 == 0, "bad arguments (channels in multiples of 8)");
  CVX_TRY(cvx_resize_bilinear(dense(x_f16, ih * iw, c), dense(out_f16, oh * ow, c), batch, ih, iw, oh, ow, c, (hipStream_t)hip_stream));
  CVX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  return 0;
}
extern "C" int cvx_l2norm_nhwc(const void* x_f16, const float* weight, int32_t batch, int32_t hw, int32_t c, void* out_f16, void* hip_stream) {
  CVX_CHECK(x_f16 && weight && out_f16 && batch > 0 && c % 8 == 0, "bad arguments (channels in multiples of 8)");
  CVX_TRY(cvx_l2norm(dense(x_f16, hw, c), dense(out_f16, hw, c), weight, batch, hw, c, (hipStream_t)hip_stream));
  CVX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  return 0;
}
// ---- their training forms: forward with the operand the backward needs, and the backward ----
extern "C" int cvx_maxpool3_train_nhwc(const void* x_f16, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t stride, void* out_f16,
                                       uint8_t* argmax, void* hip_stream) {
  CVX_CHECK(x_f16 && out_f16 && argmax && batch > 0 && c % 8 == 0, "bad arguments (channels in multiples of 8)");
  const int oh = (h - 1) / stride + 1, ow = (w - 1) / stride + 1;
  CVX_TRY(cvx_maxpool3(dense(x_f16, h * w, c), dense(out_f16, oh * ow, c), batch, h, w, c, stride, argmax, (hipStream_t)hip_stream));
  CVX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  return 0;
}
extern "C" int cvx_maxpool3_bwd_nhwc(const void* gout_f16, const uint8_t* argmax, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t stride,
                                     void* gin_f16, int32_t accumulate, void* hip_stream) {
  CVX_CHECK(gout_f16 && argmax && gin_f16 && batch > 0 && c % 8 == 0, "bad arguments (channels in multiples of 8)");
  const int oh = (h - 1) / stride + 1, ow = (w - 1) / stride + 1;
  CVX_TRY(cvx_maxpool3_bwd(dense(gout_f16, oh * ow, c), dense(gin_f16, h * w, c), batch, h, w, c, stride, argmax, accumulate, (hipStream_t)hip_stream));
  CVX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  return 0;
}
extern "C" int cvx_maxpool2_bwd_nhwc(const void* x_f16, const void* gout_f16, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t ceil_mode,
                                     void* gin_f16, int32_t accumulate, void* hip_stream) {
  CVX_CHECK(x_f16 && gout_f16 && gin_f16 && batch > 0 && c % 8 == 0, "bad arguments (channels in multiples of 8)");
  const int oh = ceil_mode ? (h + 1) / 2 : h / 2, ow = ceil_mode ? (w + 1) / 2 : w / 2;
  CVX_TRY(cvx_maxpool2_bwd(dense(x_f16, h * w, c), dense(gout_f16, oh * ow, c), dense(gin_f16, h * w, c), batch, h, w, oh, ow, c, accumulate,
                           (hipStream_t)hip_stream));
  CVX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  return 0;
}
extern "C" int cvx_dwconvt_nhwc(const void* x_f16, int32_t batch, int32_t ih, int32_t iw, int32_t c, int32_t f, const float* weight, void* out_f16,
                                void* hip_stream) {
  CVX_CHECK(x_f16 && weight && out_f16 && batch > 0 && c % 8 == 0 && f >= 2 && f % 2 == 0, "bad arguments (channels in multiples of 8, even stride)");
  CVX_TRY(cvx_dwconvt(dense(x_f16, ih * iw, c), dense(out_f16, ih * f * iw * f, c), weight, batch, ih, iw, c, f, (hipStream_t)hip_stream));
  CVX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  return 0;
}
extern "C" int cvx_dwconvt_bwd_nhwc(const void* x_f16, const void* gout_f16, int32_t batch, int32_t ih, int32_t iw, int32_t c, int32_t f,
                                    const float* weight, void* gin_f16, int32_t accumulate, float* dweight, float inv_scale, void* hip_stream) {
  CVX_CHECK(x_f16 && gout_f16 && weight && gin_f16 && dweight && batch > 0 && c % 8 == 0 && f >= 2 && f % 2 == 0,
            "bad arguments (channels in multiples of 8, even stride)");
  float* part = nullptr;
  CVX_HIP(hipMalloc((void**)&part, (size_t)cvx_dwconvt_bwd_scratch_floats((long long)batch * ih * iw, c, f) * 4));
  int rc = cvx_dwconvt_bwd(dense(x_f16, ih * iw, c), dense(gout_f16, ih * f * iw * f, c), dense(gin_f16, ih * iw, c), weight, dweight, inv_scale, batch,
                           ih, iw, c, f, accumulate, part, (hipStream_t)hip_stream);
  (void)hipStreamSynchronize((hipStream_t)hip_stream);
  (void)hipFree(part);
  return rc;
}
extern "C" int cvx_avgpool_global_bwd_nhwc(const void* gout_f16, int32_t batch, int32_t hw, int32_t c, void* gin_f16, int32_t accumulate,
                                           void* hip_stream) {
  CVX_CHECK(gout_f16 && gin_f16 && batch > 0 && c % 8 == 0, "bad arguments (channels in multiples of 8)");
  CVX_TRY(cvx_avgpool_global_bwd(dense(gout_f16, 1, c), dense(gin_f16, hw, c), batch, hw, c, accumulate, (hipStream_t)hip_stream));
  CVX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  return 0;
}
extern "C" int cvx_resize_bilinear_bwd_nhwc(const void* gout_f16, int32_t batch, int32_t ih, int32_t iw, int32_t c, int32_t oh, int32_t ow,
                                            void* gin_f16, int32_t accumulate, void* hip_stream) {
  CVX_CHECK(gout_f16 && gin_f16 && batch > 0 && c % 8 == 0, "bad arguments (channels in multiples of 8)");
  CVX_TRY(cvx_resize_bilinear_bwd(dense(gout_f16, oh * ow, c), dense(gin_f16, ih * iw, c), batch, ih, iw, oh, ow, c, accumulate,
                                  (hipStream_t)hip_stream));
  CVX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  return 0;
}
extern "C" int cvx_dropout_nhwc(const void* x_f16, int32_t batch, int32_t hw, int32_t c, float p, uint64_t seed, void* out_f16, int32_t accumulate,
                                void* hip_stream) {
  CVX_CHECK(x_f16 && out_f16 && batch > 0 && c % 8 == 0, "bad arguments (channels in multiples of 8)");
  CVX_TRY(cvx_dropout(dense(x_f16, hw, c), dense(out_f16, hw, c), batch, hw, c, p, seed, accumulate, (hipStream_t)hip_stream));
  CVX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  return 0;
}
extern "C" int cvx_upsample2_nhwc(const void* x_f16, int32_t batch, int32_t h, int32_t w, int32_t c, void* out_f16, void* hip_stream) {
  CVX_CHECK(x_f16 && out_f16 && batch > 0 && c % 8 == 0, "bad arguments (channels in multiples of 8)");
  CVX_TRY(cvx_upsample2_fwd(dense(x_f16, h * w, c), dense(out_f16, 4 * h * w, c), batch, h, w, c, (hipStream_t)hip_stream));
  CVX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  return 0;
}
extern "C" int cvx_upsample2_bwd_nhwc(const void* gout_f16, int32_t batch, int32_t h, int32_t w, int32_t c, void* gin_f16, int32_t accumulate,
                                      void* hip_stream) {
  CVX_CHECK(gout_f16 && gin_f16 && batch > 0 && c % 8 == 0, "bad arguments (channels in multiples of 8)");
  CVX_TRY(cvx_upsample2_bwd(dense(gout_f16, 4 * h * w, c), dense(gin_f16, h * w, c), batch, h, w, c, accumulate, (hipStream_t)hip_stream));
  CVX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  return 0;
}

// ---- fp32 stem --------------------------------------------------------------------------------------------
extern "C" int cvx_stem_train_nchw(const float* images, int32_t batch, int32_t h, int32_t w, const float* weight, int32_t cout,
                                   const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                   float* running_var, void* out_f16, void* xhat_f16, float* mean, float* invstd, void* hip_stream) {
  CVX_CHECK(images && weight && gamma && beta && running_mean && running_var && out_f16 && xhat_f16 && mean && invstd, "null arguments");
  hipStream_t st = (hipStream_t)hip_stream;
  const StemParams sp{images, batch, h, w, h / 2, w / 2, weight, cout};
  Scratch slab;
  CVX_TRY(slab.alloc(slab_bytes(cout)));
  CVX_TRY(cvx_stem_stats(sp, (long long*)slab.p, st));
  BnTrainArgs ta{(const long long*)slab.p, gamma, beta, mean, invstd, running_mean, running_var, eps, momentum};
  CVX_TRY(cvx_stem_apply_train(sp, ta, dense(out_f16, (h / 2) * (w / 2), cout), (half_t*)xhat_f16, st));
  CVX_HIP(hipStreamSynchronize(st));
  return 0;
}
extern "C" int cvx_stem_eval_nchw(const float* images, int32_t batch, int32_t h, int32_t w, const float* weight, int32_t cout,
                                  const float* scale, const float* shift, void* out_f16, void* hip_stream) {
  CVX_CHECK(images && weight && scale && shift && out_f16, "null arguments");
  const StemParams sp{images, batch, h, w, h / 2, w / 2, weight, cout};
  CVX_TRY(cvx_stem_apply_eval(sp, scale, shift, dense(out_f16, (h / 2) * (w / 2), cout), (hipStream_t)hip_stream));
  CVX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  return 0;
}
extern "C" int cvx_stem_wgrad_nchw(const float* images, int32_t batch, int32_t h, int32_t w, const void* dy_f16, int32_t cout, float* dw,
                                   void* hip_stream) {
  CVX_CHECK(images && dy_f16 && dw, "null arguments");
  hipStream_t st = (hipStream_t)hip_stream;
  const StemParams sp{images, batch, h, w, h / 2, w / 2, dw /* unused by the gradient kernel, must be non-null */, cout};
  const long long M = (long long)batch * (h / 2) * (w / 2);
  const int ns = cvx_stem_wgrad_splits(M);
  Scratch slabs, dsd, dbl;
  CVX_TRY(slabs.alloc((size_t)ns * cout * 144 * 4));
  CVX_TRY(cvx_stem_wgrad(sp, (const half_t*)dy_f16, (float*)slabs.p, ns, st));
  // fold the split slabs with the engine's reducer: [Cout*9 rows][3 of 16 padded columns] -> dw [Cout][3][3][3]
  SlabDesc sd{0, 0, ns, cout * 9, 3, 16, cvx_slab_lanes(ns)};
  std::vector<BlockRef> blocks;
  const long long total = (long long)sd.rows * sd.Cin;
  for (long long s0 = 0; s0 < total; s0 += 256 / sd.lanes) blocks.push_back(BlockRef{0, (int)s0});
  CVX_TRY(dsd.alloc(sizeof(sd)));
  CVX_TRY(dbl.alloc(blocks.size() * sizeof(BlockRef)));
  CVX_HIP(hipMemcpy(dsd.p, &sd, sizeof(sd), hipMemcpyHostToDevice));
  CVX_HIP(hipMemcpy(dbl.p, blocks.data(), blocks.size() * sizeof(BlockRef), hipMemcpyHostToDevice));
  CVX_HIP(hipMemsetAsync(dw, 0, (size_t)cout * 27 * 4, st));
  CVX_TRY(cvx_reduce_slabs((const float*)slabs.p, dw, 1.0f, (const SlabDesc*)dsd.p, (const BlockRef*)dbl.p, (int)blocks.size(), st));
  CVX_HIP(hipStreamSynchronize(st));
  return 0;
}

// the engine's one-pass route with the recomputed xhat (include/cvx_engine.h)
extern "C" int cvx_stem_backward_recompute_nchw(const float* images, int32_t batch, int32_t h, int32_t w, const float* weight, const void* gout_f16, int32_t cout,
                                                const float* gamma, const float* beta, const float* mean, const float* invstd, float inv_scale,
                                                float* dgamma, float* dbeta, float* dw, void* hip_stream) {
  CVX_CHECK(images && weight && gout_f16 && gamma && beta && mean && invstd && dgamma && dbeta && dw, "null arguments");
  hipStream_t st = (hipStream_t)hip_stream;
  const StemParams sp{images, batch, h, w, h / 2, w / 2, weight, cout};
  const int hw = (h / 2) * (w / 2);
  const long long M = (long long)batch * hw;
  const int ns = cvx_stem_wgrad_splits(M);
  const ViewDesc g = dense(gout_f16, hw, cout);
  CVX_CHECK(!cvx_stem_keeps_xhat(sp, g, ns), "the one-pass stem backward does not take this shape (rows of 16-byte granularity, >= 2 pixel splits)");
  Scratch slabs;
  CVX_TRY(slabs.alloc((size_t)ns * cout * 144 * 4));
  BnCoef k{invstd, gamma, beta, mean};
  CVX_HIP(hipMemsetAsync(dw, 0, (size_t)cout * 27 * 4, st));
  // (the xhat argument only has to be a 16-byte aligned non-null pointer: it is not read)
  CVX_TRY(cvx_stem_backward_fold(sp, (const half_t*)slabs.p, g, k, nullptr, inv_scale, dgamma, dbeta, dw, (float*)slabs.p, ns, st));
  CVX_HIP(hipStreamSynchronize(st));
  return 0;
}

// the stem's whole backward pass (BatchNorm + SiLU backward and the weight gradient, fused as in the engine): gout = gradient
// w.r.t. the stem's activation, fp16 (B, h/2, w/2, cout); xhat / invstd from cvx_stem_train_nchw; dgamma / dbeta accumulated
extern "C" int cvx_stem_backward_nchw(const float* images, int32_t batch, int32_t h, int32_t w, const void* xhat_f16, const void* gout_f16,
                                      int32_t cout, const float* gamma, const float* beta, const float* invstd, float inv_scale, float* dgamma,
                                      float* dbeta, float* dw, void* hip_stream) {
  CVX_CHECK(images && xhat_f16 && gout_f16 && gamma && beta && invstd && dgamma && dbeta && dw, "null arguments");
  hipStream_t st = (hipStream_t)hip_stream;
  const StemParams sp{images, batch, h, w, h / 2, w / 2, dw, cout};
  const int hw = (h / 2) * (w / 2);
  const long long M = (long long)batch * hw;
  const int ns = cvx_stem_wgrad_splits(M);
  Scratch part, slabs, dsd, dbl;
  CVX_TRY(part.alloc(slab_bytes(cout)));
  CVX_TRY(slabs.alloc((size_t)ns * cout * 144 * 4));
  BnCoef k{invstd, gamma, beta};
  const ViewDesc g = dense(gout_f16, hw, cout);
  // the engine's sequence: the BatchNorm-backward sums only where the one-pass kernel does not take the shape, then kernel + fold
  if (!cvx_stem_backward_onepass_ok(sp, (const half_t*)xhat_f16, g, ns))
    CVX_TRY(cvx_bn_bwd_reduce((const half_t*)xhat_f16, M, cout, hw, k, g, BnActKind{0, 0, ViewDesc{nullptr, 0, 0}}, (long long*)part.p, st));
  CVX_HIP(hipMemsetAsync(dw, 0, (size_t)cout * 27 * 4, st));
  CVX_TRY(cvx_stem_backward_fold(sp, (const half_t*)xhat_f16, g, k, (const long long*)part.p, inv_scale, dgamma, dbeta, dw, (float*)slabs.p, ns, st));
  CVX_HIP(hipStreamSynchronize(st));
  return 0;
}
