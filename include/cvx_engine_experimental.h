/* Entry points that exist only in the TUNING build of the library (tools/build_tuning.sh: -DCVX_WITH_CHAIN): the tile-resident chain kernel
 * of round 3 (csrc/conv_chain.hip), which fuses a Bottleneck's two 3x3 convs or a whole Detect level into one launch.  It is parity-green
 * but measured 1-6 % SLOWER than the per-layer launches it replaces, so the release library does not carry it:
 * cvx_engine_set_fusion(e, 1) fails there with a message that says so. */
#pragma once
#include "cvx_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- tile-resident convolution chains (csrc/conv_chain.hip), single-op entry points ------------------------------------
 * Eval-mode fusion groups as ONE launch each: the intermediates stay in LDS, the weights of all stages stream through one
 * LDS-DMA ring.  x / out are NHWC fp16, weights [cout][kh][kw][cin] fp16, scale / shift the folded BatchNorm (fp32);
 * th x tw is the output tile one workgroup owns; the launch is repeated `reps` times and, when `elapsed_us` is not NULL, its mean
 * device time (HIP events, after one warm-up launch) is returned there (a tuning aid; the plan is built once per call).
 * cvx_chain_pair_unit:   out = [x +] silu(bn(conv3x3(silu(bn(conv3x3(x))))))      Replaces: Bottleneck.forward,
 *                        core/models/yolov8/modules.py:124-135 (eval mode).
 * cvx_chain_conv_unit:   out = act(bn(conv_kxk, stride s (nearest_up2(x) if upsample)))   Replaces: Conv.forward_fuse, modules.py:32-33
 *                        (and nn.Upsample feeding it, core/models/yolov8/yolo_v8.py:39-41).
 * cvx_chain_detect_unit: one Detect level's train-mode rows pred[b][a_off + pixel][0:64 | 64:64+ncp] (fp32) from its input feature map:
 *                        3x3 (cb + cc channels) -> 3x3 | 3x3 -> 1x1 | 1x1 + bias.  Replaces: Detect.forward's cv2[i] / cv3[i] branches,
 *                        modules.py:428-433. */
int cvx_chain_pair_unit(const void* x_f16, int32_t batch, int32_t h, int32_t w, int32_t c, const void* w1_f16, const float* scale1,
                        const float* shift1, const void* w2_f16, const float* scale2, const float* shift2, int32_t shortcut, void* out_f16,
                        int32_t th, int32_t tw, int32_t reps, float* elapsed_us, void* hip_stream);
int cvx_chain_conv_unit(const void* x_f16, int32_t batch, int32_t ih, int32_t iw, int32_t cin, const void* w_f16, int32_t cout, int32_t k,
                        int32_t stride, int32_t upsample, const float* scale, const float* shift, int32_t act, void* out_f16, int32_t th,
                        int32_t tw, int32_t reps, float* elapsed_us, void* hip_stream);
int cvx_chain_detect_unit(const void* x_f16, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cb, int32_t cc, int32_t ncp,
                          const void* wa_f16, const float* scale_a, const float* shift_a, const void* wb1_f16, const void* wb2_f16,
                          const float* scale_b, const float* shift_b, const void* wo1_f16, const void* wo2_f16, const float* bias, float* pred,
                          int32_t anchors, int32_t a_off, int32_t th, int32_t tw, int32_t reps, float* elapsed_us, void* hip_stream);

/* ---- weight-gradient timing aid (csrc/engine.hip, -DCVX_TUNING) ------------------------------------------------------------
 * Mean device time (us, HIP events, `reps` launches after one warm-up) of ONE weight-gradient launch of a k x k (1 or 3) / stride-1 / pad k/2
 * convolution on NHWC fp16 operands with pixel pitches x_ld / dy_ld; the fp32 slabs stay in `workspace` (no reduction).  nsplit <= 0: the
 * streaming kernel's own pixel-split count.  Replaces nothing in the reference: a tuning aid for the kernels behind autograd's weight
 * gradient of nn.Conv2d (core/models/yolov8/modules.py:19-33). */
int cvx_wgrad_time_unit(const void* x_f16, const void* dy_f16, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t k,
                        int32_t x_ld, int32_t dy_ld, int32_t nsplit, int32_t reps, void* workspace, int64_t workspace_bytes, float* us_out,
                        int32_t* nsplit_out, void* hip_stream);

#ifdef __cplusplus
}
#endif
