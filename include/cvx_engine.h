/* cvx_engine.h -- C ABI of the MI355X (gfx950) detection engine, libcvx_engine.so.
 *
 * The reference (calmiLovesAI/ComputerVision.pytorch) has no C/FFI boundary: its hot path is
 * torch.nn / ATen calls made from Python.  This header is the boundary a maintainer binds instead
 * (ctypes stub in INTEGRATION.md); each entry point names the reference interface it replaces,
 * as file:line under the reference tree.
 *
 * Conventions: plain pointers and sizes only (no torch types); every function returns 0 on success
 * and -1 on failure with the message available from cvx_last_error(); all device pointers are HIP
 * device memory owned by the caller unless stated; work is enqueued on the hipStream_t passed at
 * creation (stream-ordered, no host synchronisation inside).  One engine per device, externally
 * synchronised.  Activations are NHWC fp16, accumulation fp32, parameters/gradients fp32.
 */
#ifndef CVX_ENGINE_H
#define CVX_ENGINE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CVX_ABI_VERSION 4

const char* cvx_last_error(void);
int cvx_abi_version(void);

/* ---- graph description (built by the Python mirror of core/models/yolov8/yolo_v8.py:16-107) ---- */
enum { CVX_BUF_ACT_F16 = 0, CVX_BUF_PRED_F32 = 1 };
typedef struct {
  int32_t h, w, c; /* per image; PRED buffers: h*w = anchors, c = no (=nc+64) */
  int32_t kind;
} cvx_buf_desc;

typedef struct {
  int32_t buf;     /* index into the buffer table, -1 = none */
  int32_t coff;    /* first channel of the slice */
  int32_t c;       /* channels in the slice */
  int32_t pix_off; /* first pixel row (PRED buffers: anchor offset of the level), else 0 */
} cvx_view;

enum { CVX_OP_CONV = 1, CVX_OP_MAXPOOL5 = 2, CVX_OP_UPSAMPLE2 = 3,
       /* DLA-34 / CenterNet ops (core/models/centernet_model.py), forward and backward: */
       CVX_OP_MAXPOOL2 = 4, /* 2x2 stride-2 max pool (Tree.downsample, :128-129) */
       CVX_OP_DWCONVT = 5,  /* depthwise ConvTranspose2d, kernel 2*stride, padding stride/2 (IDAUp.up_i, :256); w_off -> fp32 [C][2f][2f] */
       CVX_OP_COPY = 6,     /* channel-slice copy (a tensor that lives in two concat buffers) */
       /* DeepLabv3+ / ResNet ops (core/models/deeplabv3plus.py, core/models/resnet.py), forward and backward: */
       CVX_OP_MAXPOOL3S2 = 7, /* 3x3 stride-2 pad-1 max pool (resnet.py:163) */
       CVX_OP_AVGPOOL = 8,    /* global average pool -> (B, 1, 1, C) (ASPPPooling, deeplabv3plus.py:30) */
       CVX_OP_RESIZE = 9,     /* bilinear resize (ih, iw) -> (oh, ow), align_corners = False (deeplabv3plus.py:38,117-122) */
       /* SSD / VGG ops (core/models/ssd_model.py), forward and backward: */
       CVX_OP_MAXPOOL3S1 = 10, /* 3x3 stride-1 pad-1 max pool (VGG pool5, :30) */
       CVX_OP_L2NORM = 11,     /* x / (||x||_2 over channels + 1e-10) * weight[c] (L2Normalize, :113-128); gamma_off -> weight (C floats) */
       CVX_OP_DROPOUT = 12 };  /* nn.Dropout (deeplabv3plus.py:67): k = drop probability in units of 2^-16; identity in eval mode.
                                  Training: inverted dropout, the mask a counter-based hash of (cvx_engine_set_seed, training pass, op,
                                  element) -- a different draw than torch's Philox stream, the same distribution */
/* CVX_OP_MAXPOOL2 also serves ceil_mode = True (:18): oh = ceil(ih / 2), windows clipped at the border. */
enum { CVX_ACT_BN_SILU = 1, CVX_ACT_BIAS = 2,
       CVX_ACT_BN_RELU = 3,   /* Conv + BN + ReLU (trainable: batch statistics, backward) */
       CVX_ACT_BN_LINEAR = 4, /* Conv + BN (Tree.project, ResNet downsample; trainable) */
       /* epilogues without BatchNorm (trainable: dy = g * [out > 0] / g, bias gradient = column sums): */
       CVX_ACT_BIAS_RELU = 5,   /* Conv + bias + ReLU, fp16 output (head 3x3, :314-318) */
       CVX_ACT_BIAS_LINEAR = 6 }; /* Conv + bias, fp16 output, no activation (SSD ExtraLayer, ssd_model.py:90-110) */
#define CVX_OPF_RES_PRE_ACT 1 /* cvx_op_desc.flags: the residual is added before the activation (BasicBlock, :20-27) */
#define CVX_OPF_CONV_BIAS 2   /* a BN_* conv that also has a bias of its own at bias_off (VGG-BN: Conv2d(bias=True) + BatchNorm) */
#define CVX_OPF_RAW_F16 4     /* training: a BN_SILU conv (no residual, no bias) may keep its raw output in fp16 between the convolution and its
                                 normalisation pass -- and keeps THAT, not the normalised value, for the backward pass: 6 instead of 12 bytes
                                 per element around the forward BatchNorm.  The batch statistics still come from the fp32 accumulators.  For
                                 layers whose rounding the outputs do not see (YOLOv8: the neck and the head, modules 12 .. 22) */

typedef struct {
  int32_t type;
  cvx_view in, out, res;  /* res: residual added after the activation (Bottleneck shortcut) */
  int32_t ih, iw, oh, ow; /* spatial size of the input / output views */
  int32_t k, stride, pad, dil;
  int32_t act;            /* CONV: CVX_ACT_BN_SILU (modules.py:29-30) or CVX_ACT_BIAS (modules.py:424-425 last layer) */
  int32_t needs_dgrad;    /* 0 for the stem (input is the image) */
  int32_t w_cin;          /* input channels of the stored weight tensor (stem: 3, its view is zero-padded to 8) */
  /* element offsets into the caller's flat fp32 arenas; weights are stored [cout][kh][kw][cin] */
  int64_t w_off;
  int64_t gamma_off, beta_off; /* BN affine (param arena) */
  int64_t bias_off;            /* CVX_ACT_BIAS */
  int64_t rmean_off, rvar_off; /* BN running statistics (stats arena) */
  int32_t lane; /* 0 / 1: the main chain.  >= 2: an independent tail (Detect levels 1, 2) the engine MAY run on its high-priority lane
                   stream beside the main chain, forked at the first such op and joined after the last (a hint: results do not depend on it) */
  int32_t flags; /* CVX_OPF_* */
} cvx_op_desc;

typedef struct cvx_engine cvx_engine;

/* Every op kind and epilogue has a backward pass except: ReLU with a POST-activation residual, and CVX_ACT_BIAS_RELU / _BIAS_LINEAR with a
 * residual.  A graph that contains one of those, or none of whose convolutions asks for a data gradient, can only run
 * cvx_engine_forward(training = 0).  The op that reads `image_buf` may be the YOLO stem (below) or any
 * convolution with 3 stored input channels and needs_dgrad = 0 (the image is converted to NHWC fp16, 8 channels; its weight gradient
 * reads that copy).
 *
 * Builds an engine for a fixed input size.  `image_buf` is the index of the buffer-table entry (c == 8) that stands for
 * the caller's NCHW fp32 images; exactly one op may read it: the 3 -> 16..80 channel 3x3 stride-2 BN+SiLU stem, which runs
 * in fp32 straight from the caller's tensor (no fp16 copy of the image is made).
 * Replaces: Yolo8.__init__ graph construction, core/models/yolov8/yolo_v8.py:17-62.  *
 * Threading: the auxiliary HIP streams (weight gradients, lanes) are ONE set per device shared by all engines of the process (the
 * hardware-queue budget).  All engines of a device must therefore be driven from one host thread (or be externally
 * serialised), and while a hipGraph capture of one engine's step is open no other engine of that device may launch: its kernels would
 * land on the shared streams that the capture has pulled into capture mode. */
int cvx_engine_create(cvx_engine** out, const cvx_buf_desc* bufs, int32_t nbufs, const cvx_op_desc* ops, int32_t nops,
                      int32_t image_buf, int32_t pred_buf, int32_t device, void* hip_stream);
int cvx_engine_destroy(cvx_engine* e);

/* Binds the caller-owned flat arenas (fp32): parameters, gradients (same layout) and BN running
 * statistics.  Replaces: nn.Module parameter/buffer storage (state_dict tensors are views of these). */
int cvx_engine_bind(cvx_engine* e, float* params, float* grads, int64_t n_params, float* stats, int64_t n_stats);

/* Changes the HIP stream subsequent calls enqueue on (e.g. a stream that is being captured into a hipGraph). */
int cvx_engine_set_stream(cvx_engine* e, void* hip_stream);
/* The stream the data-parallel gradient exchange is to be queued on (hipStream_t): the engine's own weight-gradient stream (lowest
 * priority, shared by all engines of the process).  cvx_engine_grads_ready / cvx_engine_backward_exchange fold a range's weight-gradient
 * slabs and queue its all-reduce there, behind the weight gradients they depend on.  The engine works three hardware queues (the caller's
 * stream, this one, the Detect lanes); a process that puts MORE than four to work pays 2.2-2.5x on every train step on this runtime
 * (6.5 -> 16 ms with the exchange on a stream of its own beside an engine that then had three) -- so do not create a
 * stream for the exchange: the fourth queue is RCCL's internal stream's, or the caller's launch stream's.
 * Replaces: the side stream DDP's reducer owns (torch/nn/parallel/distributed.py). */
void* cvx_engine_exchange_stream(cvx_engine* e);

/* BatchNorm hyper-parameters (core/models/yolov8/torch_utils.py:17-19: eps 1e-3, momentum 0.03). */
int cvx_engine_set_bn(cvx_engine* e, float eps, float momentum);
/* Eval-mode cross-layer fusion: a Bottleneck's 3x3 -> 3x3 (+ shortcut) and a whole Detect level as ONE tile-resident launch each
 * (csrc/conv_chain.hip).  Parity-green but measured 1-6 % slower than the per-layer kernels at batch 32, so since round 4 the kernel is part
 * of the TUNING build only (tools/build_tuning.sh, -DCVX_WITH_CHAIN; include/cvx_engine_experimental.h): in the release library
 * cvx_engine_set_fusion(e, 1) FAILS with an error that says so (enable = 0 is accepted) and cvx_engine_fused_groups returns 0.
 * Replaces: the module-by-module execution of Bottleneck.forward / Detect.forward, core/models/yolov8/modules.py:124-135, 428-433. */
/* Inference with constant weights: the caller vouches that no parameter of the bound arena changed since this engine's PREVIOUS
 * cvx_engine_forward.  The next forward (only that one) then skips the fp32 -> fp16 weight conversion and the kernels' packed weight
 * images (40 us of a 1.4-ms YOLOv8-n forward, 90-220 us of the bigger models') -- provided they were made under the current batch plan and
 * by a forward of the same mode, or a training one (an eval forward does not prepare the data-gradient images); otherwise the call has no effect.
 * BatchNorm folding (running statistics -> scale / shift) is never skipped.  The reference has no counterpart: torch converts nothing. */
int cvx_engine_keep_shadows(cvx_engine* e);
int cvx_engine_set_fusion(cvx_engine* e, int32_t enable);
int32_t cvx_engine_fused_groups(const cvx_engine* e);

/* Seed of the CVX_OP_DROPOUT masks (default 0).  Replaces: torch.manual_seed for the model's nn.Dropout layers. */
int cvx_engine_set_seed(cvx_engine* e, uint64_t seed);

/* Forward pass.  images: (B,3,H,W) fp32 NCHW in [0,1], 8-byte aligned; pred: (B, A, no) fp32, anchors of the three
 * levels concatenated (80x80, 40x40, 20x20 order), channels = [64 DFL logits | nc class logits].
 * training != 0: batch statistics + running-stat update, activations kept for backward; `images` must stay valid and
 * unchanged until the backward pass of this forward has been enqueued AND has finished (the stem's weight gradient
 * reads it).
 * Replaces: Yolo8.forward, core/models/yolov8/yolo_v8.py:78-107 (+ Conv/C2f/SPPF/Detect in modules.py). */
int cvx_engine_forward(cvx_engine* e, const float* images, int32_t batch, int32_t training, float* pred);

/* Backward pass of the last training forward.  dpred: (B, A, no) fp16 = loss_scale * dLoss/dpred.
 * Parameter gradients are ACCUMULATED (+=, unscaled by 1/loss_scale) into the bound gradient arena.
 * Replaces: loss.backward() through the model, core/trainer/yolo8_train.py:103,108. */
int cvx_engine_backward(cvx_engine* e, const void* dpred_f16, float loss_scale);

/* Debug/inspection: copies activation (which=0) or gradient (which=1) buffer `buf` of the last planned
 * batch, NHWC fp16, into dst (device or host memory, `bytes` must equal batch*h*w*c*2).  which=2 / 3: `buf` is the index
 * of a conv op and the tensor is its normalised output xhat = (y-mean)*invstd / the gradient w.r.t. its raw output,
 * dense (batch, oh, ow, cout) fp16 -- the per-layer operands of the backward pass (training plans only).   which=4: the normalised output as the backward passes USE it, fp32 (bytes = batch*h*w*c*4): the kept fp16 xhat widened, or --
 * for a CVX_OPF_RAW_F16 layer, which keeps its raw fp16 output instead -- (y - mean) * invstd computed as they compute it. */
int cvx_engine_debug_copy(cvx_engine* e, int32_t buf, int32_t which, void* dst, int64_t bytes);

/* Segmented backward for data-parallel training: the same pass as cvx_engine_backward, cut into op ranges so that the
 * caller can exchange the gradients of finished parameter ranges (RCCL all-reduce on its own stream) while the rest of the
 * pass still runs.  Protocol: begin; then for ranges that tile the op list from the last op down to op 0:
 * range(op_hi, op_lo) followed at any later point by grads_ready(op_hi, op_lo, stream) -- `stream` is made to wait for
 * the range's BN/bias gradients (main stream) and weight gradients (side stream) and then folds the range's
 * weight-gradient slabs into the gradient arena ON `stream`; after it the parameter range of those ops is final there;
 * end() joins the side stream.  The caller makes the engine's stream wait for `stream` before the optimiser step.
 * Replaces: loss.backward() + DistributedDataParallel's bucketed all-reduce hooks (the reference trains single-GPU,
 * core/trainer/yolo8_train.py:93-111; BASELINE.json north_star asks for the overlapped exchange). */
int cvx_engine_backward_begin(cvx_engine* e, const void* dpred_f16, float loss_scale);
int cvx_engine_backward_range(cvx_engine* e, int32_t op_hi, int32_t op_lo);
int cvx_engine_grads_ready(cvx_engine* e, int32_t op_hi, int32_t op_lo, void* hip_stream);
int cvx_engine_backward_end(cvx_engine* e);

/* Per-kernel-class timing with HIP events recorded on the engine's launch stream.  Classes: 0 conv forward
 * (implicit GEMM), 1 conv data-gradient, 2 conv weight-gradient, 3 BN+SiLU forward passes, 4 BN+SiLU backward
 * passes, 5 misc (layout, pool, upsample, weight shadows, bias sums), 6 gradient-slab reduction.
 * cvx_engine_profile(e, 1) starts a window; cvx_engine_profile_read synchronises, returns the summed kernel
 * time (ms), algorithmic FLOPs, algorithmic bytes and launch count per class since the window start, and
 * restarts the window. */
int cvx_engine_profile(cvx_engine* e, int32_t enable);
int cvx_engine_profile_read(cvx_engine* e, int32_t n_classes, double* ms, double* flops, double* bytes, int64_t* launches);
/* Same window, per record instead of per class: writes one CSV line "class,op,flops,bytes,ms" per timed scope to
 * `path` (op = index into the op list, -1 for whole-pass scopes) and restarts the window.  Tuning aid. */
int cvx_engine_profile_dump(cvx_engine* e, const char* path);

/* Tuning aid: while `buf` (device memory, 8 x u64 per workgroup of the largest launch) is set, thread 0 of every
 * workgroup of the DMA-ring / halo convolution kernels stores 100 MHz wall-clock stamps at five phases
 * (entry, operands issued, first K-step landed, K loop done, epilogue done).  NULL switches it off. */
int cvx_debug_clock_buffer(void* buf);

/* Tuning aid / test hook: the tiling the row-band convolution kernel (csrc/conv_tile.hip: 3x3, stride 1, pad 1 on small maps) would
 * choose for a launch -- out[0..7] = rows per tile, pixel groups per wave (MT), 16-channel tiles per workgroup (NTW), channel blocks,
 * workgroups, K-steps per weight chunk, LDS bytes, estimated microseconds x 100.  Returns 0, or -1 when the kernel does not take the
 * shape.  (No reference counterpart: the reference leaves the choice of algorithm to cuDNN behind nn.Conv2d, core/models/yolov8/modules.py:19-33.) */
int cvx_debug_conv_tile_plan(int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t* out8);

/* Bytes of device memory the engine currently owns (workspaces). */
int64_t cvx_engine_workspace_bytes(const cvx_engine* e);
/* Counts re-plans: a forward with a different batch size (or the first training forward after eval-only ones) frees and
 * re-allocates every per-batch buffer.  A caller that captured launches into a hipGraph must drop the graph when this
 * number changes -- the graph holds the old buffer addresses. */
int64_t cvx_engine_plan_generation(const cvx_engine* e);

/* Copies one level of pred into an NCHW fp32 tensor (B, no, H, W) -- the reference's output format
 * (core/models/yolov8/modules.py:431-433) -- and the inverse for incoming NCHW gradients. */
int cvx_pred_level_to_nchw(const float* pred, int32_t batch, int32_t anchors, int32_t no, int32_t a_off, int32_t h, int32_t w,
                           float* out_nchw, void* hip_stream);
int cvx_nchw_grad_to_dpred(const float* grad_nchw, int32_t batch, int32_t anchors, int32_t no, int32_t a_off, int32_t h, int32_t w,
                           float scale, void* dpred_f16, void* hip_stream);

/* ---- v8 detection loss: TaskAlignedAssigner + BCE + CIoU + DFL, forward AND gradient --------------
 * pred (B,A,no) fp32; targets: n_targets rows [batch_idx, cls, cx, cy, w, h] (normalised, the
 * yolo8_collate dict flattened, core/data/collate.py:25-29); level_hw: 3 x (h, w); strides: 3.
 * Outputs: loss_items[3] = (box, cls, dfl) * gains (device fp32), dpred (B,A,no) fp16 =
 * loss_scale * d(sum(items) * B)/dpred.  workspace: cvx_loss_v8_workspace_bytes() bytes.
 * Replaces: Loss.__call__, core/algorithms/yolo_v8.py:75-124; TaskAlignedAssigner, core/utils/bboxes.py:275-470;
 * BboxLoss, core/loss/ultralytics_loss.py:25-57; bbox_iou, core/utils/ultralytics_iou.py:64-117. */
int64_t cvx_loss_v8_workspace_bytes(int32_t batch, int32_t anchors, int32_t nc, int32_t max_targets_per_image);
int cvx_loss_v8(const float* pred, int32_t batch, int32_t anchors, int32_t nc, const float* targets, int32_t n_targets,
                int32_t max_targets_per_image, const int32_t* level_hw, const float* strides, int32_t n_levels, float gain_box,
                float gain_cls, float gain_dfl, float loss_scale, float* loss_items, void* dpred_f16, void* workspace,
                int64_t workspace_bytes, void* hip_stream);
/* Same with an explicit row pitch (multiple of 4, >= 64 + nc) of pred and dpred; columns beyond 64 + nc are neither read
 * nor written (the caller keeps the padding of dpred zero). */
int cvx_loss_v8_strided(const float* pred, int32_t pred_ld, int32_t batch, int32_t anchors, int32_t nc, const float* targets,
                        int32_t n_targets, int32_t max_targets_per_image, const int32_t* level_hw, const float* strides, int32_t n_levels,
                        float gain_box, float gain_cls, float gain_dfl, float loss_scale, float* loss_items, void* dpred_f16,
                        void* workspace, int64_t workspace_bytes, void* hip_stream);

/* Per-anchor assignment of the last cvx_loss_v8* call made with this workspace and the same (batch, anchors,
 * max_targets): index of the assigned target row (-1: background) and its normalised target score (device arrays of
 * batch*anchors).  Replaces: reading TaskAlignedAssigner's target_gt_idx / fg_mask / target_scores, bboxes.py:330-345. */
int cvx_loss_v8_assignment(const void* workspace, int32_t batch, int32_t anchors, int32_t max_targets_per_image, int32_t* gt_index,
                           float* norm_score, void* hip_stream);
/* yolo8_collate's dict (core/data/collate.py:25-29; fp32 device arrays batch_idx (n), cls (n), bboxes (n,4)) -> the
 * (n,6) target rows cvx_loss_v8 reads, grouped by image in stable order.  Replaces: Loss.preprocess, yolo_v8.py:51-65. */
int cvx_pack_targets(const float* batch_idx, const float* cls, const float* bboxes, int32_t n, float* rows, void* hip_stream);

/* ---- input side: letterbox pre-processing -------------------------------
 * cvx_letterbox_u8_to_nchw: uint8 (h, w, 3) HWC image in DEVICE memory -> one (3, H, W) fp32 image of the network's input batch
 * (out_chw = batch tensor + b*3*H*W), values in [0, 1].  letterbox != 0: the reference's letter_box -- aspect-preserving
 * cv2.resize(INTER_NEAREST) to (int(h*s), int(w*s)), s = min(H/h, W/w), centred on a (128,128,128) canvas; letterbox == 0: plain
 * nearest resize to (H, W) (the reference uses INTER_CUBIC there: not built).  swap_rb: BGR source -> RGB planes (cv2.cvtColor).
 * Asynchronous on hip_stream.  cvx_letterbox_geometry: the same size / padding arithmetic on the host (the decode side needs it).
 * Replaces: letter_box + TF.to_tensor in read_image_and_convert_to_tensor, core/utils/image_process.py:29-66 (image file I/O stays
 * with the caller). */
int cvx_letterbox_geometry(int32_t h, int32_t w, int32_t H, int32_t W, int32_t* new_h, int32_t* new_w, int32_t* top, int32_t* left, double* scale);
int cvx_letterbox_u8_to_nchw(const uint8_t* image_hwc, int32_t h, int32_t w, int32_t letterbox, int32_t swap_rb, float* out_chw, int32_t H,
                             int32_t W, void* hip_stream);

/* ---- optimiser ------------------------------------------------------------------------------------
 * torch.optim.Adam semantics (core/trainer/lr_scheduler.py:37-43): lr, betas, eps, no weight decay;
 * `step` counts from 1.  found_inf (device int32, may be NULL): when non-zero the update is skipped
 * (GradScaler.step semantics, core/trainer/yolo8_train.py:104).  zero_grad: clear g afterwards. */
int cvx_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                  float eps, int32_t step, const int32_t* found_inf, int32_t zero_grad, void* hip_stream);
int cvx_check_finite(const float* grads, int64_t n, int32_t* found_inf, void* hip_stream);
/* Same update with the step state resident on the device -- state[0] = lr (written by the host when the schedule
 * changes it), state[1] = step count, state[2..3] = derived bias-correction factors, advanced by the call itself --
 * so a captured hipGraph of the whole train step can be replayed without changing any kernel argument.
 * grad_scale multiplies every gradient first (1/world_size after a SUM all-reduce of the arena; 1 otherwise). */
int cvx_adam_step_dev(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1, float beta2, float eps,
                      float* state, const int32_t* found_inf, int32_t zero_grad, float grad_scale, void* hip_stream);

/* ---- eval tail: DFL decode + sigmoid, then class-aware NMS ---------------------------------------
 * cvx_decode: pred (B,A,no) -> y (B, 4+nc, A) fp32 [cx,cy,w,h (pixels), class scores]
 *   Replaces: Detect eval branch, core/models/yolov8/modules.py:434-446.
 * cvx_nms: y -> per image up to max_det (1 .. 16384) rows [x1,y1,x2,y2,conf,cls] + the anchor index of each row;
 *   counts[b] = rows kept (-1: more candidates than the 16384 the in-LDS sort holds -- raise conf_thres).  Semantics = oracle/nms_ref.py (torchvision 0.14.1 batched_nms restated, both strategies).
 *   Replaces: non_max_suppression, core/utils/ultralytics_ops.py:131-264. */
int cvx_decode(const float* pred, int32_t batch, int32_t anchors, int32_t nc, const int32_t* level_hw, const float* strides,
               int32_t n_levels, float* y, void* hip_stream);
/* Same with an explicit row pitch of pred (>= 64 + nc): class counts that are not multiples of 8 (VOC: 20) run with the
 * class columns padded to the next multiple of 8; the padding is never read. */
int cvx_decode_strided(const float* pred, int32_t pred_ld, int32_t batch, int32_t anchors, int32_t nc, const int32_t* level_hw,
                       const float* strides, int32_t n_levels, float* y, void* hip_stream);
/* SSD decode: loc (batch, A, 4) regressions and conf (batch, A, nc+1) logits in the reference's output format, priors (A, 4)
 * corner boxes -> boxes (batch, A, 4) clipped corners, prob (batch, A, nc+1) softmax.
 * Replaces: torch.softmax + Ssd._parse_mbox_loc, core/algorithms/ssd.py:246-247,285-325. */
int cvx_ssd_decode(const float* loc, const float* conf, const float* priors, int32_t batch, int32_t anchors, int32_t num_classes_plus_bg,
                   float variance_xy, float variance_wh, float* boxes, float* prob, void* hip_stream);
/* cvx_ssd_decode that also returns, per score column, the largest probability of the whole batch (class_max: num_classes_plus_bg floats, may be
 * NULL): the caller's per-class loop (ssd.py:246-274) skips the classes nothing passes the threshold in after ONE host read. */
int cvx_ssd_decode_max(const float* loc, const float* conf, const float* priors, int32_t B, int32_t A, int32_t num_classes_plus_bg,
                       float variance_xy, float variance_wh, float* boxes, float* prob, float* class_max, void* hip_stream);
/* Column range [col0, col0 + c) of fp32 rows (batch, anchors, ld), pixels a_off .. a_off + h*w, as an NCHW block written at
 * out[b * out_bstride + out_off + ch * h*w + pix]: SSD flattens its head maps in NCHW order and concatenates the levels
 * (core/models/ssd_model.py:177-183).  Asynchronous on hip_stream. */
int cvx_pred_cols_to_nchw(const float* rows, int32_t ld, int32_t col0, int32_t c, int32_t batch, int32_t anchors, int32_t a_off, int32_t hw,
                          float* out, int64_t out_bstride, int64_t out_off, void* hip_stream);
/* YOLOv7 anchor decode.  pred: fp32 rows (batch, sum_l h_l*w_l, pred_ld) as the engine's YOLOv7 graph writes them (one row per
 * pixel, levels in the order of the network's outputs, columns a*(5+nc)+k for anchor a); anchors_wh: n_levels x 3 x (w, h) in
 * input pixels, already selected through anchors_mask.  dec: (batch, 3*sum, 5+nc) = the reference's `decoded_outputs`
 * (normalised cx, cy, w, h, objectness, class probabilities; level, anchor, pixel order).  y (optional): the same candidates as
 * (batch, 4+nc, 3*sum) channel-major [cx, cy, w, h, objectness*class_k] -- what cvx_nms_variant(CVX_NMS_VANILLA) takes, i.e. the
 * per-class greedy NMS of YOLOv7._nms with score = objectness * best class probability.
 * Replaces: YOLOv7.decode_box, core/algorithms/yolo_v7.py:234-346. */
int cvx_yolo7_decode(const float* pred, int32_t pred_ld, int32_t batch, int32_t nc, const int32_t* level_hw, const float* anchors_wh,
                     int32_t n_levels, int32_t input_h, int32_t input_w, float* dec, float* y, void* hip_stream);
enum { CVX_NMS_TV0141_CUDA = 0, /* torchvision 0.14.1's own switch for CUDA tensors: coordinate trick up to 5000 candidates */
       CVX_NMS_TV0141_CPU = 1,  /* ... for CPU tensors: up to 1000 candidates */
       CVX_NMS_OFFSET = 2,      /* _batched_nms_coordinate_trick: boxes + cls*(max+1), one class-agnostic pass */
       CVX_NMS_VANILLA = 3 };   /* _batched_nms_vanilla: per-class passes on the unshifted boxes */
#define CVX_NMS_BOXES_XYXY 0x100 /* OR-ed into `variant`: rows 0..3 of y are corner boxes already (no cx,cy,w,h conversion) */
int64_t cvx_nms_workspace_bytes(int32_t batch, int32_t anchors);
int cvx_nms(const float* y, int32_t batch, int32_t anchors, int32_t nc, float conf_thres, float iou_thres, int32_t max_det,
            float* out_rows, int32_t* out_index, int32_t* counts, void* workspace, int64_t workspace_bytes, void* hip_stream);
/* cvx_nms = cvx_nms_variant(CVX_NMS_TV0141_CUDA): what the reference's GPU predict path executes through
 * torchvision.ops.batched_nms (core/utils/ultralytics_ops.py:247). */
int cvx_nms_variant(const float* y, int32_t batch, int32_t anchors, int32_t nc, float conf_thres, float iou_thres, int32_t max_det,
                    int32_t variant, float* out_rows, int32_t* out_index, int32_t* counts, void* workspace, int64_t workspace_bytes,
                    void* hip_stream);

/* ---- CenterNet heat-map decode (BASELINE.json configs[3]) -----------------------------------------------------
 * pred: (B, H*W, pred_ld) fp32 head tensor, heat-map logits in columns [0, nc), the two channels the reference reads as
 * centre offsets at reg_col (its "wh" head), the two it reads as sizes at wh_col (its "reg" head).  Per image: sigmoid,
 * the reference's 3x3 max-pool over (x, class), top-K (score desc, flat index asc), boxes = clamp([x+dx, y+dy, w, h] / (W, H))
 * as xyxy in [0, 1], mask score >= conf, class-agnostic greedy DIoU-NMS (a box survives a kept one when DIoU <= nms_thr).
 * Outputs: boxes (B,K,4), scores (B,K), classes (B,K), topk_index (B,K) = (y*W+x)*nc+c (-1: fewer than K peaks) for the whole
 * top-K list; keep (B,K) = list positions of the survivors in descending score; counts (B) = survivors (-1: more than 2048
 * scores tie at the K-th value).  The letterbox inverse (a few scalars per image) stays with the caller.
 * Replaces: CenterNetA.decode_boxes, core/algorithms/centernet.py:271-338; diou_nms, core/utils/nms.py:9-31. */
int64_t cvx_centernet_decode_workspace_bytes(int32_t batch, int32_t h, int32_t w, int32_t nc);
int cvx_centernet_decode(const float* pred, int32_t pred_ld, int32_t batch, int32_t h, int32_t w, int32_t nc, int32_t reg_col, int32_t wh_col,
                         int32_t k, float conf, float nms_thr, int32_t use_nms, float* boxes, float* scores, int32_t* classes,
                         int32_t* topk_index, int32_t* keep, int32_t* counts, void* workspace, int64_t workspace_bytes, void* hip_stream);

/* ---- single-op entry points (unit tests and other model families reuse them) -----------------------
 * NHWC fp16 convolution, weights [cout][kh][kw][cin] fp16.  mode 0: out fp16 = conv; mode 1: out fp16 =
 * silu(conv*scale+shift); mode 2: out fp32 = conv + bias; mode 3 (the training epilogue): out fp32 = conv, and the
 * per-channel (sum, sum of squares) are added into `shift` viewed as zeroed fixed-point replica slabs.  mode | 0x100 runs the GEMM-shaped
 * kernel (conv_gemm.hip) whatever the dispatcher would pick (needs cin % 8 == 0, cout % 4 == 0).  Replaces: F.conv2d as used by
 * Conv.forward, core/models/yolov8/modules.py:29-30. */
int cvx_conv2d_nhwc(const void* x_f16, int32_t batch, int32_t ih, int32_t iw, int32_t cin, const void* w_f16, int32_t cout, int32_t k,
                    int32_t stride, int32_t pad, int32_t dil, int32_t mode, const float* scale_or_bias, const float* shift, void* out,
                    void* hip_stream);
/* data gradient: dx (B,ih,iw,cin) fp16 from dy (B,oh,ow,cout) fp16 and w_t [cin][kh][kw][cout] fp16 */
int cvx_conv2d_dgrad_nhwc(const void* dy_f16, int32_t batch, int32_t ih, int32_t iw, int32_t cin, const void* wt_f16, int32_t cout,
                          int32_t k, int32_t stride, int32_t pad, int32_t dil, void* dx_f16, void* hip_stream);
/* weight gradient: dw [cout][kh][kw][cin] fp32 (overwritten) from x and dy; workspace >= cvx_conv2d_wgrad_workspace_bytes */
int64_t cvx_conv2d_wgrad_workspace_bytes(int32_t batch, int32_t oh, int32_t ow, int32_t cin, int32_t cout, int32_t k);
int cvx_conv2d_wgrad_nhwc(const void* x_f16, const void* dy_f16, int32_t batch, int32_t ih, int32_t iw, int32_t cin, int32_t cout,
                          int32_t k, int32_t stride, int32_t pad, int32_t dil, float* dw, void* workspace, int64_t workspace_bytes,
                          void* hip_stream);


/* ---- streaming ops on dense NHWC fp16 tensors (the kernels the engine runs between the convolutions) ----------------
 * Train-mode BatchNorm + SiLU.  y: raw conv output FP32 (B*hw, C); statistics are taken from it, running statistics
 * updated (momentum, unbiased variance); out = silu(gamma*xhat+beta) (+res) fp16, xhat = (y-mean)*invstd fp16 (the
 * operand of the backward op), mean / invstd (C) returned.  C multiple of 8, <= 2048.
 * Replaces: Conv.forward's bn + act, core/models/yolov8/modules.py:29-30 (eps / momentum: torch_utils.py:17-19). */
int cvx_bn_silu_train_nhwc(const float* y_f32, int32_t batch, int32_t hw, int32_t c, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, const void* res_f16, void* out_f16, void* xhat_f16,
                           float* mean, float* invstd, void* hip_stream);
/* Its backward: dy (gradient w.r.t. the raw conv output) from gout (gradient w.r.t. out); dgamma / dbeta are ACCUMULATED
 * (scaled by inv_scale); gres (optional) receives the residual branch's gradient = gout (+= when res_accumulate). */
int cvx_bn_silu_bwd_nhwc(const void* xhat_f16, const void* gout_f16, int32_t batch, int32_t hw, int32_t c, const float* gamma,
                         const float* beta, const float* invstd, float inv_scale, float* dgamma, float* dbeta, void* dy_f16, void* gres_f16,
                         int32_t res_accumulate, void* hip_stream);
/* The same two passes for the other Conv + BatchNorm blocks of the reference (C up to 2048).  act: 0 SiLU, 1 ReLU, 2 none.
 * res_pre = 1: the residual joins the PRE-activation, out = act(gamma*xhat + beta + res) (Bottleneck.forward, resnet.py:139-141:
 * out += identity; out = relu(out)) and gres receives the pre-activation gradient dz; res_pre = 0: out = act(.) + res, gres
 * receives gout.  ReLU's backward mask [out > 0] is kept by the forward pass in the lowest mantissa bit of xhat (so xhat of a ReLU layer is
 * accurate to one fp16 ulp instead of half); SiLU with a pre-activation residual (YOLOv7 RepConv: silu(bn(conv3x3) + bn(conv1x1)),
 * yolov7_model.py:250-262) needs the residual's forward VALUE in `out_f16`; otherwise out_f16 may be NULL.  ReLU with a post-activation
 * residual is refused.
 * Replaces: nn.BatchNorm2d + nn.ReLU in training mode (resnet.py:121-143, deeplabv3plus.py:19-27,63-68) and their autograd. */
int cvx_bn_act_train_nhwc(const float* y_f32, int32_t batch, int32_t hw, int32_t c, const float* gamma, const float* beta, float eps,
                          float momentum, float* running_mean, float* running_var, const void* res_f16, int32_t act, int32_t res_pre,
                          void* out_f16, void* xhat_f16, float* mean, float* invstd, void* hip_stream);
int cvx_bn_act_bwd_nhwc(const void* xhat_f16, const void* gout_f16, const void* out_f16, int32_t batch, int32_t hw, int32_t c,
                        const float* gamma, const float* beta, const float* invstd, int32_t act, int32_t res_pre, float inv_scale,
                        float* dgamma, float* dbeta, void* dy_f16, void* gres_f16, int32_t res_accumulate, void* hip_stream);
/* 5x5 / stride 1 / pad 2 max pool (SPPF, core/models/yolov8/modules.py:312-318) and its backward; argmax (optional in
 * the forward): one byte per element, the window tap 0..24 of the first maximum in row-major scan order. */
int cvx_maxpool5_nhwc(const void* x_f16, int32_t batch, int32_t h, int32_t w, int32_t c, void* out_f16, uint8_t* argmax, void* hip_stream);
int cvx_maxpool5_bwd_nhwc(const void* gout_f16, const uint8_t* argmax, int32_t batch, int32_t h, int32_t w, int32_t c, void* gin_f16,
                          int32_t accumulate, void* hip_stream);
/* The inference-only pooling / resampling / normalisation kernels of the DLA, ResNet + DeepLab and VGG + SSD graphs on dense NHWC fp16
 * tensors (unit parity tests, host code that wants one op); they synchronise before returning.
 *   cvx_maxpool_nhwc: kernel 2 / stride 2 (ceil_mode 0 | 1: centernet_model.py:128, ssd_model.py:16-18), kernel 3 / pad 1 / stride 1 | 2
 *     (ssd_model.py:30, resnet.py:163); output (batch, oh, ow, c) with torch's output-size rule.
 *   cvx_avgpool_global_nhwc: (batch, hw, c) -> (batch, 1, c) mean in fp32 (deeplabv3plus.py:30).
 *   cvx_resize_bilinear_nhwc: align_corners = False (deeplabv3plus.py:38,117-122).
 *   cvx_l2norm_nhwc: x / (sqrt(sum_c x^2) + 1e-10) * weight[c] (ssd_model.py:113-128). */
int cvx_maxpool_nhwc(const void* x_f16, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t kernel, int32_t stride, int32_t ceil_mode,
                     void* out_f16, void* hip_stream);
int cvx_avgpool_global_nhwc(const void* x_f16, int32_t batch, int32_t hw, int32_t c, void* out_f16, void* hip_stream);
int cvx_resize_bilinear_nhwc(const void* x_f16, int32_t batch, int32_t ih, int32_t iw, int32_t c, int32_t oh, int32_t ow, void* out_f16,
                             void* hip_stream);
int cvx_l2norm_nhwc(const void* x_f16, const float* weight, int32_t batch, int32_t hw, int32_t c, void* out_f16, void* hip_stream);
/* Bilinear resize (align_corners = False) of fp32 rows (batch, ih*iw, ld) -- e.g. a PRED buffer holding segmentation logits --
 * into an NCHW fp32 tensor (batch, c, oh, ow).  Asynchronous on hip_stream.
 * Replaces: F.interpolate(x, size=input_shape, mode="bilinear", align_corners=False), core/models/deeplabv3plus.py:147. */
int cvx_resize_bilinear_rows_to_nchw(const float* rows_f32, int32_t ld, int32_t batch, int32_t c, int32_t ih, int32_t iw, int32_t oh, int32_t ow,
                                     float* out_nchw, void* hip_stream);
/* Segmentation loss with its gradient, for the DeepLabv3+ step.  rows: the engine's fp32 logits rows (batch, ih*iw, ld) at decoder
 * resolution; target: (batch, oh, ow) int64 class indices at label resolution (= the network input size).  Per label pixel the nc
 * logits are interpolated bilinearly (align_corners = False, deeplabv3plus.py:147), then
 *   mode 0: focal loss  alpha * (1 - pt)^gamma * ce, pt = exp(-ce), mean over ALL batch*oh*ow pixels (focal_loss.py:14-22;
 *           ignored pixels contribute 0 to the sum and count in the mean, like the reference's .mean());
 *   mode 1: cross-entropy, mean over the non-ignored pixels (nn.CrossEntropyLoss(reduction="mean"), segmentation_2d.py:61).
 * loss_out: 1 float (device).  dpred: (batch, ih*iw, ld) fp16 = loss_scale * dLoss/drows (columns nc..ld-1 zero) -- the operand of
 * cvx_engine_backward.  bad_target: 1 int32 (device), set non-zero when a label is neither ignore_index nor in [0, nc) (torch
 * raises a device assert there; the pixel is skipped).  workspace: cvx_seg_loss_workspace_bytes() bytes.  Asynchronous on hip_stream.
 * Replaces: F.interpolate + FocalLoss.forward / nn.CrossEntropyLoss + loss.backward() down to the classifier's output,
 * core/trainer/segmentation_trainer.py:121-130. */
int64_t cvx_seg_loss_workspace_bytes(int32_t batch, int32_t nc, int32_t oh, int32_t ow);
int cvx_seg_loss(const float* rows_f32, int32_t ld, int32_t batch, int32_t nc, int32_t ih, int32_t iw, int32_t oh, int32_t ow,
                 const int64_t* target, int32_t mode, float alpha, float gamma, int64_t ignore_index, float loss_scale, float* loss_out,
                 void* dpred_f16, int32_t* bad_target, void* workspace, void* hip_stream);
/* CenterNet's CombinedLoss with its gradient.  rows: the engine's fp32 head rows (batch, anchors = h*w, ld): heat-map logits in columns
 * [0, nc), the loss's "reg" pair at columns col_a, col_a+1 (= the model output's columns nc, nc+1) and its "wh" pair at col_b, col_b+1 (= the
 * output's last two) -- the reference's loss names are swapped against the heads that produce them, reproduced as is.  Targets as
 * centernet_collate builds them: heat_true (batch, h, w, nc) fp32, true_a / true_b (batch, K, 2), mask (batch, K) fp32, indices (batch, K)
 * int64 = y*w + x.  loss_items: 4 floats (device): total, heat-map focal, L1(a), L1(b) (unweighted).  dpred: (batch, anchors, ld) fp16 =
 * loss_scale * dLoss/drows.  bad_index: set non-zero when a masked object's index lies outside [0, anchors).  Asynchronous on hip_stream.
 * Replaces: CombinedLoss.__call__ + loss.backward() down to the head outputs, core/loss/centernet_loss.py:5-67,
 * core/trainer/centernet_train.py:104-118. */
int64_t cvx_centernet_loss_workspace_bytes(int32_t batch, int32_t anchors);
int cvx_centernet_loss(const float* rows_f32, int32_t ld, int32_t batch, int32_t anchors, int32_t nc, int32_t col_a, int32_t col_b,
                       const float* heat_true, const float* true_a, const float* true_b, const float* mask, const int64_t* indices,
                       int32_t max_objects, float hm_weight, float a_weight, float b_weight, float loss_scale, float* loss_items, void* dpred_f16,
                       int32_t* bad_index, void* workspace, void* hip_stream);
/* SSD's MultiBoxLossV2 with its gradient.  loc (batch, anchors, 4) and conf (batch, anchors, nc1 = num_classes + 1) fp32 as the model
 * returns them; y_true (batch, anchors, 4 + nc1 + 1) fp32 as ssd_collate encodes it: box targets, one-hot class incl. background, positive
 * flag.  Softmax cross-entropy (probabilities clamped at 1e-7) on the positives and on the k hardest negatives of the WHOLE batch
 * (k = sum over images of min(ratio * num_pos, anchors - num_pos), 100 when no image has a positive; hardness = sum of the non-background
 * probabilities), smooth-L1 on the positives, total = (1 - alpha) * conf + alpha * loc with both sums divided by sum(num_pos or 1).  The
 * top-k is a radix SELECT (no sort); keys equal to the k-th are taken in flat-index order (torch.topk leaves that choice open).
 * loss_items: 3 floats (device): total, loc, conf.  dloc / dconf: grad_scale * dLoss/d(loc, conf), fp32, same shapes.  Asynchronous.
 * Replaces: MultiBoxLossV2.__call__ + loss.backward() down to the model outputs, core/loss/multi_box_loss.py:77-192. */
int64_t cvx_multibox_loss_workspace_bytes(int32_t batch, int32_t anchors);
int cvx_multibox_loss(const float* loc, const float* conf, const float* y_true, int32_t batch, int32_t anchors, int32_t nc1, float neg_pos_ratio,
                      float alpha, float grad_scale, float* loss_items, float* dloc, float* dconf, void* workspace, void* hip_stream);
/* YOLOv7's loss (Yolo7Loss) with its gradient.  rows: the engine's fp32 head rows (batch, anchors, ld): level l (coarsest first, level_hw =
 * {h0, w0, h1, w1, h2, w2}) occupies rows a_off_l .. + h_l*w_l in (gj, gi) order, anchor a of a cell in columns a*(5+nc) .. : x, y, w, h,
 * objectness, classes.  anchors_px: 9 (w, h) pairs in pixels, level-major (the reference's anchors[anchors_mask[l]]); strides {32, 16, 8};
 * targets: (n_targets, 6) fp32 [image, class, cx, cy, w, h] normalised; img_size = imgs[b].shape[1] (the reference scales all four
 * coordinates by it).  Candidate generation (find_3_positive), SimOTA assignment per image (build_targets: dynamic k = int(sum of the
 * top-20 IoUs) >= 1, the k cheapest candidates, conflicts to the cheapest ground truth) and the CIoU / objectness / class terms all run on
 * the device.  loss_items: 4 floats: total, box * box_ratio, obj * obj_ratio, cls * cls_ratio.  dpred: (batch, anchors, ld) fp16 =
 * loss_scale * dLoss/drows.  bad: bit 0 = more than 64 ground truths in an image, bit 1 = more than 2880 candidates (both truncated).
 * Replaces: Yolo7Loss.__call__ + loss.backward() down to the head outputs, core/loss/yolo7_loss.py:14-444. */
int64_t cvx_yolo7_loss_workspace_bytes(int32_t batch, int32_t anchors, int32_t ld, int32_t n_targets);
int cvx_yolo7_loss(const float* rows_f32, int32_t ld, int32_t batch, int32_t nc, const int32_t* level_hw, const float* anchors_px, const float* strides,
                   const float* targets, int32_t n_targets, float img_size, float box_ratio, float obj_ratio, float cls_ratio, float label_smoothing,
                   float loss_scale, float* loss_items, void* dpred_f16, int32_t* bad, void* workspace, void* hip_stream);
/* CenterNet target drawing on the device: what centernet_collate does per image on the CPU.  labels: (batch, max_boxes, 5) fp32 rows
 * [class id, cx, cy, w, h] (normalised), counts (batch) valid rows (<= max_boxes = cfg.train.max_num_boxes).  Outputs in the reference's
 * formats: heatmap (batch, fh, fw, nc) -- per object a (2r+1)^2 Gaussian, r the CornerNet radius of its integer size, merged by maximum;
 * reg (batch, max_boxes, 2) = fractional part of the centre; wh = integer (w, h); reg_mask; indices = y*fw + x as float32.  Asynchronous.
 * Replaces: CenterNet.generate_targets, core/algorithms/centernet.py:66-112 + core/utils/gaussian.py:5-57 (called from collate.py:52-68). */
int cvx_centernet_draw_targets(const float* labels, const int32_t* counts, int32_t batch, int32_t max_boxes, int32_t fh, int32_t fw, int32_t nc,
                               float* heatmap, float* reg, float* wh, float* reg_mask, float* indices, void* hip_stream);
/* SSD target encoding on the device: what ssd_collate does per image on the CPU.  labels: (batch, max_boxes, 5) fp32 rows
 * [class id (0-based), cx, cy, w, h] (normalised), counts (batch) valid rows per image; priors (anchors, 4) fp32 corner boxes
 * (Ssd._get_ssd_anchors).  y_true: (batch, anchors, 4 + nc1 + 1) fp32: encoded box | one-hot class incl. the background column 0 | positive
 * flag -- the input of cvx_multibox_loss.  workspace: batch * max_boxes int32.  Asynchronous on hip_stream.
 * Replaces: Ssd.generate_targets + _encode_box, core/algorithms/ssd.py:327-480 (called from core/data/collate.py:32-49). */
int cvx_ssd_encode_targets(const float* labels, const int32_t* counts, int32_t batch, int32_t max_boxes, const float* priors, int32_t anchors, int32_t nc1,
                           float overlap_threshold, float variance_xy, float variance_wh, float* y_true, int32_t* workspace, void* hip_stream);
/* Adjoint of cvx_resize_bilinear_rows_to_nchw: a gradient w.r.t. the full-resolution logits (batch, nc, oh, ow) fp32 -> scale * the
 * gradient w.r.t. the rows (batch, ih*iw, ld) fp16 (a deterministic gather).  For callers that compute their own loss on the
 * model's NCHW output.  Asynchronous on hip_stream. */
int cvx_resize_bilinear_nchw_grad_to_rows(const float* grad_nchw, int32_t batch, int32_t nc, int32_t ih, int32_t iw, int32_t oh, int32_t ow,
                                          float scale, void* dpred_f16, int32_t ld, void* hip_stream);
/* Training forms of the ResNet / DeepLab pooling and resampling ops (dense NHWC fp16; they synchronise before returning):
 *   cvx_maxpool3_train_nhwc: the 3x3 / pad 1 / stride 1 | 2 max pool that also stores `argmax` (one byte per OUTPUT element, the
 *     window tap dy*3+dx of the first maximum in row-major scan order = torch's choice); cvx_maxpool3_bwd_nhwc routes gout through it
 *     (resnet.py:163 + autograd);
 *   cvx_avgpool_global_bwd_nhwc: gin (+)= gout / hw on every pixel (deeplabv3plus.py:30);
 *   cvx_resize_bilinear_bwd_nhwc: the adjoint of cvx_resize_bilinear_nhwc (align_corners = False) as a deterministic gather
 *     (deeplabv3plus.py:38,117-122);
 *   cvx_dropout_nhwc: inverted dropout, out (+)= keep ? x / (1 - p) : 0, the mask a counter-based hash of (seed, element index) --
 *     the same call with the same seed on the output gradient is its backward (deeplabv3plus.py:67).  The mask is NOT torch's
 *     Philox sample: same distribution, different draw.
 * accumulate != 0 adds onto the existing contents of the gradient tensor. */
int cvx_maxpool3_train_nhwc(const void* x_f16, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t stride, void* out_f16, uint8_t* argmax,
                            void* hip_stream);
int cvx_maxpool3_bwd_nhwc(const void* gout_f16, const uint8_t* argmax, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t stride,
                          void* gin_f16, int32_t accumulate, void* hip_stream);
int cvx_avgpool_global_bwd_nhwc(const void* gout_f16, int32_t batch, int32_t hw, int32_t c, void* gin_f16, int32_t accumulate, void* hip_stream);
/* Gradient of cvx_l2norm_nhwc (L2Normalize, ssd_model.py:113-128): gin (+)= w*g/(n+eps) - x*(sum_c w g x)/(n (n+eps)^2), and
 * dweight[c] += inv_scale * sum over pixels of g*x/(n+eps) (fixed summation order).  c <= 512. */
int cvx_l2norm_bwd_nhwc(const void* x_f16, const void* gout_f16, const float* weight, int32_t batch, int32_t hw, int32_t c, float inv_scale,
                        void* gin_f16, float* dweight, int32_t accumulate, void* hip_stream);
/* Adjoint of cvx_pred_cols_to_nchw: a gradient laid out like its output (element (b, ch, pix) at grad[b*grad_bstride + grad_off + ch*hw + pix]:
 * SSD's NCHW-order flattening, ssd_model.py:177-183) -> scale * gradient in fp16 on columns [col0, col0 + c) of rows a_off .. a_off + hw. */
int cvx_nchw_cols_grad_to_pred(const float* grad, int64_t grad_bstride, int64_t grad_off, int32_t c, int32_t batch, int32_t anchors, int32_t a_off,
                               int32_t hw, float scale, void* dpred_f16, int32_t ld, int32_t col0, void* hip_stream);
/* Gradient of the 2x2 stride-2 max pool of cvx_maxpool_nhwc (floor or ceil mode): x is the forward input, the first maximum of each
 * window in row-major scan order takes the gradient (torch's rule).  Replaces: nn.MaxPool2d(2, 2) autograd (yolov7_model.py:74-86). */
int cvx_maxpool2_bwd_nhwc(const void* x_f16, const void* gout_f16, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t ceil_mode,
                          void* gin_f16, int32_t accumulate, void* hip_stream);
/* Depthwise ConvTranspose2d of CenterNet's IDAUp.up_i (kernel 2f, stride f, padding f/2, groups = c, no bias; centernet_model.py:256) on a dense
 * NHWC fp16 tensor, and its gradients: gin (+)= data gradient, dweight[c][2f][2f] += inv_scale * weight gradient (fp32, deterministic).
 * weight: fp32 [c][2f][2f] (the master tensor).  Replaces: nn.ConvTranspose2d forward + autograd. */
int cvx_dwconvt_nhwc(const void* x_f16, int32_t batch, int32_t ih, int32_t iw, int32_t c, int32_t f, const float* weight, void* out_f16,
                     void* hip_stream);
int cvx_dwconvt_bwd_nhwc(const void* x_f16, const void* gout_f16, int32_t batch, int32_t ih, int32_t iw, int32_t c, int32_t f, const float* weight,
                         void* gin_f16, int32_t accumulate, float* dweight, float inv_scale, void* hip_stream);
int cvx_resize_bilinear_bwd_nhwc(const void* gout_f16, int32_t batch, int32_t ih, int32_t iw, int32_t c, int32_t oh, int32_t ow, void* gin_f16,
                                 int32_t accumulate, void* hip_stream);
int cvx_dropout_nhwc(const void* x_f16, int32_t batch, int32_t hw, int32_t c, float p, uint64_t seed, void* out_f16, int32_t accumulate,
                     void* hip_stream);

/* nearest-neighbour x2 upsample (nn.Upsample(scale_factor=2), core/models/yolov8/yolo_v8.py:39,41) and its backward */
int cvx_upsample2_nhwc(const void* x_f16, int32_t batch, int32_t h, int32_t w, int32_t c, void* out_f16, void* hip_stream);
int cvx_upsample2_bwd_nhwc(const void* gout_f16, int32_t batch, int32_t h, int32_t w, int32_t c, void* gin_f16, int32_t accumulate,
                           void* hip_stream);
/* The stem (model.0: Conv(3, c, 3, 2), core/models/yolov8/yolo_v8.py:28) in fp32 from NCHW fp32 images: weight fp32
 * [cout][kh][kw][ci]; train = batch statistics + running update, eval = folded scale / shift; wgrad: dw [cout][3][3][3]
 * (overwritten) from dy fp16 (B, h/2, w/2, cout). */
int cvx_stem_train_nchw(const float* images, int32_t batch, int32_t h, int32_t w, const float* weight, int32_t cout, const float* gamma,
                        const float* beta, float eps, float momentum, float* running_mean, float* running_var, void* out_f16,
                        void* xhat_f16, float* mean, float* invstd, void* hip_stream);
int cvx_stem_eval_nchw(const float* images, int32_t batch, int32_t h, int32_t w, const float* weight, int32_t cout, const float* scale,
                       const float* shift, void* out_f16, void* hip_stream);
int cvx_stem_wgrad_nchw(const float* images, int32_t batch, int32_t h, int32_t w, const void* dy_f16, int32_t cout, float* dw,
                        void* hip_stream);
/* The stem's backward pass as the engine runs it: BatchNorm + SiLU backward fused with the weight gradient (the gradient of
 * the raw conv output is formed in registers and never stored).  gout: gradient w.r.t. the stem's activation, fp16
 * (B, h/2, w/2, cout); xhat / invstd from cvx_stem_train_nchw; dgamma / dbeta accumulated, dw [cout][3][3][3] overwritten,
 * all scaled by inv_scale. */
int cvx_stem_backward_nchw(const float* images, int32_t batch, int32_t h, int32_t w, const void* xhat_f16, const void* gout_f16, int32_t cout,
                           const float* gamma, const float* beta, const float* invstd, float inv_scale, float* dgamma, float* dbeta, float* dw,
                           void* hip_stream);
/* The same pass where the engine keeps no xhat (round 5; images with 16-byte granular rows, >= 2 pixel splits: every YOLOv8 training shape): the
 * kernel recomputes the conv output from the images and `weight` ([cout][3][3][3] fp32, as cvx_stem_train_nchw takes it) and normalises it with the
 * forward's batch mean / invstd (cvx_stem_train_nchw's outputs),
 * and the BatchNorm-backward sums come out of the same pass (no reduction over gout beforehand).  Fails with an error on shapes the one-pass
 * kernel does not take (the engine falls back to cvx_stem_backward_nchw's route there). */
int cvx_stem_backward_recompute_nchw(const float* images, int32_t batch, int32_t h, int32_t w, const float* weight, const void* gout_f16, int32_t cout,
                                     const float* gamma, const float* beta, const float* mean, const float* invstd, float inv_scale,
                                     float* dgamma, float* dbeta, float* dw, void* hip_stream);

/* ---- data-parallel gradient exchange over RCCL (csrc/comm.hip), SURVEY.md section 8(b) / 8(e) --------------------------------------
 * One process per GPU.  Rank 0 calls cvx_comm_unique_id (128 bytes, ncclGetUniqueId), ships them to every rank by any means, all ranks
 * call cvx_comm_create (ncclCommInitRank; RCCL is dlopen'ed on first use).  `comm` below is the ncclComm_t.
 * cvx_allreduce_grads: SUM all-reduce of the engine's whole flat gradient arena on `hip_stream` (after cvx_engine_backward).
 * cvx_engine_backward_exchange: the overlapped form, the whole data-parallel backward pass in one call -- `buckets` holds n_buckets rows
 *   (op_hi, op_lo, p_start, p_end), op ranges in backward order with their slice of the gradient arena (graph.grad_buckets); each range
 *   is queued on the engine's stream, `comm_stream` waits for it, folds its weight-gradient slabs and all-reduces its slice while the next
 *   range runs; on return the engine's stream waits for the last exchange.  The mean's 1/world belongs to the optimiser step.
 * The reference has no distributed path; north_star: RCCL all-reduce of gradients overlapped with the backward pass. */
int cvx_comm_unique_id(void* out128);
int cvx_comm_create(void** comm, const void* unique_id128, int32_t rank, int32_t world, int32_t device);
int cvx_comm_destroy(void* comm);
int cvx_allreduce_f32(float* data, int64_t count, void* comm, void* hip_stream);
int cvx_allreduce_grads(cvx_engine* e, void* comm, void* hip_stream);
int cvx_engine_backward_exchange(cvx_engine* e, const void* dpred_f16, float loss_scale, void* comm, const int64_t* buckets, int32_t n_buckets,
                                 void* comm_stream);

/* (The tile-resident chain kernel's unit entry points -- cvx_chain_pair_unit / _conv_unit / _detect_unit, csrc/conv_chain.hip -- live in
 * include/cvx_engine_experimental.h: the kernel measured slower than the per-layer launches and is built into the tuning library only.) */


#ifdef __cplusplus
}
#endif
#endif /* CVX_ENGINE_H */
