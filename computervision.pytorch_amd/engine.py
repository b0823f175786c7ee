"""Python handle on a ``cvx_engine`` (one per device and input size) plus the standalone C-ABI ops.

PyTorch appears here only as the owner of device memory and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import warnings

import torch

from . import _lib as L
from .graph import Graph, ParamLayout, build_yolov8_graph, grad_buckets


def _need_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise L.CvxError(f"{what} must live on an MI355X device (got {t.device}); the engine has no CPU path")


class Engine:
    _warned_stream = False
    # False: every forward converts the weights again (for callers that write parameters in ways neither torch's version counter nor this
    # package can see: `p.data.mul_()` -- `.data` has a version counter of its own --, raw kernels of their own, another process)
    keep_weight_images = True

    def __init__(self, graph: Graph, device: torch.device):
        self.lib = L.load()
        self.graph = graph
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.CvxError("the engine needs a HIP device (torch device type 'cuda'); there is no CPU path")
        bufs, ops = graph.c_arrays()
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self.lib.cvx_engine_create(C.byref(h), bufs, len(graph.bufs), ops, len(graph.ops), graph.image_buf, graph.pred_buf,
                                               self.device.index or 0, L.stream_ptr(self.device)), "cvx_engine_create")
        self.handle = h
        self._bound = None

    def bind(self, params: torch.Tensor, grads: Optional[torch.Tensor], stats: torch.Tensor):
        for t, n in ((params, "params"), (stats, "stats")):
            _need_gpu(t, n)
            assert t.dtype == torch.float32 and t.is_contiguous()
        key = (params.data_ptr(), 0 if grads is None else grads.data_ptr(), stats.data_ptr())
        if key == self._bound:
            return
        L.check(self.lib.cvx_engine_bind(self.handle, L.ptr(params), L.ptr(grads), params.numel(), L.ptr(stats), stats.numel()),
                "cvx_engine_bind")
        self._bound = key
        self._keep = (params, grads, stats)

    def set_bn(self, eps: float, momentum: float):
        L.check(self.lib.cvx_engine_set_bn(self.handle, eps, momentum), "cvx_engine_set_bn")

    def set_fusion(self, enable: bool):
        """eval-mode cross-layer fusion (Bottleneck pairs, Detect levels as one launch each): tuning build only -- the release library raises
        CvxError for enable=True (the chain kernel measured slower than the per-layer launches and left it in round 4)"""
        L.check(self.lib.cvx_engine_set_fusion(self.handle, int(bool(enable))), "cvx_engine_set_fusion")

    def fused_groups(self) -> int:
        return int(self.lib.cvx_engine_fused_groups(self.handle))

    def set_seed(self, seed: int):
        """Seed of the graph's dropout masks (nn.Dropout layers of the reference model; torch.manual_seed there)."""
        if seed != getattr(self, "_seed", None):
            L.check(self.lib.cvx_engine_set_seed(self.handle, int(seed) & (2 ** 64 - 1)), "cvx_engine_set_seed")
            self._seed = seed

    def forward(self, images: torch.Tensor, training: bool, pred: Optional[torch.Tensor] = None) -> torch.Tensor:
        _need_gpu(images, "images")
        images = images.contiguous().float()
        b = images.shape[0]
        no = self.graph.bufs[self.graph.pred_buf][2]
        if pred is None:
            pred = torch.empty(b, self.graph.anchors, no, device=images.device, dtype=torch.float32)
        self._use_current_stream()
        # constant weights between two forwards of this engine (inference loops): skip the fp32 -> fp16 weight conversion.  "Constant" =
        # torch's in-place version counter of the bound parameter arena (optimisers on views, load_state_dict, broadcasts all bump it) AND
        # this package's own counter of raw-pointer writers (the Adam kernels) are where the previous forward left them.
        params = self._keep[0] if getattr(self, "_keep", None) else None
        wkey = None if params is None else (params.data_ptr(), params._version, PARAM_WRITES.get(params.data_ptr(), 0))
        if Engine.keep_weight_images and wkey is not None and wkey == getattr(self, "_weights_key", None) and not torch.cuda.is_current_stream_capturing():
            L.check(self.lib.cvx_engine_keep_shadows(self.handle), "cvx_engine_keep_shadows")
        self._weights_key = wkey
        L.check(self.lib.cvx_engine_forward(self.handle, L.ptr(images), b, 1 if training else 0, L.ptr(pred)), "cvx_engine_forward")
        # the fp32 stem reads the caller's tensor directly, and its weight gradient reads it again in the backward pass
        self._images = images if training else None
        return pred

    def exchange_stream(self) -> torch.cuda.Stream:
        """The stream the data-parallel gradient exchange is queued on, as a torch stream: the engine's own weight-gradient stream
        (include/cvx_engine.h: cvx_engine_exchange_stream) -- a stream of its own for the exchange would be one hardware queue too many
        once RCCL's internal stream joins in."""
        if getattr(self, "_xstream", None) is None:
            ptr = self.lib.cvx_engine_exchange_stream(self.handle)
            if not ptr:
                raise L.CvxError("cvx_engine_exchange_stream failed")
            self._xstream = torch.cuda.ExternalStream(int(ptr), device=self.device)
        return self._xstream

    def plan_generation(self) -> int:
        """Counts re-allocations of the per-batch buffers (a captured hipGraph is stale once this changes)."""
        return int(self.lib.cvx_engine_plan_generation(self.handle))

    def _use_current_stream(self):
        """Enqueue on torch's current stream of this device (it may be a stream under hipGraph capture)."""
        s = torch.cuda.current_stream(self.device).cuda_stream
        if s != 0 and not Engine._warned_stream and not torch.cuda.is_current_stream_capturing():
            Engine._warned_stream = True
            warnings.warn("the engine is being launched on a non-default HIP stream: fine by itself (the engine leaves one hardware queue "
                          "to the caller), but a process that works more than four queues -- this stream, the default stream, the engine's two, "
                          "a data-parallel exchange stream, further torch streams -- pays 2.2-2.5x per train step on the ROCm 7 runtime "
                          "(tools/stream_probe.py, DESIGN.md section 6)", RuntimeWarning, stacklevel=3)
        if s != getattr(self, "_stream", None):
            L.check(self.lib.cvx_engine_set_stream(self.handle, C.c_void_p(s)), "cvx_engine_set_stream")
            self._stream = s

    def backward(self, dpred_f16: torch.Tensor, loss_scale: float):
        assert dpred_f16.dtype == torch.float16 and dpred_f16.is_contiguous()
        self._use_current_stream()
        L.check(self.lib.cvx_engine_backward(self.handle, L.ptr(dpred_f16), float(loss_scale)), "cvx_engine_backward")

    # ---- segmented backward (data-parallel overlap, include/cvx_engine.h) ----
    def backward_begin(self, dpred_f16: torch.Tensor, loss_scale: float):
        assert dpred_f16.dtype == torch.float16 and dpred_f16.is_contiguous()
        self._use_current_stream()
        L.check(self.lib.cvx_engine_backward_begin(self.handle, L.ptr(dpred_f16), float(loss_scale)), "cvx_engine_backward_begin")

    def backward_range(self, op_hi: int, op_lo: int):
        L.check(self.lib.cvx_engine_backward_range(self.handle, op_hi, op_lo), "cvx_engine_backward_range")

    def grads_ready(self, op_hi: int, op_lo: int, stream: torch.cuda.Stream):
        """Makes `stream` wait for the range's gradients and fold its weight-gradient slabs into the arena there."""
        if stream is None:                                           # inline exchange: fold on the launch stream itself
            stream = torch.cuda.current_stream(self.device)
        L.check(self.lib.cvx_engine_grads_ready(self.handle, op_hi, op_lo, C.c_void_p(stream.cuda_stream)), "cvx_engine_grads_ready")

    def backward_end(self):
        L.check(self.lib.cvx_engine_backward_end(self.handle), "cvx_engine_backward_end")

    def grad_buckets(self, layout: ParamLayout, n_buckets: int):
        return grad_buckets(self.graph, layout, n_buckets)

    def read_buffer(self, buf: int, batch: int, grad: bool = False) -> torch.Tensor:
        """Debug: NHWC fp16 copy of an engine buffer (activation or gradient)."""
        h, w, c, _ = self.graph.bufs[buf]
        out = torch.empty(batch, h, w, c, dtype=torch.float16, device=self.device)
        L.check(self.lib.cvx_engine_debug_copy(self.handle, buf, 1 if grad else 0, L.ptr(out), out.numel() * 2), "cvx_engine_debug_copy")
        return out

    def read_layer(self, op: int, batch: int, what: str = "xhat") -> torch.Tensor:
        """Debug: a conv op's normalised output (``xhat``, fp16 as kept -- or rounded from the kept raw output of a CVX_OPF_RAW_F16 layer;
        ``xhat32``: in fp32, exactly as the backward passes use it) or the gradient w.r.t. its raw output (``dy``), NHWC."""
        o = self.graph.ops[op]
        wide = what == "xhat32"
        out = torch.empty(batch, o["oh"], o["ow"], o["out"][2], dtype=torch.float32 if wide else torch.float16, device=self.device)
        which = {"xhat": 2, "dy": 3, "xhat32": 4}[what]
        L.check(self.lib.cvx_engine_debug_copy(self.handle, op, which, L.ptr(out), out.numel() * (4 if wide else 2)), "cvx_engine_debug_copy")
        return out

    PROFILE_CLASSES = ("conv_fwd", "conv_dgrad", "conv_wgrad", "bn_silu_fwd", "bn_silu_bwd", "misc", "slab_reduce")

    def profile(self, enable: bool):
        L.check(self.lib.cvx_engine_profile(self.handle, 1 if enable else 0), "cvx_engine_profile")

    def profile_read(self):
        """{class: dict(ms, flops, bytes, launches)} accumulated since the profiling window started."""
        n = len(self.PROFILE_CLASSES)
        ms, fl, by, la = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)(), (C.c_int64 * n)()
        L.check(self.lib.cvx_engine_profile_read(self.handle, n, ms, fl, by, la), "cvx_engine_profile_read")
        return {name: dict(ms=ms[i], flops=fl[i], bytes=by[i], launches=int(la[i])) for i, name in enumerate(self.PROFILE_CLASSES)}

    def profile_dump(self, path: str):
        """Per-scope CSV (class, op, flops, bytes, ms) of the current profiling window."""
        L.check(self.lib.cvx_engine_profile_dump(self.handle, str(path).encode()), "cvx_engine_profile_dump")

    def workspace_bytes(self) -> int:
        return int(self.lib.cvx_engine_workspace_bytes(self.handle))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.cvx_engine_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


# ---- standalone ops ---------------------------------------------------------------------------------
def _levels(level_hw: Sequence[Sequence[int]], strides: Sequence[float]):
    flat = [int(v) for hw in level_hw for v in hw]
    return (C.c_int32 * len(flat))(*flat), (C.c_float * len(strides))(*[float(s) for s in strides]), len(strides)


class V8LossOp:
    """cvx_loss_v8: loss items + fp16 gradient w.r.t. pred.  Returns ``(items, dpred)`` with dpred of shape
    (B, A, ld), ld = row pitch of pred (>= no; columns beyond no are zero) -- the layout cvx_engine_backward reads."""

    def __init__(self, nc: int, gains=(7.5, 0.5, 1.5)):
        self.lib = L.load()
        self.nc = nc
        self.gains = tuple(float(g) for g in gains)
        self._ws = None

    def __call__(self, pred: torch.Tensor, targets: torch.Tensor, level_hw, strides, loss_scale: float,
                 dpred: Optional[torch.Tensor] = None):
        _need_gpu(pred, "pred")
        # pred may be the [..., :64+nc] view of a row-padded buffer (class counts that are not multiples of 8)
        B, A, no = pred.shape
        assert pred.dtype == torch.float32 and no == self.nc + 64
        if not (pred.stride(2) == 1 and pred.stride(0) == A * pred.stride(1) and pred.stride(1) % 4 == 0):
            padded = torch.zeros(B, A, (no + 3) & ~3, dtype=torch.float32, device=pred.device)   # 16-byte row pitch
            padded[..., :no] = pred
            pred = padded[..., :no]
        ld = pred.stride(1)
        n = int(targets.shape[0])
        if n:
            _need_gpu(targets, "targets")
            targets = targets.contiguous().float()
        cap = max(n, 1)
        need = int(self.lib.cvx_loss_v8_workspace_bytes(B, A, self.nc, cap))
        if self._ws is None or self._ws.numel() < need or self._ws.device != pred.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=pred.device)
        if dpred is None:
            dpred = torch.zeros(B, A, ld, dtype=torch.float16, device=pred.device)   # padding columns stay zero
        assert dpred.is_contiguous() and dpred.shape[2] == ld
        items = torch.empty(3, dtype=torch.float32, device=pred.device)
        lv, st, nl = _levels(level_hw, strides)
        L.check(self.lib.cvx_loss_v8_strided(L.ptr(pred), ld, B, A, self.nc, L.ptr(targets) if n else C.c_void_p(0), n, cap, lv, st, nl,
                                             self.gains[0], self.gains[1], self.gains[2], float(loss_scale), L.ptr(items), L.ptr(dpred),
                                             L.ptr(self._ws), self._ws.numel(), L.stream_ptr(pred.device)), "cvx_loss_v8")
        self._last_shape = (B, A, cap)
        return items, dpred

    def assignment(self, B: int, A: int, n_targets: int):
        """Per-anchor result of the task-aligned assigner of the LAST call: (target row index or -1, normalised target score)."""
        assert self._ws is not None and self._last_shape == (B, A, max(n_targets, 1)), "call the loss first (same shapes)"
        dev = self._ws.device
        idx = torch.empty(B, A, dtype=torch.int32, device=dev)
        norm = torch.empty(B, A, dtype=torch.float32, device=dev)
        L.check(self.lib.cvx_loss_v8_assignment(L.ptr(self._ws), B, A, max(n_targets, 1), L.ptr(idx), L.ptr(norm), L.stream_ptr(dev)),
                "cvx_loss_v8_assignment")
        return idx.long(), norm


# writes to a parameter arena that torch does not see (raw-pointer kernels): data_ptr -> count.  Engine.forward compares it, together with
# the tensor's own in-place version, against what it was at the engine's previous forward before it lets the engine keep its fp16 shadows.
PARAM_WRITES: dict = {}


def _note_param_write(params: torch.Tensor):
    PARAM_WRITES[params.data_ptr()] = PARAM_WRITES.get(params.data_ptr(), 0) + 1


def adam_step(params, grads, exp_avg, exp_avg_sq, lr, betas, eps, step, found_inf=None, zero_grad=True):
    lib = L.load()
    _need_gpu(params, "params")
    _note_param_write(params)
    L.check(lib.cvx_adam_step(L.ptr(params), L.ptr(grads), L.ptr(exp_avg), L.ptr(exp_avg_sq), params.numel(), lr, betas[0], betas[1], eps,
                              step, L.ptr(found_inf), 1 if zero_grad else 0, L.stream_ptr(params.device)), "cvx_adam_step")


def adam_step_dev(params, grads, exp_avg, exp_avg_sq, betas, eps, state, found_inf=None, zero_grad=True, grad_scale=1.0):
    """Adam with the step state on the device (state = [lr, step, lr/bc1, 1/sqrt(bc2)]): graph-replayable."""
    lib = L.load()
    _need_gpu(params, "params")
    _note_param_write(params)
    L.check(lib.cvx_adam_step_dev(L.ptr(params), L.ptr(grads), L.ptr(exp_avg), L.ptr(exp_avg_sq), params.numel(), betas[0], betas[1], eps,
                                  L.ptr(state), L.ptr(found_inf), 1 if zero_grad else 0, float(grad_scale), L.stream_ptr(params.device)),
            "cvx_adam_step_dev")


def check_finite(grads: torch.Tensor, found_inf: torch.Tensor):
    lib = L.load()
    L.check(lib.cvx_check_finite(L.ptr(grads), grads.numel(), L.ptr(found_inf), L.stream_ptr(grads.device)), "cvx_check_finite")


def decode(pred: torch.Tensor, nc: int, level_hw, strides) -> torch.Tensor:
    lib = L.load()
    _need_gpu(pred, "pred")
    B, A, no = pred.shape
    if not (pred.stride(2) == 1 and pred.stride(0) == A * pred.stride(1)):
        pred = pred.contiguous()
    y = torch.empty(B, 4 + nc, A, dtype=torch.float32, device=pred.device)
    lv, st, nl = _levels(level_hw, strides)
    L.check(lib.cvx_decode_strided(L.ptr(pred), pred.stride(1), B, A, nc, lv, st, nl, L.ptr(y), L.stream_ptr(pred.device)), "cvx_decode")
    return y


_nms_ws = {}


NMS_VARIANTS = {"tv0141_cuda": 0, "tv0141_cpu": 1, "offset": 2, "vanilla": 3}


def nms(y: torch.Tensor, conf_thres: float, iou_thres: float, max_det: int = 300, variant: str = "tv0141_cuda", boxes_xyxy: bool = False):
    """y (B, 4+nc, A) fp32 -> (rows (B,max_det,6), anchor_index (B,max_det) int32, counts (B,) int32), all on device.
    ``variant``: torchvision 0.14.1 ``batched_nms`` strategy (include/cvx_engine.h); the default is the library's own
    switch for CUDA tensors, i.e. what the reference's GPU predict path runs."""
    lib = L.load()
    _need_gpu(y, "y")
    if not (0 <= conf_thres <= 1 and 0 <= iou_thres <= 1):
        raise AssertionError("thresholds must lie in [0, 1]")       # ultralytics_ops.py:173-174
    y = y.contiguous().float()
    B, ch, A = y.shape
    nc = ch - 4
    need = int(lib.cvx_nms_workspace_bytes(B, A))
    ws = _nms_ws.get(y.device)
    if ws is None or ws.numel() < need:
        ws = _nms_ws[y.device] = torch.empty(need, dtype=torch.uint8, device=y.device)
    rows = torch.zeros(B, max_det, 6, dtype=torch.float32, device=y.device)
    index = torch.zeros(B, max_det, dtype=torch.int32, device=y.device)
    counts = torch.zeros(B, dtype=torch.int32, device=y.device)
    L.check(lib.cvx_nms_variant(L.ptr(y), B, A, nc, conf_thres, iou_thres, max_det, NMS_VARIANTS[variant] | (0x100 if boxes_xyxy else 0), L.ptr(rows), L.ptr(index),
                                L.ptr(counts), L.ptr(ws), ws.numel(), L.stream_ptr(y.device)), "cvx_nms")
    return rows, index, counts


def yolo7_decode(rows: torch.Tensor, nc: int, level_hw, anchors_wh, input_hw, want_nms_input: bool = True):
    """rows (B, sum h*w, ld) fp32 head rows of the YOLOv7 graph -> (dec (B, 3*sum, 5+nc), y (B, 4+nc, 3*sum) or None), on device
    (cvx_yolo7_decode, include/cvx_engine.h).  ``anchors_wh``: per level three (w, h) pairs in input pixels."""
    lib = L.load()
    _need_gpu(rows, "rows")
    rows = rows.contiguous().float()
    B, R, ld = rows.shape
    assert R == sum(h * w for h, w in level_hw)
    dev = rows.device
    dec = torch.empty(B, 3 * R, 5 + nc, dtype=torch.float32, device=dev)
    y = torch.empty(B, 4 + nc, 3 * R, dtype=torch.float32, device=dev) if want_nms_input else None
    lv = (C.c_int32 * (2 * len(level_hw)))(*[int(v) for hw in level_hw for v in hw])
    an = (C.c_float * (6 * len(level_hw)))(*[float(v) for lvl in anchors_wh for wh in lvl for v in wh])
    L.check(lib.cvx_yolo7_decode(L.ptr(rows), ld, B, nc, lv, an, len(level_hw), int(input_hw[0]), int(input_hw[1]), L.ptr(dec), L.ptr(y),
                                 L.stream_ptr(dev)), "cvx_yolo7_decode")
    return dec, y


def ssd_decode(loc: torch.Tensor, conf: torch.Tensor, priors: torch.Tensor, variances=(0.1, 0.2), with_class_max: bool = False):
    """loc (B, A, 4), conf (B, A, nc+1) fp32 (the SSD model's outputs), priors (A, 4) -> (boxes (B, A, 4) clipped corners,
    prob (B, A, nc+1) softmax), on device (cvx_ssd_decode, include/cvx_engine.h); with_class_max: also the largest probability per
    score column over the batch ((nc+1,) tensor, cvx_ssd_decode_max)."""
    lib = L.load()
    _need_gpu(loc, "loc")
    loc, conf, priors = loc.contiguous().float(), conf.contiguous().float(), priors.contiguous().float()
    B, A, _ = loc.shape
    boxes, prob = torch.empty_like(loc), torch.empty_like(conf)
    cmax = torch.empty(conf.shape[2], device=loc.device) if with_class_max else None
    L.check(lib.cvx_ssd_decode_max(L.ptr(loc), L.ptr(conf), L.ptr(priors), B, A, conf.shape[2], float(variances[0]), float(variances[1]), L.ptr(boxes),
                                   L.ptr(prob), L.ptr(cmax) if with_class_max else None, L.stream_ptr(loc.device)), "cvx_ssd_decode_max")
    return (boxes, prob, cmax) if with_class_max else (boxes, prob)


_cn_ws = {}


def centernet_decode(pred: torch.Tensor, h: int, w: int, nc: int, reg_col: int, wh_col: int, k: int = 100, conf: float = 0.1,
                     nms_thr: float = 0.5, use_nms: bool = True):
    """pred (B, h*w, ld) fp32 head tensor -> dict(boxes (B,k,4) xyxy in [0,1], scores, classes, topk_index, keep, counts), on device
    (cvx_centernet_decode, include/cvx_engine.h)."""
    lib = L.load()
    _need_gpu(pred, "pred")
    pred = pred.contiguous().float()
    B, hw, ld = pred.shape
    assert hw == h * w
    need = int(lib.cvx_centernet_decode_workspace_bytes(B, h, w, nc))
    ws = _cn_ws.get(pred.device)
    if ws is None or ws.numel() < need:
        ws = _cn_ws[pred.device] = torch.empty(need, dtype=torch.uint8, device=pred.device)
    dev = pred.device
    out = dict(boxes=torch.zeros(B, k, 4, device=dev), scores=torch.zeros(B, k, device=dev),
               classes=torch.zeros(B, k, dtype=torch.int32, device=dev), topk_index=torch.zeros(B, k, dtype=torch.int32, device=dev),
               keep=torch.zeros(B, k, dtype=torch.int32, device=dev), counts=torch.zeros(B, dtype=torch.int32, device=dev))
    L.check(lib.cvx_centernet_decode(L.ptr(pred), ld, B, h, w, nc, reg_col, wh_col, k, float(conf), float(nms_thr), 1 if use_nms else 0,
                                     L.ptr(out["boxes"]), L.ptr(out["scores"]), L.ptr(out["classes"]), L.ptr(out["topk_index"]),
                                     L.ptr(out["keep"]), L.ptr(out["counts"]), L.ptr(ws), ws.numel(), L.stream_ptr(dev)), "cvx_centernet_decode")
    return out


def letterbox_geometry(h: int, w: int, H: int, W: int):
    """(new_h, new_w, top, left, scale) of the reference's letter_box (core/utils/image_process.py:56-62); host arithmetic of the library."""
    import ctypes as C
    lib = L.load()
    nh, nw, top, left, sc = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_double()
    L.check(lib.cvx_letterbox_geometry(h, w, H, W, C.byref(nh), C.byref(nw), C.byref(top), C.byref(left), C.byref(sc)), "cvx_letterbox_geometry")
    return nh.value, nw.value, top.value, left.value, sc.value


def letterbox_u8(image_u8: torch.Tensor, out_chw: torch.Tensor, letterbox: bool = True, swap_rb: bool = False) -> torch.Tensor:
    """uint8 (h, w, 3) device image -> out_chw (3, H, W) fp32 (a slot of the batch tensor): letter_box + to_tensor in one launch."""
    _need_gpu(image_u8, "image")
    if image_u8.dtype != torch.uint8 or image_u8.dim() != 3 or image_u8.shape[2] != 3:
        raise ValueError("image must be a uint8 (h, w, 3) tensor")
    if out_chw.dtype != torch.float32 or out_chw.dim() != 3 or out_chw.shape[0] != 3 or not out_chw.is_contiguous():
        raise ValueError("out must be a contiguous float32 (3, H, W) tensor")
    image_u8 = image_u8.contiguous()
    lib = L.load()
    L.check(lib.cvx_letterbox_u8_to_nchw(L.ptr(image_u8), int(image_u8.shape[0]), int(image_u8.shape[1]), 1 if letterbox else 0, 1 if swap_rb else 0,
                                         L.ptr(out_chw), int(out_chw.shape[1]), int(out_chw.shape[2]), L.stream_ptr(image_u8.device)),
            "cvx_letterbox_u8_to_nchw")
    return out_chw


def ssd_encode_targets(labels: torch.Tensor, counts: torch.Tensor, priors: torch.Tensor, num_classes: int, overlap_threshold: float = 0.5,
                       variances=(0.1, 0.2)) -> torch.Tensor:
    """labels (B, Nmax, 5) fp32 [class id, cx, cy, w, h], counts (B) int32, priors (A, 4) fp32 -> y_true (B, A, 4 + (nc + 1) + 1) fp32: the
    reference's Ssd.generate_targets for a whole batch in two launches (core/algorithms/ssd.py:327-480)."""
    _need_gpu(labels, "labels")
    lib = L.load()
    B, nmax = int(labels.shape[0]), int(labels.shape[1])
    A, nc1 = int(priors.shape[0]), int(num_classes) + 1
    labels, priors = labels.contiguous().float(), priors.to(labels.device).contiguous().float()
    counts = counts.to(labels.device).to(torch.int32).contiguous()
    y = torch.empty(B, A, 4 + nc1 + 1, device=labels.device)
    if nmax == 0:
        y.zero_()
        y[:, :, 4] = 1.0
        return y
    ws = torch.empty(B * nmax, dtype=torch.int32, device=labels.device)
    L.check(lib.cvx_ssd_encode_targets(L.ptr(labels), L.ptr(counts), B, nmax, L.ptr(priors), A, nc1, float(overlap_threshold), float(variances[0]),
                                       float(variances[1]), L.ptr(y), L.ptr(ws), L.stream_ptr(labels.device)), "cvx_ssd_encode_targets")
    return y


def centernet_draw_targets(labels: torch.Tensor, counts: torch.Tensor, feature_hw, num_classes: int):
    """labels (B, K, 5) fp32 [class id, cx, cy, w, h], counts (B) -> [heatmap (B,h,w,nc), reg (B,K,2), wh (B,K,2), reg_mask (B,K),
    indices (B,K)]: the reference's CenterNet.generate_targets for a whole batch in one launch (core/algorithms/centernet.py:66-112)."""
    _need_gpu(labels, "labels")
    lib = L.load()
    B, K = int(labels.shape[0]), int(labels.shape[1])
    h, w = feature_hw
    dev = labels.device
    labels = labels.contiguous().float()
    counts = counts.to(dev).to(torch.int32).contiguous()
    heat = torch.empty(B, h, w, int(num_classes), device=dev)
    reg, wh = torch.empty(B, K, 2, device=dev), torch.empty(B, K, 2, device=dev)
    mask, ind = torch.empty(B, K, device=dev), torch.empty(B, K, device=dev)
    L.check(lib.cvx_centernet_draw_targets(L.ptr(labels), L.ptr(counts), B, K, h, w, int(num_classes), L.ptr(heat), L.ptr(reg), L.ptr(wh), L.ptr(mask),
                                           L.ptr(ind), L.stream_ptr(dev)), "cvx_centernet_draw_targets")
    return [heat, reg, wh, mask, ind]
