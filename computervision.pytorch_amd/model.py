"""``Yolo8``: the reference's model object contract on top of the MI355X engine.

What callers of the reference rely on and is kept here:
* ``nn.Module`` whose ``state_dict`` has the reference's 355 keys / shapes / order for scale "n"
  (``model.<i>...``; core/models/yolov8/yolo_v8.py:16-62), so reference ``.pth`` files load;
* ``model.model[-1]`` exposes ``stride``, ``nc``, ``no``, ``reg_max`` (read by the loss,
  core/algorithms/yolo_v8.py:32-45);
* ``model.train()`` forward returns three NCHW tensors ``(B, nc+64, H/8.., W/8..)``; ``model.eval()``
  forward returns ``(y (B, 4+nc, A), [those three])`` (core/models/yolov8/modules.py:428-446);
* random init equals the reference constructor's under the same global seed (weights drawn in module
  construction order, BN buffers after the constructor's stride probe, Detect.bias_init).

What differs underneath: the sub-modules hold no compute.  Every tensor is a strided view of three
flat arenas (parameters, gradients, BN statistics) and ``forward`` runs the whole graph through
``cvx_engine_forward``; autograd sees one custom Function whose backward calls ``cvx_engine_backward``.
There is no eager/CPU fallback: calling the model on a CPU tensor raises.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib as L
from .engine import Engine, decode
from .graph import REG_MAX, STRIDES, ParamLayout, build_yolov8_graph

BN_EPS, BN_MOMENTUM = 1e-3, 0.03       # core/models/yolov8/torch_utils.py:17-19


# ---- parameter-only module tree (names = the reference's) ---------------------------------------
class _Holder(nn.Module):
    """A module that only owns tensors; compute happens in the engine."""

    def forward(self, *a, **k):  # pragma: no cover
        raise L.CvxError(f"{type(self).__name__} has no standalone forward: the engine executes the whole graph (call the Yolo8 model)")


class Conv2dParams(_Holder):
    def __init__(self, c1, c2, k, bias):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = c1, c2, (k, k)
        self.weight = nn.Parameter(torch.empty(0))
        if bias:
            self.bias = nn.Parameter(torch.empty(0))
        else:
            self.register_parameter("bias", None)


class BatchNormParams(_Holder):
    def __init__(self, c):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, BN_EPS, BN_MOMENTUM
        self.weight = nn.Parameter(torch.empty(0))
        self.bias = nn.Parameter(torch.empty(0))
        self.register_buffer("running_mean", torch.empty(0))
        self.register_buffer("running_var", torch.empty(0))
        self.register_buffer("num_batches_tracked", torch.empty(0, dtype=torch.long))


class Conv(_Holder):                      # modules.py:19-33
    def __init__(self, c1, c2, k=1, s=1):
        super().__init__()
        self.conv = Conv2dParams(c1, c2, k, bias=False)
        self.bn = BatchNormParams(c2)
        self.stride = s


class Bottleneck(_Holder):                # modules.py:124-135
    def __init__(self, c, shortcut):
        super().__init__()
        self.cv1 = Conv(c, c, 3)
        self.cv2 = Conv(c, c, 3)
        self.add = shortcut


class C2f(_Holder):                       # modules.py:189-202
    def __init__(self, c1, c2, n, shortcut):
        super().__init__()
        self.c = c2 // 2
        self.cv1 = Conv(c1, 2 * self.c, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, shortcut) for _ in range(n))


class SPPF(_Holder):                      # modules.py:304-318
    def __init__(self, c1, c2):
        super().__init__()
        self.cv1 = Conv(c1, c1 // 2, 1)
        self.cv2 = Conv((c1 // 2) * 4, c2, 1)


class Upsample(_Holder):
    pass


class Concat(_Holder):
    pass


class DFL(_Holder):                       # modules.py:67-83
    def __init__(self, c1=REG_MAX):
        super().__init__()
        self.conv = Conv2dParams(c1, 1, 1, bias=False)
        self.conv.weight.requires_grad_(False)
        self.c1 = c1


class Detect(_Holder):                    # modules.py:407-455
    def __init__(self, nc, ch, c_box, c_cls):
        super().__init__()
        self.nc, self.nl, self.reg_max = nc, len(ch), REG_MAX
        self.no = nc + 4 * REG_MAX
        self.stride = torch.tensor([float(s) for s in STRIDES])
        self.cv2 = nn.ModuleList(nn.Sequential(Conv(x, c_box, 3), Conv(c_box, c_box, 3), Conv2dParams(c_box, 4 * REG_MAX, 1, True)) for x in ch)
        self.cv3 = nn.ModuleList(nn.Sequential(Conv(x, c_cls, 3), Conv(c_cls, c_cls, 3), Conv2dParams(c_cls, nc, 1, True)) for x in ch)
        self.dfl = DFL(REG_MAX)


class PredList(list):
    """The three NCHW level tensors, plus the fused (B, A, no) tensor they are views/copies of."""
    pred: Optional[torch.Tensor] = None
    level_hw = None


class _EngineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, anchor, model):
        ctx.model = model
        pred = model._run_forward(images, training=True)
        return pred

    @staticmethod
    def backward(ctx, gpred):
        model = ctx.model
        model._run_backward(gpred)
        return None, None, None


class Yolo8(nn.Module):
    def __init__(self, scale_name: str = "n", num_classes: int = 80, ch: int = 3, loss_scale: float = 1024.0):
        super().__init__()
        if ch != 3:
            raise ValueError("the MI355X engine is built for 3-channel images")
        lay = ParamLayout(scale_name, num_classes)
        self.layout = lay
        self.scale_name, self.num_classes = scale_name, num_classes
        self.loss_scale = float(loss_scale)
        c64, c128, c256, c512, c1024 = lay.c64, lay.c128, lay.c256, lay.c512, lay.c1024
        layers = [
            Conv(ch, c64, 3, 2), Conv(c64, c128, 3, 2), C2f(c128, c128, lay.n3, True), Conv(c128, c256, 3, 2),
            C2f(c256, c256, lay.n6, True), Conv(c256, c512, 3, 2), C2f(c512, c512, lay.n6, True), Conv(c512, c1024, 3, 2),
            C2f(c1024, c1024, lay.n3, True), SPPF(c1024, c1024),
            Upsample(), Concat(), C2f(c1024 + c512, c512, lay.n3, False),
            Upsample(), Concat(), C2f(c512 + c256, c256, lay.n3, False),
            Conv(c256, c256, 3, 2), Concat(), C2f(c256 + c512, c512, lay.n3, False),
            Conv(c512, c512, 3, 2), Concat(), C2f(c512 + c1024, c1024, lay.n3, False),
            Detect(num_classes, lay.head_in, lay.c_box, lay.c_cls),
        ]
        self.model = nn.Sequential(*layers)
        self.stride = self.model[-1].stride
        # ---- flat arenas; parameters / buffers become views of them ----
        self._flat = {
            "param": torch.zeros(lay.n_params),
            "grad": None,
            "stat": torch.zeros(lay.n_stats),
            "nbt": torch.zeros(lay.n_bn, dtype=torch.long),
        }
        self._anchor = torch.zeros(1, requires_grad=True)
        self._engines: Dict = {}
        self._grads_attached = False
        self._attach_views()
        self._init_like_reference()

    # ---- arenas <-> module tree -------------------------------------------------------------------
    def _named_slots(self):
        sd_names = dict(self.named_parameters(recurse=True))
        sd_names.update(dict(self.named_buffers(recurse=True)))
        return sd_names

    def _attach_views(self):
        lay = self.layout
        modules = dict(self.named_modules())
        i_bn = 0
        for key, slot in lay.slots.items():
            mod_name, attr = key.rsplit(".", 1)
            mod = modules[mod_name]
            arena = self._flat[slot.arena]
            view = torch.as_strided(arena, slot.shape, slot.strides, slot.offset)
            if slot.trainable:
                old = mod._parameters[attr]
                p = nn.Parameter(view, requires_grad=True if old is None else old.requires_grad)
                mod._parameters[attr] = p
            else:
                mod._buffers[attr] = view
        # num_batches_tracked: scalar views of one int64 arena, in BN registration order
        for name, mod in modules.items():
            if isinstance(mod, BatchNormParams):
                mod._buffers["num_batches_tracked"] = self._flat["nbt"][i_bn]
                i_bn += 1
        dfl = self.model[-1].dfl.conv
        if dfl.weight.numel() == 0:
            dfl._parameters["weight"] = nn.Parameter(torch.arange(REG_MAX, dtype=torch.float32).view(1, REG_MAX, 1, 1), requires_grad=False)
        self._grads_attached = False

    def _apply(self, fn, recurse=True):
        """Move / cast the ARENAS, then rebuild every parameter and buffer as a view of them."""
        for k in ("param", "stat", "nbt", "grad"):
            if self._flat[k] is not None:
                t = fn(self._flat[k])
                if k == "nbt":
                    t = t.long()
                elif t.dtype != torch.float32:
                    raise L.CvxError("the engine keeps fp32 master parameters; half()/bfloat16() are not supported (compute is fp16 inside)")
                self._flat[k] = t.contiguous()
        dfl = self.model[-1].dfl.conv
        dfl._parameters["weight"] = nn.Parameter(fn(dfl.weight.data), requires_grad=False)
        self.model[-1].stride = fn(self.model[-1].stride)
        self.stride = self.model[-1].stride
        self._anchor = fn(self._anchor.detach()).requires_grad_(True)
        self._attach_views()
        self._engines.clear()
        return self

    def _init_like_reference(self):
        """Same draws, same order, as ``get_yolo8_*`` under the caller's global seed
        (yolo_v8.py:26-62 + torch's Conv2d.reset_parameters), then the constructor's side effects."""
        det = self.model[-1]
        with torch.no_grad():
            for mod in self.modules():
                if isinstance(mod, Conv2dParams):
                    k = mod.kernel_size[0]
                    w = torch.empty(mod.out_channels, mod.in_channels, k, k)
                    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
                    if mod.weight.requires_grad:
                        mod.weight.copy_(w)
                    if mod.bias is not None:
                        bound = 1.0 / math.sqrt(mod.in_channels * k * k)
                        b = torch.empty(mod.out_channels)
                        nn.init.uniform_(b, -bound, bound)
                        mod.bias.copy_(b)
                elif isinstance(mod, BatchNormParams):
                    mod.weight.fill_(1.0)
                    mod.bias.zero_()
                    mod.running_mean.zero_()
                    mod.running_var.fill_(0.9)       # stride-probe forward on zeros, momentum 0.1 (yolo_v8.py:53-58)
            self._flat["nbt"].fill_(1)
            for a, b, s in zip(det.cv2, det.cv3, STRIDES):           # Detect.bias_init, modules.py:448-455
                a[-1].bias.fill_(1.0)
                b[-1].bias[:det.nc] = math.log(5 / det.nc / (640 / s) ** 2)

    # ---- engine plumbing ------------------------------------------------------------------------------
    @property
    def flat_params(self) -> torch.Tensor:
        return self._flat["param"]

    @property
    def flat_stats(self) -> torch.Tensor:
        return self._flat["stat"]

    @property
    def flat_grads(self) -> torch.Tensor:
        if self._flat["grad"] is None or self._flat["grad"].device != self._flat["param"].device:
            self._flat["grad"] = torch.zeros_like(self._flat["param"])
            self._grads_attached = False
        return self._flat["grad"]

    def engine_for(self, h: int, w: int) -> Engine:
        dev = self._flat["param"].device
        key = (h, w, dev)
        eng = self._engines.get(key)
        if eng is None:
            if dev.type != "cuda":
                raise L.CvxError("Yolo8 runs on an MI355X only: move the model with .to('cuda') first (there is no CPU fallback)")
            eng = Engine(build_yolov8_graph(self.layout, h, w), dev)
            eng.set_bn(BN_EPS, BN_MOMENTUM)
            self._engines[key] = eng
        eng.bind(self._flat["param"], self.flat_grads, self._flat["stat"])
        return eng

    def _run_forward(self, images: torch.Tensor, training: bool, pred: Optional[torch.Tensor] = None) -> torch.Tensor:
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError("expected images of shape (B, 3, H, W)")
        eng = self.engine_for(int(images.shape[2]), int(images.shape[3]))
        self._last_engine = eng
        out = eng.forward(images, training, pred)          # (B, A, no_pad): class columns padded to a multiple of 8
        if training:
            self._flat["nbt"] += 1
        no = self.layout.no
        return out if out.shape[2] == no else out[..., :no]  # the reference's (B, A, 64 + nc) as a view of the padded rows

    def attach_grads(self):
        """Make ``p.grad`` of every parameter a view of the flat gradient arena (torch optimisers / GradScaler)."""
        g = self.flat_grads
        modules = dict(self.named_modules())
        for key, slot in self.layout.slots.items():
            if not slot.trainable:
                continue
            mod_name, attr = key.rsplit(".", 1)
            modules[mod_name]._parameters[attr].grad = torch.as_strided(g, slot.shape, slot.strides, slot.offset)
        self._grads_attached = True

    def _run_backward(self, gpred: torch.Tensor):
        first = next(p for p in self.parameters() if p.requires_grad)
        if first.grad is None:               # optimizer.zero_grad(set_to_none=True) happened (or first step)
            self.flat_grads.zero_()
            self._grads_attached = False
        dpred = (gpred * self.loss_scale).to(torch.float16)
        if self.layout.no_pad != self.layout.no:             # the engine reads rows of no_pad values; the padding is zero
            padded = torch.zeros(*dpred.shape[:2], self.layout.no_pad, dtype=torch.float16, device=dpred.device)
            padded[..., :self.layout.no] = dpred
            dpred = padded
        self._last_engine.backward(dpred.contiguous(), self.loss_scale)
        if not self._grads_attached or first.grad is None:
            self.attach_grads()

    def level_shapes(self, h: int, w: int):
        return [(h // s, w // s) for s in STRIDES]

    # ---- the reference's forward contract --------------------------------------------------------------
    def forward(self, x: torch.Tensor):
        B, _, H, W = x.shape
        det = self.model[-1]
        if self.training and torch.is_grad_enabled():
            pred = _EngineFn.apply(x, self._anchor, self)
        else:
            pred = self._run_forward(x, training=self.training)
        outs = PredList()
        off = 0
        for (h, w) in self.level_shapes(H, W):
            outs.append(pred[:, off:off + h * w, :].permute(0, 2, 1).reshape(B, det.no, h, w))
            off += h * w
        outs.pred = pred
        outs.level_hw = self.level_shapes(H, W)
        if self.training:
            return outs
        y = decode(pred.detach(), det.nc, outs.level_hw, STRIDES)
        return y, outs


def get_yolo8(model_type: str, nc: int = 80, **kw) -> Yolo8:
    return Yolo8(model_type, nc, **kw)
