"""DeepLabv3+ (ResNet-101, output stride 16) on the MI355X engine: inference AND training (SURVEY.md section 8 row a19 / (f)2,
BASELINE.json configs[5]).

Mirrors ``core/models/deeplabv3plus.py:10-149`` + ``core/models/resnet.py:82-277`` of the reference as an engine graph:

* every Conv + BatchNorm (+ ReLU) is one convolution launch with the running statistics folded into its epilogue; a
  Bottleneck's ``relu(bn3(conv3) + identity)`` is the same launch (residual added before the activation); the
  downsample branch is Conv + BN without activation;
* layer4 runs dilated (``replace_stride_with_dilation=[False, False, True]``, resnet.py:212-220): its first block keeps
  dilation 1 at stride 1, the other two use dilation 2;
* ASPP (deeplabv3plus.py:43-75): the five branches write into the channel slices of ONE 1280-channel buffer (no
  ``torch.cat``); the three atrous 3x3 convolutions (rates 6, 12, 18) are tap tables of the generic implicit-GEMM kernel; the
  pooling branch is a global average pool, a 1x1 convolution on a 1x1 map and a bilinear resize from 1x1 (= broadcast);
  ``Dropout(0.1)`` is the identity in eval mode;
* decoder (deeplabv3plus.py:78-123): the 48-channel low-level projection and the bilinearly resized ASPP output (33 -> 129
  at 513 input, ``align_corners=False``) land in the two slices of one 304-channel buffer; the classifier's 1x1 (+bias)
  writes fp32 logits rows, which one last kernel resizes to the input resolution as the reference's NCHW tensor.

All parameters live in one flat fp32 arena, BN statistics in a second one; ``state_dict`` has the reference's 674 keys and
shapes in its order and is bit-identical to ``DeeplabV3Plus(num_classes, 16, pretrained_backbone=False)`` under the same
global seed.

Training (``model.train()``): the same graph with batch-statistics BatchNorm (bn_act.hip: ReLU / linear passes, the Bottleneck
residual inside the activation), ``Dropout(0.1)`` as an engine op with a counter-based mask, the backward of every op
(data gradients incl. the dilated and the 1x1 stride-2 convolutions, weight gradients, 3x3 max pool, global average pool,
bilinear resizes) and ``SegLoss`` (csrc/loss_seg.hip: upsampling + focal / cross-entropy loss + gradient down to the logits rows
in three launches).  ``SegTrainStep`` is the reference's ``train_loop`` (segmentation_trainer.py:114-131) as C-ABI calls.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib as L
from .engine import Engine
from .graph import Graph, TensorSlot

LAYERS = (3, 4, 23, 3)                            # resnet101 (resnet.py:272-277)
PLANES = (64, 128, 256, 512)
ASPP_RATES = (6, 12, 18)                          # output stride 16 (deeplabv3plus.py:134-136)
ASPP_OUT = 256
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def _blocks():
    """ResNet._make_layer (resnet.py:190-232) for resnet101 with replace_stride_with_dilation = [False, False, True] as data:
    [(prefix, inplanes, width, outplanes, stride, dilation, has_downsample)] in module order."""
    out, inplanes, dilation = [], 64, 1
    for li, (planes, n, stride, dilate) in enumerate(zip(PLANES, LAYERS, (1, 2, 2, 2), (False, False, False, True))):
        prev = dilation
        if dilate:
            dilation *= stride
            stride = 1
        for b in range(n):
            first = b == 0
            out.append(dict(prefix=f"backbone.layer{li + 1}.{b}", cin=inplanes, width=planes, cout=planes * 4, stride=stride if first else 1,
                            dil=prev if first else dilation, down=first and (stride != 1 or inplanes != planes * 4), layer=li + 1))
            inplanes = planes * 4
    return out


class DeepLabLayout:
    """Arena offsets for every tensor of the reference's DeeplabV3Plus ``state_dict`` (same keys, shapes, order)."""

    def __init__(self, nc: int = 21):
        self.nc = nc
        self.nc_pad = (nc + 7) & ~7
        self.slots: "OrderedDict[str, TensorSlot]" = OrderedDict()
        self.nbt_keys: List[str] = []
        self.convs: Dict[str, dict] = {}
        self.blocks = _blocks()
        self._p = self._s = 0
        self._plan()
        self.n_params = (self._p + 3) & ~3
        self.n_stats = (self._s + 3) & ~3

    def _take(self, arena, n):
        if arena == "param":
            off, self._p = self._p, (self._p + n + 3) & ~3
        else:
            off, self._s = self._s, (self._s + n + 3) & ~3
        return off

    def conv(self, key, cout, cin, k, bias=False):
        ce = (cout + 7) & ~7
        spec = dict(cout=cout, cout_eng=ce, cin=cin, k=k, w_off=self._take("param", ce * k * k * cin))
        self.slots[key + ".weight"] = TensorSlot("param", spec["w_off"], (cout, cin, k, k), (k * k * cin, 1, k * cin, cin))
        if bias:
            spec["bias_off"] = self._take("param", ce)
            self.slots[key + ".bias"] = TensorSlot("param", spec["bias_off"], (cout,), (1,))
        self.convs[key] = spec
        return spec

    def bn(self, key, c, spec):
        spec.update(gamma_off=self._take("param", c), beta_off=self._take("param", c), rmean_off=self._take("stat", c),
                    rvar_off=self._take("stat", c))
        self.slots[key + ".weight"] = TensorSlot("param", spec["gamma_off"], (c,), (1,))
        self.slots[key + ".bias"] = TensorSlot("param", spec["beta_off"], (c,), (1,))
        self.slots[key + ".running_mean"] = TensorSlot("stat", spec["rmean_off"], (c,), (1,), False)
        self.slots[key + ".running_var"] = TensorSlot("stat", spec["rvar_off"], (c,), (1,), False)
        self.slots[key + ".num_batches_tracked"] = TensorSlot("nbt", len(self.nbt_keys), (), (), False)
        self.nbt_keys.append(key + ".num_batches_tracked")

    def conv_bn(self, ckey, bkey, cout, cin, k):
        self.bn(bkey, cout, self.conv(ckey, cout, cin, k))

    def _plan(self):
        """state_dict order = module registration order (a Bottleneck's downsample comes after its bn3, resnet.py:118-119)."""
        self.conv_bn("backbone.conv1", "backbone.bn1", 64, 3, 7)
        for b in self.blocks:
            p = b["prefix"]
            self.conv_bn(p + ".conv1", p + ".bn1", b["width"], b["cin"], 1)
            self.conv_bn(p + ".conv2", p + ".bn2", b["width"], b["width"], 3)
            self.conv_bn(p + ".conv3", p + ".bn3", b["cout"], b["width"], 1)
            if b["down"]:
                self.conv_bn(p + ".downsample.0", p + ".downsample.1", b["cout"], b["cin"], 1)
        c = "classifier."
        self.conv_bn(c + "project.0", c + "project.1", 48, 256, 1)
        self.conv_bn(c + "aspp.convs.0.0", c + "aspp.convs.0.1", ASPP_OUT, 2048, 1)
        for i in range(3):
            self.conv_bn(c + f"aspp.convs.{i + 1}.0", c + f"aspp.convs.{i + 1}.1", ASPP_OUT, 2048, 3)
        self.conv_bn(c + "aspp.convs.4.1", c + "aspp.convs.4.2", ASPP_OUT, 2048, 1)
        self.conv_bn(c + "aspp.project.0", c + "aspp.project.1", ASPP_OUT, 5 * ASPP_OUT, 1)
        self.conv_bn(c + "classifier.0", c + "classifier.1", 256, 304, 3)
        self.conv(c + "classifier.3", self.nc, 256, 1, bias=True)

    # the order in which the reference CONSTRUCTS its convolutions (each draws its default init from the global RNG):
    # _make_layer builds a layer's downsample before its first block (resnet.py:205-216)
    def construction_order(self, part):
        keys = []
        if part == "backbone":
            keys.append("backbone.conv1")
            for b in self.blocks:
                p = b["prefix"]
                if b["down"]:
                    keys.append(p + ".downsample.0")
                keys += [p + ".conv1", p + ".conv2", p + ".conv3"]
        else:
            c = "classifier."
            keys += [c + "project.0", c + "aspp.convs.0.0", c + "aspp.convs.1.0", c + "aspp.convs.2.0", c + "aspp.convs.3.0", c + "aspp.convs.4.1",
                     c + "aspp.project.0", c + "classifier.0", c + "classifier.3"]
        return keys

    def module_order(self, part):
        return [k for k in self.convs if k.startswith(part)]

    def views(self, arena, which="param"):
        return {k: torch.as_strided(arena, sl.shape, sl.strides, sl.offset) for k, sl in self.slots.items() if sl.arena == which}


def conv_out(n, k, stride, pad, dil):
    return (n + 2 * pad - dil * (k - 1) - 1) // stride + 1


def build_deeplab_graph(lay: DeepLabLayout, H: int, W: int, dropout_p: float = 0.1) -> Graph:
    """Buffer plan + op list for an (H, W) input (any size from 33 up: the reference runs 513 x 513).  Every convolution but the
    first carries its data gradient, so the one graph serves eval and training forwards."""
    if H < 33 or W < 33:
        raise ValueError("input height/width must be at least 33")
    g = Graph()

    def buf(h, w, ch, kind=L.BUF_ACT_F16):
        g.bufs.append((h, w, ch, kind))
        return len(g.bufs) - 1

    def V(b, off, ch, pix=0):
        return (b, off, ch, pix)

    def conv(ckey, vin, vout, hin, win, stride=1, dil=1, act=L.ACT_BN_RELU, res=None):
        s = lay.convs[ckey]
        k = s["k"]
        pad = dil * (k // 2)
        ho, wo = conv_out(hin, k, stride, pad, dil), conv_out(win, k, stride, pad, dil)
        op = dict(type=L.OP_CONV, name=ckey, out=vout, ih=hin, iw=win, oh=ho, ow=wo, k=k, stride=stride, pad=pad, dil=dil, act=act,
                  needs_dgrad=0 if ckey == "backbone.conv1" else 1,
                  w_cin=s["cin"], w_off=s["w_off"], gamma_off=s.get("gamma_off", 0), beta_off=s.get("beta_off", 0), bias_off=s.get("bias_off", 0),
                  rmean_off=s.get("rmean_off", 0), rvar_off=s.get("rvar_off", 0), flags=L.OPF_RES_PRE_ACT if res is not None else 0)
        op["in"] = vin
        if res is not None:
            op["res"] = res
        g.ops.append(op)
        return ho, wo

    def simple(kind, name, vin, vout, ih, iw, oh, ow):
        op = dict(type=kind, name=name, out=vout, ih=ih, iw=iw, oh=oh, ow=ow)
        op["in"] = vin
        g.ops.append(op)

    img = buf(H, W, 8)
    g.image_buf = img
    h1, w1 = conv_out(H, 7, 2, 3, 1), conv_out(W, 7, 2, 3, 1)
    c1 = buf(h1, w1, 64)
    conv("backbone.conv1", V(img, 0, 8), V(c1, 0, 64), H, W, stride=2)
    h, w = (h1 - 1) // 2 + 1, (w1 - 1) // 2 + 1
    cur = V(buf(h, w, 64), 0, 64)
    simple(L.OP_MAXPOOL3S2, "backbone.maxpool", V(c1, 0, 64), cur, h1, w1, h, w)
    low = None
    for b in lay.blocks:                                         # Bottleneck.forward (resnet.py:121-143)
        p, s, d = b["prefix"], b["stride"], b["dil"]
        ho, wo = conv_out(h, 3, s, d, d), conv_out(w, 3, s, d, d)
        t1, t2 = V(buf(h, w, b["width"]), 0, b["width"]), V(buf(ho, wo, b["width"]), 0, b["width"])
        out = V(buf(ho, wo, b["cout"]), 0, b["cout"])
        conv(p + ".conv1", cur, t1, h, w)
        conv(p + ".conv2", t1, t2, h, w, stride=s, dil=d)
        if b["down"]:
            ident = V(buf(ho, wo, b["cout"]), 0, b["cout"])
            conv(p + ".downsample.0", cur, ident, h, w, stride=s, act=L.ACT_BN_LINEAR)
        else:
            ident = cur
        conv(p + ".conv3", t2, out, ho, wo, res=ident)
        cur, h, w = out, ho, wo
        if b["layer"] == 1:
            low, lh, lw = cur, h, w                              # "low_level": the output of layer1 (resnet.py:245-246)

    # ASPP (deeplabv3plus.py:43-75) -> five slices of one buffer
    c = "classifier."
    cat = buf(h, w, 5 * ASPP_OUT)
    conv(c + "aspp.convs.0.0", cur, V(cat, 0, ASPP_OUT), h, w)
    for i, r in enumerate(ASPP_RATES):
        conv(c + f"aspp.convs.{i + 1}.0", cur, V(cat, (i + 1) * ASPP_OUT, ASPP_OUT), h, w, dil=r)
    pooled, pconv = V(buf(1, 1, 2048), 0, 2048), V(buf(1, 1, ASPP_OUT), 0, ASPP_OUT)
    simple(L.OP_AVGPOOL, c + "aspp.convs.4.0", cur, pooled, h, w, 1, 1)
    conv(c + "aspp.convs.4.1", pooled, pconv, 1, 1)
    simple(L.OP_RESIZE, c + "aspp.convs.4.up", pconv, V(cat, 4 * ASPP_OUT, ASPP_OUT), 1, 1, h, w)
    proj = V(buf(h, w, ASPP_OUT), 0, ASPP_OUT)
    conv(c + "aspp.project.0", V(cat, 0, 5 * ASPP_OUT), proj, h, w)
    aspp = V(buf(h, w, ASPP_OUT), 0, ASPP_OUT)
    g.ops.append(dict(type=L.OP_DROPOUT, name=c + "aspp.project.3", out=aspp, ih=h, iw=w, oh=h, ow=w, k=int(round(dropout_p * 65536))))
    g.ops[-1]["in"] = proj                                        # nn.Dropout(0.1) (deeplabv3plus.py:67): identity in eval
    # decoder (deeplabv3plus.py:113-123): [low-level 48 | resized ASPP 256]
    dec = buf(lh, lw, 48 + ASPP_OUT)
    conv(c + "project.0", low, V(dec, 0, 48), lh, lw)
    simple(L.OP_RESIZE, c + "aspp.up", aspp, V(dec, 48, ASPP_OUT), h, w, lh, lw)
    hid = V(buf(lh, lw, 256), 0, 256)
    conv(c + "classifier.0", V(dec, 0, 48 + ASPP_OUT), hid, lh, lw)
    pred = buf(lh * lw, 1, lay.nc_pad, L.BUF_PRED_F32)
    g.pred_buf = pred
    conv(c + "classifier.3", hid, V(pred, 0, lay.nc_pad, 0), lh, lw, act=L.ACT_BIAS)
    g.level_hw = [(lh, lw)]
    g.anchors = lh * lw
    return g


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise L.CvxError("parameter holder: the engine executes the whole graph (call the DeepLabv3+ model)")


class DeepLabV3PlusR101(nn.Module):
    """``DeeplabV3Plus(num_classes, output_stride=16, pretrained_backbone=False)`` of the reference (deeplabv3plus.py:126-149)
    on the engine: ``model(x)`` returns the (B, num_classes, H, W) fp32 logits; in training mode (grad enabled) the tensor is
    connected to the engine's backward pass, and carries the low-resolution rows as ``.rows`` for the fused ``SegLoss``."""

    def __init__(self, num_classes: int = 21, loss_scale: float = 65536.0, dropout_p: float = 0.1):
        super().__init__()
        self.layout = lay = DeepLabLayout(num_classes)
        self.num_classes = num_classes
        self.loss_scale = float(loss_scale)          # torch.cuda.amp.GradScaler's initial scale (the reference trains under AMP)
        self.dropout_p = float(dropout_p)            # aspp.project.3 = nn.Dropout(0.1); 0 disables it (parity tests)
        self.seed = 0                                # seed of the dropout masks (cvx_engine_set_seed)
        self._flat = {"param": torch.zeros(lay.n_params), "stat": torch.zeros(lay.n_stats), "nbt": torch.zeros(len(lay.nbt_keys), dtype=torch.long),
                      "grad": None}
        self._anchor = torch.zeros(1, requires_grad=True)
        self._grads_attached = False
        self._engines: Dict = {}
        self._build_tree()
        self._attach_views()
        self._init_like_reference()
        self.last_rows = None

    def _build_tree(self):
        for key in self.layout.slots:
            mod = self
            for name in key.split(".")[:-1]:
                if name not in mod._modules:
                    mod.add_module(name, _Holder())
                mod = mod._modules[name]

    def _attach_views(self):
        for key, sl in self.layout.slots.items():
            mod = self
            parts = key.split(".")
            for name in parts[:-1]:
                mod = mod._modules[name]
            if sl.arena == "nbt":
                mod._buffers[parts[-1]] = self._flat["nbt"][sl.offset]
                continue
            view = torch.as_strided(self._flat[sl.arena], sl.shape, sl.strides, sl.offset)
            if sl.trainable:
                old = mod._parameters.get(parts[-1])
                mod._parameters[parts[-1]] = nn.Parameter(view, requires_grad=True if old is None else old.requires_grad)
            else:
                mod._buffers[parts[-1]] = view

    def _apply(self, fn, recurse=True):
        self._flat["grad"] = None
        self._grads_attached = False
        self._anchor = fn(self._anchor.detach()).requires_grad_(True)
        for k in ("param", "stat", "nbt"):
            t = fn(self._flat[k])
            if k != "nbt" and t.dtype != torch.float32:
                raise L.CvxError("the engine keeps fp32 master parameters; half()/bfloat16() are not supported (compute is fp16 inside)")
            self._flat[k] = t.long().contiguous() if k == "nbt" else t.contiguous()
        self._attach_views()
        self._engines.clear()
        return self

    def _init_like_reference(self):
        """The reference's RNG consumption, draw for draw (resnet.py:150-178, deeplabv3plus.py:99-110): every nn.Conv2d draws its
        default init when it is constructed (weight, then bias) -- a layer's downsample before its first block --; then ResNet
        re-initialises its convolutions with kaiming_normal_(fan_out, relu) in module order; then the head is constructed and
        re-initialises its convolution WEIGHTS with kaiming_normal_ (fan_in) in module order: the classifier's bias keeps the
        value drawn at construction."""
        lay = self.layout
        sd = dict(self.named_parameters())
        sd.update(dict(self.named_buffers()))

        def default_init(key):
            sl = lay.slots[key + ".weight"]
            w = torch.empty(sl.shape)
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))
            sd[key + ".weight"].copy_(w)
            if "bias_off" in lay.convs[key]:
                bound = 1.0 / math.sqrt(sl.shape[1] * sl.shape[2] * sl.shape[3])
                bb = torch.empty(lay.slots[key + ".bias"].shape)
                nn.init.uniform_(bb, -bound, bound)
                sd[key + ".bias"].copy_(bb)

        with torch.no_grad():
            for key in lay.construction_order("backbone"):
                default_init(key)
            for key in lay.module_order("backbone"):
                w = torch.empty(lay.slots[key + ".weight"].shape)
                nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
                sd[key + ".weight"].copy_(w)
            for key in lay.construction_order("classifier"):
                default_init(key)
            for key in lay.module_order("classifier"):
                w = torch.empty(lay.slots[key + ".weight"].shape)
                nn.init.kaiming_normal_(w)
                sd[key + ".weight"].copy_(w)
            for key, sl in lay.slots.items():
                stem, leaf = key.rsplit(".", 1)
                if (stem + ".running_mean") in lay.slots:
                    if leaf in ("weight", "running_var"):
                        sd[key].fill_(1.0)
                    elif leaf in ("bias", "running_mean"):
                        sd[key].zero_()
            self._flat["nbt"].zero_()

    # ---- engine plumbing ---------------------------------------------------------------------------------
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
        key = (h, w, dev, self.dropout_p)
        eng = self._engines.get(key)
        if eng is None:
            if dev.type != "cuda":
                raise L.CvxError("DeepLabV3PlusR101 runs on an MI355X only: move the model with .to('cuda') first (there is no CPU fallback)")
            eng = Engine(build_deeplab_graph(self.layout, h, w, self.dropout_p), dev)
            eng.set_bn(BN_EPS, BN_MOMENTUM)
            self._engines[key] = eng
        eng.bind(self._flat["param"], self.flat_grads if self.training else self._flat["grad"], self._flat["stat"])
        eng.set_seed(self.seed)
        return eng

    def _run_forward(self, x: torch.Tensor, training: bool, pred: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(B,3,H,W) -> the engine's fp32 logits rows (B, h*w, nc_pad) at the decoder's resolution (stride 4)."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected images of shape (B, 3, H, W)")
        if training and x.shape[0] < 2:
            raise ValueError("training-mode BatchNorm needs more than one value per channel (ASPPPooling's 1x1 map): batch >= 2")
        eng = self.engine_for(int(x.shape[2]), int(x.shape[3]))
        self._last_engine = eng
        rows = eng.forward(x, training, pred)
        if training:
            self._flat["nbt"] += 1
        return rows

    def forward_rows(self, x: torch.Tensor) -> torch.Tensor:
        return self._run_forward(x, self.training)

    def rows_to_nchw(self, rows: torch.Tensor, H: int, W: int) -> torch.Tensor:
        """F.interpolate(classifier(features), size=(H, W), mode="bilinear", align_corners=False) (deeplabv3plus.py:147)."""
        B = rows.shape[0]
        lh, lw = self._last_engine.graph.level_hw[0]
        out = torch.empty(B, self.num_classes, H, W, dtype=torch.float32, device=rows.device)
        lib = L.load()
        L.check(lib.cvx_resize_bilinear_rows_to_nchw(L.ptr(rows), self.layout.nc_pad, B, self.num_classes, lh, lw, H, W, L.ptr(out),
                                                      L.stream_ptr(rows.device)), "cvx_resize_bilinear_rows_to_nchw")
        return out

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

    def backward_rows(self, dpred_f16: torch.Tensor, loss_scale: float):
        """Engine backward from loss_scale * dLoss/drows (B, h*w, nc_pad) fp16: parameter gradients are accumulated into the arena."""
        first = next(p for p in self.parameters() if p.requires_grad)
        if first.grad is None:               # optimizer.zero_grad(set_to_none=True) happened (or first step)
            self.flat_grads.zero_()
            self._grads_attached = False
        self._last_engine.backward(dpred_f16, loss_scale)
        if not self._grads_attached or first.grad is None:
            self.attach_grads()

    def _backward_from_nchw(self, g: torch.Tensor):
        """Gradient w.r.t. the full-resolution logits -> rows (adjoint of the final resize) -> engine backward."""
        B, nc, H, W = g.shape
        lh, lw = self._last_engine.graph.level_hw[0]
        dpred = torch.empty(B, lh * lw, self.layout.nc_pad, dtype=torch.float16, device=g.device)
        lib = L.load()
        L.check(lib.cvx_resize_bilinear_nchw_grad_to_rows(L.ptr(g.contiguous().float()), B, nc, lh, lw, H, W, self.loss_scale, L.ptr(dpred),
                                                           self.layout.nc_pad, L.stream_ptr(g.device)), "cvx_resize_bilinear_nchw_grad_to_rows")
        self.backward_rows(dpred, self.loss_scale)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, _, H, W = x.shape
        if self.training and torch.is_grad_enabled():
            out = _SegFn.apply(x, self._anchor, self)
        else:
            self.last_rows = self._run_forward(x, self.training)
            out = self.rows_to_nchw(self.last_rows, H, W)
        out.rows, out.model = self.last_rows, self     # the fused loss starts from the low-resolution rows
        return out


class _SegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, anchor, model):
        ctx.model = model
        ctx.set_materialize_grads(False)      # the fused SegLoss hands no gradient to the logits: then there is nothing to do here
        rows = model._run_forward(images, training=True)
        model.last_rows = rows
        return model.rows_to_nchw(rows, int(images.shape[2]), int(images.shape[3]))

    @staticmethod
    def backward(ctx, g):
        if g is not None:
            ctx.model._backward_from_nchw(g)
        return None, None, None


class _SegLossFn(torch.autograd.Function):
    """loss = SegLoss(rows, target) with the gradient taken by the fused kernel; ``.backward()`` runs the engine's backward."""

    @staticmethod
    def forward(ctx, logits, owner, rows, model, target, hw):
        loss, dpred = owner.op(rows, target, hw, model.loss_scale)
        ctx.model, ctx.dpred = model, dpred
        model.last_dpred = dpred
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        model = ctx.model
        # d loss is a scalar factor on dpred: fold it into the loss scale the engine divides by (1.0 for loss.backward())
        model.backward_rows(ctx.dpred, model.loss_scale / float(gout))
        return None, None, None, None, None, None


class SegLoss:
    """``FocalLoss()`` / ``nn.CrossEntropyLoss(reduction="mean")`` of ``DeeplabV3PlusA.build_loss`` (segmentation_2d.py:59-64,
    core/loss/focal_loss.py:6-22) on the engine: ``loss = criterion(preds, targets)`` with ``preds = model(images)``.  Value and
    gradient come from ``cvx_seg_loss`` on the low-resolution rows behind ``preds`` (no autograd tape through the upsampling);
    ``loss.backward()`` then runs the engine's backward pass."""

    def __init__(self, loss_type: str = "focal", alpha: float = 0.25, gamma: float = 2.0, ignore_index: int = -100, check_targets: bool = True):
        if loss_type not in ("focal", "ce"):
            raise ValueError("loss_type must be 'focal' or 'ce' (segmentation_2d.py:59-64)")
        self.mode = 0 if loss_type == "focal" else 1
        self.alpha, self.gamma, self.ignore_index = float(alpha), float(gamma), int(ignore_index)
        self.check_targets = check_targets           # False: skip the host check of the bad-label flag (no sync in the step)
        self._ws = None
        self._dpred = None
        self._bad = None

    def bad_targets(self) -> bool:
        """True when the last call saw a label that is neither ignore_index nor a class (synchronises)."""
        return self._bad is not None and int(self._bad.item()) != 0

    def op(self, rows: torch.Tensor, target: torch.Tensor, hw, loss_scale: float, dpred: Optional[torch.Tensor] = None,
           check: Optional[bool] = None):
        """rows (B, lh*lw, ld) fp32, target (B, H, W) int64, hw = (lh, lw) -> (loss (1,), dpred (B, lh*lw, ld) fp16)."""
        if rows.device.type != "cuda":
            raise L.CvxError("SegLoss runs on an MI355X only (there is no CPU path)")
        lib = L.load()
        B, A, ld = rows.shape
        lh, lw = hw
        if target.dim() != 3 or target.shape[0] != B:
            raise ValueError("targets must have shape (B, H, W)")
        H, W = int(target.shape[1]), int(target.shape[2])
        nc = getattr(self, "nc", None) or ld
        target = target.to(device=rows.device, dtype=torch.long).contiguous()
        need = int(lib.cvx_seg_loss_workspace_bytes(B, nc, H, W))
        if self._ws is None or self._ws.numel() < need or self._ws.device != rows.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=rows.device)
            self._bad = torch.zeros(1, dtype=torch.int32, device=rows.device)
        if dpred is None:
            dpred = torch.empty(B, A, ld, dtype=torch.float16, device=rows.device)
        loss = torch.empty(1, device=rows.device)
        L.check(lib.cvx_seg_loss(L.ptr(rows), ld, B, nc, lh, lw, H, W, L.ptr(target), self.mode, self.alpha, self.gamma, self.ignore_index,
                                 float(loss_scale), L.ptr(loss), L.ptr(dpred), L.ptr(self._bad), L.ptr(self._ws), L.stream_ptr(rows.device)),
                "cvx_seg_loss")
        if (self.check_targets if check is None else check) and self.bad_targets():
            raise L.CvxError("SegLoss: a target label is neither ignore_index nor in [0, num_classes)")
        return loss, dpred

    def __call__(self, preds: torch.Tensor, targets: torch.Tensor):
        rows, model = getattr(preds, "rows", None), getattr(preds, "model", None)
        if rows is None or model is None:
            raise L.CvxError("SegLoss needs the output of DeepLabV3PlusR101.forward (it carries the logits rows the loss starts from)")
        self.nc = model.num_classes
        hw = model._last_engine.graph.level_hw[0]
        if model.training and torch.is_grad_enabled():
            return _SegLossFn.apply(preds, self, rows, model, targets, hw)
        return self.op(rows, targets, hw, model.loss_scale)[0].reshape(())


class SegTrainStep:
    """One optimisation step of the reference's ``DeeplabV3PlusTrainer.train_loop`` (segmentation_trainer.py:114-131:
    zero_grad -> forward -> criterion -> backward -> Adam under AMP) as C-ABI calls: engine forward (training), ``cvx_seg_loss``,
    engine backward, [gradient all-reduce over RCCL], fused Adam with the inf/nan check of GradScaler.step.  Returns the loss (1,).
    Data parallel: the flat gradient arena is summed over the ranks in a few large slices on a side stream after the backward
    pass (the mean's 1/world is folded into Adam); BatchNorm statistics stay per rank, as in the reference."""

    def __init__(self, model: DeepLabV3PlusR101, criterion: SegLoss, optimizer, scaler=None, process_group=None, n_buckets: int = 4):
        self.model, self.criterion, self.optimizer, self.scaler = model, criterion, optimizer, scaler
        self.pg, self.n_buckets = process_group, n_buckets
        self.world, self.distributed = 1, False
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
            self.distributed = True
        self._pred = self._dpred = self._side = None

    def __call__(self, images: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        from .engine import check_finite
        m, crit = self.model, self.criterion
        if not m.training:
            raise L.CvxError("SegTrainStep: call model.train() first")
        dev = m.flat_params.device
        B, _, H, W = images.shape
        self.optimizer.sync_lr()
        eng = m.engine_for(H, W)
        lh, lw = eng.graph.level_hw[0]
        if self._pred is None or self._pred.shape[0] != B or self._pred.shape[1] != lh * lw:
            self._pred = torch.empty(B, lh * lw, m.layout.nc_pad, device=dev)
            self._dpred = torch.empty(B, lh * lw, m.layout.nc_pad, device=dev, dtype=torch.float16)
        scale = self.scaler.begin_step() if self.scaler is not None else m.loss_scale
        rows = m._run_forward(images, True, self._pred)
        crit.nc = m.num_classes
        loss, dpred = crit.op(rows, targets, (lh, lw), scale, self._dpred, check=False)   # no host sync in the step: crit.bad_targets() polls
        if self.distributed and dev.type == "cuda":               # gradient exchange overlapped with the backward pass, bucket by bucket
            if self._side is None:
                from .train import OverlappedExchange
                self._side = OverlappedExchange(self.pg, self.n_buckets)
            self._side.backward(eng, m.flat_grads, dpred, scale)
        else:
            eng.backward(dpred, scale)
        if self.scaler is not None:
            check_finite(m.flat_grads, self.scaler.found_inf)
            self.optimizer.found_inf = self.scaler.found_inf
        self.optimizer.step(zero_grad=True, grad_scale=1.0 / self.world)
        if self.scaler is not None:
            self.scaler.end_step()
        return loss
