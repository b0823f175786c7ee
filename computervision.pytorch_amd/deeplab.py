"""DeepLabv3+ (ResNet-101, output stride 16) on the MI355X engine -- INFERENCE path.

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
global seed.  Training (FocalLoss, backward) is not built this round: ``model.train()`` forward raises.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List

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


def build_deeplab_graph(lay: DeepLabLayout, H: int, W: int) -> Graph:
    """Buffer plan + op list for an (H, W) input (any size from 33 up: the reference runs 513 x 513)."""
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
        op = dict(type=L.OP_CONV, name=ckey, out=vout, ih=hin, iw=win, oh=ho, ow=wo, k=k, stride=stride, pad=pad, dil=dil, act=act, needs_dgrad=0,
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
    aspp = V(buf(h, w, ASPP_OUT), 0, ASPP_OUT)
    conv(c + "aspp.project.0", V(cat, 0, 5 * ASPP_OUT), aspp, h, w)
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
    on the engine: ``model.eval(); model(x)`` returns the (B, num_classes, H, W) fp32 logits."""

    def __init__(self, num_classes: int = 21):
        super().__init__()
        self.layout = lay = DeepLabLayout(num_classes)
        self.num_classes = num_classes
        self._flat = {"param": torch.zeros(lay.n_params), "stat": torch.zeros(lay.n_stats), "nbt": torch.zeros(len(lay.nbt_keys), dtype=torch.long)}
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
                mod._parameters[parts[-1]] = nn.Parameter(view, requires_grad=False)
            else:
                mod._buffers[parts[-1]] = view

    def _apply(self, fn, recurse=True):
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
    def engine_for(self, h: int, w: int) -> Engine:
        dev = self._flat["param"].device
        key = (h, w, dev)
        eng = self._engines.get(key)
        if eng is None:
            if dev.type != "cuda":
                raise L.CvxError("DeepLabV3PlusR101 runs on an MI355X only: move the model with .to('cuda') first (there is no CPU fallback)")
            eng = Engine(build_deeplab_graph(self.layout, h, w), dev)
            eng.set_bn(BN_EPS, BN_MOMENTUM)
            self._engines[key] = eng
        eng.bind(self._flat["param"], None, self._flat["stat"])
        return eng

    def forward_rows(self, x: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) -> the engine's fp32 logits rows (B, h*w, nc_pad) at the decoder's resolution (stride 4)."""
        if self.training:
            raise L.CvxError("DeepLabv3+ on the MI355X engine is inference-only this round: call model.eval() first")
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected images of shape (B, 3, H, W)")
        eng = self.engine_for(int(x.shape[2]), int(x.shape[3]))
        self._last_engine = eng
        return eng.forward(x, False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        rows = self.forward_rows(x)
        self.last_rows = rows
        B, _, H, W = x.shape
        lh, lw = self._last_engine.graph.level_hw[0]
        out = torch.empty(B, self.num_classes, H, W, dtype=torch.float32, device=x.device)
        lib = L.load()
        L.check(lib.cvx_resize_bilinear_rows_to_nchw(L.ptr(rows), self.layout.nc_pad, B, self.num_classes, lh, lw, H, W, L.ptr(out),
                                                      L.stream_ptr(x.device)), "cvx_resize_bilinear_rows_to_nchw")
        return out
