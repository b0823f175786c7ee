"""YOLOv7-l on the MI355X engine -- INFERENCE path.

Mirrors ``core/models/yolov7_model.py:14-525`` (phi = 'l') of the reference as an engine graph:

* every ``ConvBNSiLU`` is one convolution launch with the running statistics folded into its epilogue;
* ``Multi_Concat_Block`` (ELAN): the tensors its final 1x1 reads are channel slices of ONE buffer in ``torch.cat`` order --
  cv1 / cv2 and the selected 3x3 stages write straight into them, the unselected stages own a buffer each;
* ``Transition_Block``: the stride-2 branch and the max-pool branch write the two halves of the consumer's concat buffer; in
  the PANet the third operand of that concat (P4 / P5) is produced into the same buffer by its own block;
* ``SPPCSPC``: the 5 / 9 / 13 max pools are the chained 5x5 pools of the YOLOv8 SPPF (identical values: max pooling with
  implicit -inf padding composes), written into the slices of the buffer cv5 reads;
* ``RepConv`` keeps its training form (no re-parametrisation of the caller's weights): 3x3 + BN into a buffer, then 1x1 + BN with
  that buffer added before the SiLU (c1 != c2 in YOLOv7-l: no identity branch);
* the three heads write fp32 rows (B, 400 + 1600 + 6400, 3 * (5 + nc) padded to 8) in the order of the reference's outputs
  (coarsest first); ``forward`` returns them as the reference's NCHW tensors, ``cvx_yolo7_decode`` reads the rows in place.

``state_dict``: the reference's 558 keys / shapes / order, bit-identical to ``Yolo7(cfg)`` (``cfg.train.pretrained = False``)
under the same global seed.  Training (Yolo7Loss with its SimOTA matching) is not built: ``model.train()`` forward raises.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional, List

import torch
import torch.nn as nn

from . import _lib as L
from .engine import Engine
from .graph import Graph, TensorSlot

BN_EPS, BN_MOMENTUM = 1e-3, 0.03                 # ConvBNSiLU / RepConv BatchNorms (yolov7_model.py:18,195-203)
TC, BC, PC, E, N = 32, 32, 32, 2, 4               # phi = 'l' (yolov7_model.py:366-371)
IDS_BACKBONE = (-1, -3, -5, -6)
IDS_NECK = (-1, -2, -3, -4, -5, -6)


def _mcb(prefix, c1, c2, c3, e, ids):
    c_ = int(c2 * e)
    convs = [(prefix + ".cv1", c_, c1, 1, 1), (prefix + ".cv2", c_, c1, 1, 1)]
    convs += [(prefix + f".cv3.{i}", c2, c_ if i == 0 else c2, 3, 1) for i in range(N)]
    convs.append((prefix + ".cv4", c3, c_ * 2 + c2 * (len(ids) - 2), 1, 1))
    return convs


def _trans(prefix, c1, c2):
    return [(prefix + ".cv1", c2, c1, 1, 1), (prefix + ".cv2", c2, c1, 1, 1), (prefix + ".cv3", c2, c2, 3, 2)]


def conv_bn_specs():
    """Every ConvBNSiLU, module registration order: (key, cout, cin, k, stride)."""
    t, b = TC, BC
    s = [("backbone.stem.0", t, 3, 3, 1), ("backbone.stem.1", 2 * t, t, 3, 2), ("backbone.stem.2", 2 * t, 2 * t, 3, 1),
         ("backbone.dark2.0", 4 * t, 2 * t, 3, 2)]
    s += _mcb("backbone.dark2.1", 4 * t, 2 * b, 8 * t, 1, IDS_BACKBONE)
    s += _trans("backbone.dark3.0", 8 * t, 4 * t) + _mcb("backbone.dark3.1", 8 * t, 4 * b, 16 * t, 1, IDS_BACKBONE)
    s += _trans("backbone.dark4.0", 16 * t, 8 * t) + _mcb("backbone.dark4.1", 16 * t, 8 * b, 32 * t, 1, IDS_BACKBONE)
    s += _trans("backbone.dark5.0", 32 * t, 16 * t) + _mcb("backbone.dark5.1", 32 * t, 8 * b, 32 * t, 1, IDS_BACKBONE)
    c_ = 16 * t
    s += [("sppcspc.cv1", c_, 32 * t, 1, 1), ("sppcspc.cv2", c_, 32 * t, 1, 1), ("sppcspc.cv3", c_, c_, 3, 1), ("sppcspc.cv4", c_, c_, 1, 1),
          ("sppcspc.cv5", c_, 4 * c_, 1, 1), ("sppcspc.cv6", c_, c_, 3, 1), ("sppcspc.cv7", 16 * t, 2 * c_, 1, 1)]
    s += [("conv_for_P5", 8 * t, 16 * t, 1, 1), ("conv_for_feat2", 8 * t, 32 * t, 1, 1)]
    s += _mcb("conv3_for_upsample1", 16 * t, 4 * PC, 8 * t, E, IDS_NECK)
    s += [("conv_for_P4", 4 * t, 8 * t, 1, 1), ("conv_for_feat1", 4 * t, 16 * t, 1, 1)]
    s += _mcb("conv3_for_upsample2", 8 * t, 2 * PC, 4 * t, E, IDS_NECK)
    s += _trans("down_sample1", 4 * t, 4 * t) + _mcb("conv3_for_downsample1", 16 * t, 4 * PC, 8 * t, E, IDS_NECK)
    s += _trans("down_sample2", 8 * t, 8 * t) + _mcb("conv3_for_downsample2", 32 * t, 8 * PC, 16 * t, E, IDS_NECK)
    return s


REP = (("rep_conv_1", 4 * TC, 8 * TC), ("rep_conv_2", 8 * TC, 16 * TC), ("rep_conv_3", 16 * TC, 32 * TC))
HEADS = (("yolo_head_P3", 8 * TC), ("yolo_head_P4", 16 * TC), ("yolo_head_P5", 32 * TC))


class Yolo7Layout:
    """Arena offsets for every tensor of the reference's Yolo7 ``state_dict`` (same keys, shapes, order)."""

    def __init__(self, nc: int = 20):
        self.nc = nc
        self.no = 3 * (5 + nc)
        self.no_pad = (self.no + 7) & ~7
        self.slots: "OrderedDict[str, TensorSlot]" = OrderedDict()
        self.nbt_keys: List[str] = []
        self.convs: Dict[str, dict] = {}
        self.init_order: List[tuple] = []                       # ("conv" | "bn", key) in modules() order
        self._p = self._s = 0
        for key, cout, cin, k, s in conv_bn_specs():
            self.conv_bn(key + ".conv", key + ".bn", cout, cin, k, stride=s)
        for key, c1, c2 in REP:
            self.conv_bn(key + ".rbr_dense.0", key + ".rbr_dense.1", c2, c1, 3)
            self.conv_bn(key + ".rbr_1x1.0", key + ".rbr_1x1.1", c2, c1, 1)
        for key, c in HEADS:
            self.conv(key, self.no, c, 1, bias=True)
        self.n_params = (self._p + 3) & ~3
        self.n_stats = (self._s + 3) & ~3

    def _take(self, arena, n):
        if arena == "param":
            off, self._p = self._p, (self._p + n + 3) & ~3
        else:
            off, self._s = self._s, (self._s + n + 3) & ~3
        return off

    def conv(self, key, cout, cin, k, bias=False, stride=1):
        ce = (cout + 7) & ~7
        spec = dict(cout=cout, cout_eng=ce, cin=cin, k=k, stride=stride, w_off=self._take("param", ce * k * k * cin))
        self.slots[key + ".weight"] = TensorSlot("param", spec["w_off"], (cout, cin, k, k), (k * k * cin, 1, k * cin, cin))
        if bias:
            spec["bias_off"] = self._take("param", ce)
            self.slots[key + ".bias"] = TensorSlot("param", spec["bias_off"], (cout,), (1,))
        self.convs[key] = spec
        self.init_order.append(("conv", key))
        return spec

    def conv_bn(self, ckey, bkey, cout, cin, k, stride=1):
        spec = self.conv(ckey, cout, cin, k, stride=stride)
        spec.update(gamma_off=self._take("param", cout), beta_off=self._take("param", cout), rmean_off=self._take("stat", cout),
                    rvar_off=self._take("stat", cout))
        self.slots[bkey + ".weight"] = TensorSlot("param", spec["gamma_off"], (cout,), (1,))
        self.slots[bkey + ".bias"] = TensorSlot("param", spec["beta_off"], (cout,), (1,))
        self.slots[bkey + ".running_mean"] = TensorSlot("stat", spec["rmean_off"], (cout,), (1,), False)
        self.slots[bkey + ".running_var"] = TensorSlot("stat", spec["rvar_off"], (cout,), (1,), False)
        self.slots[bkey + ".num_batches_tracked"] = TensorSlot("nbt", len(self.nbt_keys), (), (), False)
        self.nbt_keys.append(bkey + ".num_batches_tracked")
        self.init_order.append(("bn", bkey))


def build_yolov7_graph(lay: Yolo7Layout, H: int, W: int) -> Graph:
    """Buffer plan + op list for an (H, W) input (multiples of 32)."""
    if H % 32 or W % 32:
        raise ValueError("input height/width must be multiples of 32")
    g = Graph()
    t = TC

    def buf(h, w, ch, kind=L.BUF_ACT_F16):
        g.bufs.append((h, w, ch, kind))
        return len(g.bufs) - 1

    def V(b, off, ch, pix=0):
        return (b, off, ch, pix)

    def conv(ckey, vin, vout, hin, win, act=L.ACT_BN_SILU, res=None):
        s = lay.convs[ckey]
        k, st = s["k"], s["stride"]
        ho, wo = (hin + 2 * (k // 2) - k) // st + 1, (win + 2 * (k // 2) - k) // st + 1
        op = dict(type=L.OP_CONV, name=ckey, out=vout, ih=hin, iw=win, oh=ho, ow=wo, k=k, stride=st, pad=k // 2, dil=1, act=act, needs_dgrad=0 if ckey == "backbone.stem.0.conv" else 1,
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

    def cbs(key, vin, h, w, vout=None):
        s = lay.convs[key + ".conv"]
        ho, wo = ((h + 2 * (s["k"] // 2) - s["k"]) // s["stride"] + 1, (w + 2 * (s["k"] // 2) - s["k"]) // s["stride"] + 1)
        if vout is None:
            vout = V(buf(ho, wo, s["cout"]), 0, s["cout"])
        conv(key + ".conv", vin, vout, h, w)
        return vout

    def mcb(p, vin, h, w, ids, vout=None):
        """Multi_Concat_Block.forward (yolov7_model.py:41-52): x_all = [cv1, cv2, cv3[0], ..., cv3[3]]; cat(x_all[ids]) -> cv4."""
        c_, c2 = lay.convs[p + ".cv1.conv"]["cout"], lay.convs[p + ".cv3.0.conv"]["cout"]
        widths = [c_, c_] + [c2] * N
        order = [i % (N + 2) for i in ids]                       # x_all index of every concat slot, in cat order
        cat = buf(h, w, sum(widths[i] for i in order))
        slot, off = {}, 0
        for i in order:
            slot[i] = V(cat, off, widths[i])
            off += widths[i]
        outs = {}
        for i in range(N + 2):
            outs[i] = slot[i] if i in slot else V(buf(h, w, widths[i]), 0, widths[i])
        cbs(p + ".cv1", vin, h, w, outs[0])
        cbs(p + ".cv2", vin, h, w, outs[1])
        for i in range(N):
            cbs(p + f".cv3.{i}", outs[1 + i], h, w, outs[2 + i])
        return cbs(p + ".cv4", V(cat, 0, off), h, w, vout)

    def trans(p, vin, h, w, dst, off):
        """Transition_Block.forward (yolov7_model.py:74-86): cat([cv3(cv2(x)), cv1(mp(x))]) into dst[off : off + 2 c2]."""
        c1, c2 = vin[2], lay.convs[p + ".cv1.conv"]["cout"]
        pooled = V(buf(h // 2, w // 2, c1), 0, c1)
        simple(L.OP_MAXPOOL2, p + ".mp", vin, pooled, h, w, h // 2, w // 2)
        cbs(p + ".cv1", pooled, h // 2, w // 2, V(dst, off + c2, c2))
        mid = cbs(p + ".cv2", vin, h, w)
        cbs(p + ".cv3", mid, h, w, V(dst, off, c2))

    img = buf(H, W, 8)
    g.image_buf = img
    y = cbs("backbone.stem.0", V(img, 0, 8), H, W)
    y = cbs("backbone.stem.1", y, H, W)
    h, w = H // 2, W // 2
    y = cbs("backbone.stem.2", y, h, w)
    y = cbs("backbone.dark2.0", y, h, w)
    h, w = h // 2, w // 2
    y = mcb("backbone.dark2.1", y, h, w, IDS_BACKBONE)
    feats = []
    for name, cin in (("dark3", 8 * t), ("dark4", 16 * t), ("dark5", 32 * t)):
        c2 = lay.convs[f"backbone.{name}.0.cv1.conv"]["cout"]
        cat = buf(h // 2, w // 2, 2 * c2)
        trans(f"backbone.{name}.0", y, h, w, cat, 0)
        h, w = h // 2, w // 2
        y = mcb(f"backbone.{name}.1", V(cat, 0, 2 * c2), h, w, IDS_BACKBONE)
        feats.append((y, h, w))
    (feat1, h1, w1), (feat2, h2, w2), (feat3, h3, w3) = feats

    # SPPCSPC (yolov7_model.py:160-164)
    c_ = 16 * t
    x1 = cbs("sppcspc.cv1", feat3, h3, w3)
    x1 = cbs("sppcspc.cv3", x1, h3, w3)
    spp = buf(h3, w3, 4 * c_)
    cbs("sppcspc.cv4", x1, h3, w3, V(spp, 0, c_))
    for j in range(3):                                           # 5, 9 = 5 o 5, 13 = 5 o 5 o 5
        simple(L.OP_MAXPOOL5, f"sppcspc.m.{j}", V(spp, j * c_, c_), V(spp, (j + 1) * c_, c_), h3, w3, h3, w3)
    y1 = cbs("sppcspc.cv5", V(spp, 0, 4 * c_), h3, w3)
    cat7 = buf(h3, w3, 2 * c_)
    cbs("sppcspc.cv6", y1, h3, w3, V(cat7, 0, c_))
    cbs("sppcspc.cv2", feat3, h3, w3, V(cat7, c_, c_))
    # P5 also feeds the last concat [down_sample2(P4) | P5] (yolov7_model.py:500-503): produce it there
    cat_d2 = buf(h3, w3, 32 * t)
    P5 = cbs("sppcspc.cv7", V(cat7, 0, 2 * c_), h3, w3, V(cat_d2, 16 * t, 16 * t))

    # top-down
    cat_u1 = buf(h2, w2, 16 * t)                                 # [conv_for_feat2(feat2) | up(conv_for_P5(P5))]
    cbs("conv_for_feat2", feat2, h2, w2, V(cat_u1, 0, 8 * t))
    p5c = cbs("conv_for_P5", P5, h3, w3)
    simple(L.OP_UPSAMPLE2, "upsample.p5", p5c, V(cat_u1, 8 * t, 8 * t), h3, w3, h2, w2)
    cat_d1 = buf(h2, w2, 16 * t)                                 # [down_sample1(P3) (2 x 4t) | P4 (8t)]
    P4 = mcb("conv3_for_upsample1", V(cat_u1, 0, 16 * t), h2, w2, IDS_NECK, V(cat_d1, 8 * t, 8 * t))
    cat_u2 = buf(h1, w1, 8 * t)
    cbs("conv_for_feat1", feat1, h1, w1, V(cat_u2, 0, 4 * t))
    p4c = cbs("conv_for_P4", P4, h2, w2)
    simple(L.OP_UPSAMPLE2, "upsample.p4", p4c, V(cat_u2, 4 * t, 4 * t), h2, w2, h1, w1)
    P3 = mcb("conv3_for_upsample2", V(cat_u2, 0, 8 * t), h1, w1, IDS_NECK)
    # bottom-up
    trans("down_sample1", P3, h1, w1, cat_d1, 0)
    P4 = mcb("conv3_for_downsample1", V(cat_d1, 0, 16 * t), h2, w2, IDS_NECK)
    trans("down_sample2", P4, h2, w2, cat_d2, 0)
    P5 = mcb("conv3_for_downsample2", V(cat_d2, 0, 32 * t), h3, w3, IDS_NECK)

    # RepConv (training form) + heads; rows in the order of the reference's outputs: out0 (P5), out1 (P4), out2 (P3)
    g.level_hw = [(h3, w3), (h2, w2), (h1, w1)]
    g.anchors = sum(a * b for a, b in g.level_hw)
    pred = buf(g.anchors, 1, lay.no_pad, L.BUF_PRED_F32)
    g.pred_buf = pred
    a_off = 0
    for rep, head, (feat, hh, ww) in (("rep_conv_3", "yolo_head_P5", (P5, h3, w3)), ("rep_conv_2", "yolo_head_P4", (P4, h2, w2)),
                                       ("rep_conv_1", "yolo_head_P3", (P3, h1, w1))):
        c2 = lay.convs[rep + ".rbr_dense.0"]["cout"]
        dense, fused = V(buf(hh, ww, c2), 0, c2), V(buf(hh, ww, c2), 0, c2)
        conv(rep + ".rbr_dense.0", feat, dense, hh, ww, act=L.ACT_BN_LINEAR)
        conv(rep + ".rbr_1x1.0", feat, fused, hh, ww, act=L.ACT_BN_SILU, res=dense)
        conv(head, fused, V(pred, 0, lay.no_pad, a_off), hh, ww, act=L.ACT_BIAS)
        a_off += hh * ww
    return g


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise L.CvxError("parameter holder: the engine executes the whole graph (call the YOLOv7 model)")


class Yolo7L(nn.Module):
    """``Yolo7(cfg)`` of the reference (phi 'l', yolov7_model.py:355-525) on the engine: ``model(x)`` returns (out0, out1, out2),
    each (B, 3 * (5 + nc), H_l, W_l) fp32, coarsest level first.  In training mode (grad enabled) the outputs are connected to the
    engine's backward pass: any torch loss on them (the reference's Yolo7Loss is plain torch code on these tensors,
    core/loss/yolo7_loss.py) back-propagates into batch-statistics BatchNorm, the RepConv sum inside the SiLU, 2x2 / 5x5 max pools,
    upsampling and every convolution's data / weight gradient; parameter gradients land in ``p.grad`` (views of one flat arena)."""

    def __init__(self, num_classes: int = 20, loss_scale: float = 1024.0):
        super().__init__()
        self.layout = lay = Yolo7Layout(num_classes)
        self.num_classes = num_classes
        self.loss_scale = float(loss_scale)
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
            t_ = fn(self._flat[k])
            if k != "nbt" and t_.dtype != torch.float32:
                raise L.CvxError("the engine keeps fp32 master parameters; half()/bfloat16() are not supported (compute is fp16 inside)")
            self._flat[k] = t_.long().contiguous() if k == "nbt" else t_.contiguous()
        self._attach_views()
        self._engines.clear()
        return self

    def _init_like_reference(self):
        """The reference's draws from the global RNG: every nn.Conv2d's default init at construction (weight, then bias), module
        registration order; then ``init_weights`` (yolov7_model.py:449-458) over ``modules()``: conv weight N(0, 0.02), conv
        bias 0, BatchNorm weight N(1, 0.02), bias 0."""
        lay = self.layout
        sd = dict(self.named_parameters())
        sd.update(dict(self.named_buffers()))
        with torch.no_grad():
            for kind, key in lay.init_order:
                if kind != "conv":
                    continue
                sl = lay.slots[key + ".weight"]
                w = torch.empty(sl.shape)
                nn.init.kaiming_uniform_(w, a=math.sqrt(5))
                if "bias_off" in lay.convs[key]:
                    bb = torch.empty(lay.slots[key + ".bias"].shape)
                    bound = 1.0 / math.sqrt(sl.shape[1] * sl.shape[2] * sl.shape[3])
                    nn.init.uniform_(bb, -bound, bound)
            for kind, key in lay.init_order:
                w = torch.empty(lay.slots[key + ".weight"].shape)
                if kind == "conv":
                    nn.init.normal_(w, 0, 0.02)
                    sd[key + ".weight"].copy_(w)
                    if "bias_off" in lay.convs[key]:
                        sd[key + ".bias"].zero_()
                else:
                    nn.init.normal_(w, 1, 0.02)
                    sd[key + ".weight"].copy_(w)
                    sd[key + ".bias"].zero_()
                    sd[key + ".running_mean"].zero_()
                    sd[key + ".running_var"].fill_(1.0)
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
        key = (h, w, dev)
        eng = self._engines.get(key)
        if eng is None:
            if dev.type != "cuda":
                raise L.CvxError("Yolo7L runs on an MI355X only: move the model with .to('cuda') first (there is no CPU fallback)")
            eng = Engine(build_yolov7_graph(self.layout, h, w), dev)
            eng.set_bn(BN_EPS, BN_MOMENTUM)
            self._engines[key] = eng
        eng.bind(self._flat["param"], self.flat_grads if self.training else self._flat["grad"], self._flat["stat"])
        return eng

    def _run_forward(self, x: torch.Tensor, training: bool) -> torch.Tensor:
        """(B,3,H,W) -> the engine's fp32 head rows (B, sum_l H_l*W_l, no_pad): what ``cvx_yolo7_decode`` reads."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected images of shape (B, 3, H, W)")
        eng = self.engine_for(int(x.shape[2]), int(x.shape[3]))
        self._last_engine = eng
        rows = eng.forward(x, training)
        if training:
            self._flat["nbt"] += 1
        return rows

    def forward_rows(self, x: torch.Tensor) -> torch.Tensor:
        return self._run_forward(x, self.training)

    def _rows_to_levels(self, rows: torch.Tensor):
        B = int(rows.shape[0])
        lay, lib = self.layout, L.load()
        outs, a_off = [], 0
        A = self._last_engine.graph.anchors
        for (hh, ww) in self._last_engine.graph.level_hw:
            o = torch.empty(B, lay.no_pad, hh, ww, dtype=torch.float32, device=rows.device)
            L.check(lib.cvx_pred_level_to_nchw(L.ptr(rows), B, A, lay.no_pad, a_off, hh, ww, L.ptr(o), L.stream_ptr(rows.device)), "cvx_pred_level_to_nchw")
            outs.append(o)
            a_off += hh * ww
        return outs

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

    def _backward_levels(self, grads):
        """Gradients w.r.t. the three (B, no_pad, h, w) level tensors -> loss_scale * dLoss/drows in fp16 -> engine backward."""
        eng, lay, lib = self._last_engine, self.layout, L.load()
        ref = next(g for g in grads if g is not None)
        B, A = int(ref.shape[0]), eng.graph.anchors
        dpred = torch.zeros(B, A, lay.no_pad, dtype=torch.float16, device=ref.device)
        a_off = 0
        for g, (hh, ww) in zip(grads, eng.graph.level_hw):
            if g is not None:
                L.check(lib.cvx_nchw_grad_to_dpred(L.ptr(g.contiguous().float()), B, A, lay.no_pad, a_off, hh, ww, self.loss_scale, L.ptr(dpred),
                                                   L.stream_ptr(ref.device)), "cvx_nchw_grad_to_dpred")
            a_off += hh * ww
        first = next(p for p in self.parameters() if p.requires_grad)
        if first.grad is None:               # optimizer.zero_grad(set_to_none=True) happened (or first step)
            self.flat_grads.zero_()
            self._grads_attached = False
        self.last_dpred = dpred
        eng.backward(dpred, self.loss_scale)
        if not self._grads_attached or first.grad is None:
            self.attach_grads()

    def forward(self, x: torch.Tensor):
        if self.training and torch.is_grad_enabled():
            outs = _Y7Fn.apply(x, self._anchor, self)
        else:
            self.last_rows = self._run_forward(x, self.training)
            outs = self._rows_to_levels(self.last_rows)
        res = tuple(o[:, :self.layout.no] for o in outs)
        res[0].model = self                    # the fused Yolo7Loss starts from the head rows (last_rows)
        return res


class _Y7Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, anchor, model):
        ctx.model = model
        ctx.set_materialize_grads(False)
        model.last_rows = model._run_forward(images, training=True)
        return tuple(model._rows_to_levels(model.last_rows))

    @staticmethod
    def backward(ctx, *grads):
        if any(g is not None for g in grads):
            ctx.model._backward_levels(grads)
        return None, None, None


ANCHORS_PX = (12, 16, 19, 36, 40, 28, 36, 75, 76, 55, 72, 146, 142, 110, 192, 243, 459, 401)    # configs/yolo7_cfg.py
ANCHORS_MASK = ((6, 7, 8), (3, 4, 5), (0, 1, 2))
LOSS_STRIDES = (32.0, 16.0, 8.0)


class _Y7LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out0, owner, model, targets, img_size):
        items, dpred = owner.op(model.last_rows.detach(), model._last_engine.graph.level_hw, targets, img_size, model.loss_scale)
        ctx.model, ctx.dpred = model, dpred
        model.last_dpred = dpred
        owner.last_items = items
        return items[0].reshape(())

    @staticmethod
    def backward(ctx, gout):
        m = ctx.model
        first = next(p for p in m.parameters() if p.requires_grad)
        if first.grad is None:
            m.flat_grads.zero_()
            m._grads_attached = False
        m._last_engine.backward(ctx.dpred, m.loss_scale / float(gout))
        if not m._grads_attached or first.grad is None:
            m.attach_grads()
        return None, None, None, None, None


class Yolo7Loss:
    """``Yolo7Loss(anchors, num_classes, input_shape, anchors_mask, label_smoothing)`` of the reference (core/loss/yolo7_loss.py:14-444) on
    the engine: ``loss, box, obj, cls = criterion(predictions, targets, imgs)`` with ``predictions = model(imgs)`` and ``targets`` (N, 6)
    [image, class, cx, cy, w, h] as ``yolo7_collate`` builds them.  Candidate generation, the SimOTA assignment and the three loss terms
    with their gradient run in ``cvx_yolo7_loss`` on the head rows behind ``predictions``; ``loss.backward()`` runs the engine's backward."""

    def __init__(self, anchors=None, num_classes: int = 20, input_shape=(640, 640), anchors_mask=ANCHORS_MASK, label_smoothing: float = 0.0):
        import numpy as np
        a = np.asarray(ANCHORS_PX if anchors is None else anchors, dtype=np.float32).reshape(-1, 2)
        self.anchors_px = torch.from_numpy(np.concatenate([a[list(m)] for m in anchors_mask]).astype(np.float32).copy())   # level-major
        self.nc = int(num_classes)
        self.box_ratio = 0.05
        self.obj_ratio = 1.0 * (input_shape[0] * input_shape[1]) / (640 ** 2)
        self.cls_ratio = 0.5 * (num_classes / 80)
        self.label_smoothing = float(label_smoothing)
        self._ws = self._bad = None
        self.last_items = None

    def overflowed(self) -> int:
        """bit 0: an image had more than 64 ground truths, bit 1: more than 2880 candidates (the surplus was dropped); synchronises."""
        return 0 if self._bad is None else int(self._bad.item())

    def op(self, rows: torch.Tensor, level_hw, targets: torch.Tensor, img_size: float, loss_scale: float, dpred: Optional[torch.Tensor] = None):
        """rows (B, A, ld) fp32 -> (items (4,): total, box, obj, cls (ratios applied); dpred (B, A, ld) fp16)"""
        if rows.device.type != "cuda":
            raise L.CvxError("Yolo7Loss runs on an MI355X only (there is no CPU path)")
        lib = L.load()
        B, A, ld = rows.shape
        dev = rows.device
        targets = targets.to(dev).float().contiguous()
        N = int(targets.shape[0])
        need = int(lib.cvx_yolo7_loss_workspace_bytes(B, A, ld, N))
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
            self._bad = torch.zeros(1, dtype=torch.int32, device=dev)
        if dpred is None:
            dpred = torch.empty(B, A, ld, dtype=torch.float16, device=dev)
        items = torch.empty(4, device=dev)
        import ctypes as C
        hw = (C.c_int32 * 6)(*[int(v) for pair in level_hw for v in pair])
        anc = (C.c_float * 18)(*[float(v) for v in self.anchors_px.flatten().tolist()])
        strides = (C.c_float * 3)(*LOSS_STRIDES)
        L.check(lib.cvx_yolo7_loss(L.ptr(rows), ld, B, self.nc, hw, anc, strides, L.ptr(targets) if N else None, N, float(img_size), self.box_ratio,
                                   self.obj_ratio, self.cls_ratio, self.label_smoothing, float(loss_scale), L.ptr(items), L.ptr(dpred),
                                   L.ptr(self._bad), L.ptr(self._ws), L.stream_ptr(dev)), "cvx_yolo7_loss")
        return items, dpred

    def __call__(self, predictions, targets, imgs):
        model = getattr(predictions[0], "model", None)
        if model is None:
            raise L.CvxError("Yolo7Loss needs the output of Yolo7L.forward (it carries the head rows the loss starts from)")
        img_size = float(imgs.shape[2])                           # imgs[b].shape[1]
        if model.training and torch.is_grad_enabled():
            total = _Y7LossFn.apply(predictions[0], self, model, targets, img_size)
            it = self.last_items
            return total, it[1], it[2], it[3]
        it = self.op(model.last_rows, model._last_engine.graph.level_hw, targets, img_size, model.loss_scale)[0]
        return it[0], it[1], it[2], it[3]


class Yolo7TrainStep:
    """One optimisation step of the reference's ``Yolo7Trainer.train_loop`` (core/trainer/yolo7_train.py:79-97) as C-ABI calls: engine
    forward (training), ``cvx_yolo7_loss``, engine backward, [gradient sum over the ranks], fused Adam with GradScaler's inf/nan check."""

    def __init__(self, model: Yolo7L, criterion: Yolo7Loss, optimizer, scaler=None, process_group=None, n_buckets: int = 4):
        self.model, self.criterion, self.optimizer, self.scaler = model, criterion, optimizer, scaler
        self.pg, self.n_buckets = process_group, n_buckets
        self.world, self.distributed = 1, False
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
            self.distributed = True
        self._dpred = self._side = None

    def __call__(self, images: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        from .engine import check_finite
        m, crit = self.model, self.criterion
        if not m.training:
            raise L.CvxError("Yolo7TrainStep: call model.train() first")
        dev = m.flat_params.device
        self.optimizer.sync_lr()
        rows = m._run_forward(images, True)
        m.last_rows = rows
        eng = m._last_engine
        if self._dpred is None or self._dpred.shape != rows.shape:
            self._dpred = torch.empty(rows.shape, device=dev, dtype=torch.float16)
        scale = self.scaler.begin_step() if self.scaler is not None else m.loss_scale
        items, dpred = crit.op(rows, eng.graph.level_hw, targets, float(images.shape[2]), scale, self._dpred)
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
        return items
