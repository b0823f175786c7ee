"""SSD300 (VGG16-BN) on the MI355X engine -- INFERENCE path.

Mirrors ``core/models/ssd_model.py:6-191`` of the reference as an engine graph:

* VGG16 with BatchNorm: ``Conv2d(bias=True) + BatchNorm2d + ReLU`` is one convolution launch -- the running statistics AND the
  convolution's own bias folded into the epilogue (``CVX_OPF_CONV_BIAS``); the 2x2 pools (the third with ``ceil_mode``), the
  3x3 / stride-1 pool5, the dilated conv6 (rate 6, a tap table of the generic kernel) and conv7 (bias + ReLU epilogues);
* ``L2Normalize`` on conv4_3 (one wave per pixel); ``ExtraLayer``: eight convolutions with bias and, as in the reference's
  ``forward`` (ssd_model.py:90-110), NO activation between them;
* the six (loc, conf) 3x3 heads write fp32 rows (B, 1940, [loc 24 | conf 128]); ``forward`` returns the reference's tensors
  (B, 8732, 4) and (B, 8732, 21) -- flattened in NCHW order per level, as the reference does (no permute, :177-183).

``state_dict``: the reference's 136 keys / shapes / order, bit-identical to ``SSD(cfg)`` under the same global seed (torch's
default initialisation in the reference's construction order: loc_i / conf_i alternately).  Training (MultiBoxLossV2) is not built.
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

VGG_PARAMS = (64, 64, "M", 128, 128, "M", 256, 256, 256, "C", 512, 512, 512, "M", 512, 512, 512)
BN_EPS, BN_MOMENTUM = 1e-5, 0.1
ASPECT_RATIOS = ([1, 2, 0.5], [1, 2, 0.5, 3, 1.0 / 3], [1, 2, 0.5, 3, 1.0 / 3], [1, 2, 0.5, 3, 1.0 / 3], [1, 2, 0.5], [1, 2, 0.5])
FEATURE_CHANNELS = (512, 1024, 512, 256, 256, 256)
BOXES_PER_PIXEL = tuple(len(a) + 1 for a in ASPECT_RATIOS)
EXTRAS = (("conv1", 256, 1024, 1, 1, 0), ("conv2", 512, 256, 3, 2, 1), ("conv3", 128, 512, 1, 1, 0), ("conv4", 256, 128, 3, 2, 1),
          ("conv5", 128, 256, 1, 1, 0), ("conv6", 256, 128, 3, 1, 0), ("conv7", 128, 256, 1, 1, 0), ("conv8", 256, 128, 3, 1, 0))
LOC_COLS = 24                                    # widest loc head (6 boxes x 4); the conf heads start at this column


def vgg_plan():
    """backbone.layers as data (ssd_model.py:9-38): (index, 'conv', cout, cin, k, pad, dil, bn_index | None) or (index, 'M' | 'C' | 'P5')."""
    plan, idx, cin = [], 0, 3
    for v in VGG_PARAMS:
        if v in ("M", "C"):
            plan.append((idx, v))
            idx += 1
        else:
            plan.append((idx, "conv", v, cin, 3, 1, 1, idx + 1))
            idx += 3
            cin = v
    plan += [(idx, "P5"), (idx + 1, "conv", 1024, cin, 3, 6, 6, None), (idx + 3, "conv", 1024, 1024, 1, 0, 1, None)]
    return plan


class SsdLayout:
    """Arena offsets for every tensor of the reference's SSD ``state_dict`` (same keys, shapes, order)."""

    def __init__(self, nc: int = 20):
        self.nc = nc
        self.conf_cols = max((n * (nc + 1) + 7) & ~7 for n in BOXES_PER_PIXEL)
        self.pred_ld = LOC_COLS + self.conf_cols
        self.slots: "OrderedDict[str, TensorSlot]" = OrderedDict()
        self.nbt_keys: List[str] = []
        self.convs: Dict[str, dict] = {}
        self.construction: List[str] = []
        self._p = self._s = 0
        for item in vgg_plan():
            if item[1] != "conv":
                continue
            idx, _, cout, cin, k, pad, dil, bn = item
            spec = self.conv(f"backbone.layers.{idx}", cout, cin, k, pad=pad, dil=dil)
            if bn is not None:
                self.bn(f"backbone.layers.{bn}", cout, spec)
        self.l2_off = self._take("param", 512)
        self.slots["l2_norm.weight"] = TensorSlot("param", self.l2_off, (512,), (1,))
        for name, cout, cin, k, s, p in EXTRAS:
            self.conv(f"extras.{name}", cout, cin, k, stride=s, pad=p)
        heads = [(f"locs.{i}", n * 4, c) for i, (c, n) in enumerate(zip(FEATURE_CHANNELS, BOXES_PER_PIXEL))]
        heads += [(f"confs.{i}", n * (nc + 1), c) for i, (c, n) in enumerate(zip(FEATURE_CHANNELS, BOXES_PER_PIXEL))]
        for key, cout, cin in heads:                              # state_dict order: every loc head, then every conf head
            self.conv(key, cout, cin, 3, pad=1, construct=False)
        for i in range(6):                                        # construction order: loc_i, conf_i alternately (:131-162)
            self.construction += [f"locs.{i}", f"confs.{i}"]
        self.n_params = (self._p + 3) & ~3
        self.n_stats = (self._s + 3) & ~3

    def _take(self, arena, n):
        if arena == "param":
            off, self._p = self._p, (self._p + n + 3) & ~3
        else:
            off, self._s = self._s, (self._s + n + 3) & ~3
        return off

    def conv(self, key, cout, cin, k, stride=1, pad=0, dil=1, construct=True):
        ce = (cout + 7) & ~7
        spec = dict(cout=cout, cout_eng=ce, cin=cin, k=k, stride=stride, pad=pad, dil=dil, w_off=self._take("param", ce * k * k * cin),
                    bias_off=self._take("param", ce))
        self.slots[key + ".weight"] = TensorSlot("param", spec["w_off"], (cout, cin, k, k), (k * k * cin, 1, k * cin, cin))
        self.slots[key + ".bias"] = TensorSlot("param", spec["bias_off"], (cout,), (1,))
        self.convs[key] = spec
        if construct:
            self.construction.append(key)
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


def conv_out(n, k, stride, pad, dil=1):
    return (n + 2 * pad - dil * (k - 1) - 1) // stride + 1


def build_ssd_graph(lay: SsdLayout, H: int, W: int) -> Graph:
    if (H, W) != (300, 300):
        raise ValueError("the SSD graph is built for 300 x 300 inputs (the reference's ExtraLayer type '300')")
    g = Graph()

    def buf(h, w, ch, kind=L.BUF_ACT_F16):
        g.bufs.append((h, w, ch, kind))
        return len(g.bufs) - 1

    def V(b, off, ch, pix=0):
        return (b, off, ch, pix)

    def conv(key, vin, vout, hin, win, act, flags=0):
        s = lay.convs[key]
        ho, wo = conv_out(hin, s["k"], s["stride"], s["pad"], s["dil"]), conv_out(win, s["k"], s["stride"], s["pad"], s["dil"])
        op = dict(type=L.OP_CONV, name=key, out=vout, ih=hin, iw=win, oh=ho, ow=wo, k=s["k"], stride=s["stride"], pad=s["pad"], dil=s["dil"], act=act,
                  needs_dgrad=0 if vin[0] == img else 1, w_cin=s["cin"], w_off=s["w_off"], gamma_off=s.get("gamma_off", 0), beta_off=s.get("beta_off", 0),
                  bias_off=s["bias_off"], rmean_off=s.get("rmean_off", 0), rvar_off=s.get("rvar_off", 0), flags=flags)
        op["in"] = vin
        g.ops.append(op)
        return ho, wo

    def simple(kind, name, vin, vout, ih, iw, oh, ow, **kw):
        op = dict(type=kind, name=name, out=vout, ih=ih, iw=iw, oh=oh, ow=ow, **kw)
        op["in"] = vin
        g.ops.append(op)

    img = buf(H, W, 8)
    g.image_buf = img
    cur, h, w, c = V(img, 0, 8), H, W, 8
    sources = []
    for item in vgg_plan():
        kind = item[1]
        if kind in ("M", "C"):
            oh, ow = ((h + 1) // 2, (w + 1) // 2) if kind == "C" else (h // 2, w // 2)
            nxt = V(buf(oh, ow, c), 0, c)
            simple(L.OP_MAXPOOL2, f"backbone.layers.{item[0]}", cur, nxt, h, w, oh, ow)
            cur, h, w = nxt, oh, ow
        elif kind == "P5":
            nxt = V(buf(h, w, c), 0, c)
            simple(L.OP_MAXPOOL3S1, f"backbone.layers.{item[0]}", cur, nxt, h, w, h, w)
            cur = nxt
        else:
            idx, _, cout, cin, k, pad, dil, bn = item
            nxt = V(buf(h, w, cout), 0, cout)
            if bn is not None:
                conv(f"backbone.layers.{idx}", cur, nxt, h, w, L.ACT_BN_RELU, flags=L.OPF_CONV_BIAS)
            else:
                conv(f"backbone.layers.{idx}", cur, nxt, h, w, L.ACT_BIAS_RELU)
            cur, c = nxt, cout
            if bn == 31:                                          # conv4_3 + BN + ReLU = layers[32]: the first source (ssd_model.py:52)
                normed = V(buf(h, w, c), 0, c)
                simple(L.OP_L2NORM, "l2_norm", cur, normed, h, w, h, w, gamma_off=lay.l2_off)
                sources.append((normed, h, w, c))
    sources.append((cur, h, w, c))
    e, eh, ew = cur, h, w
    for j, (name, cout, cin, k, s, p) in enumerate(EXTRAS):       # no activation between the extra convolutions (ssd_model.py:90-110)
        oh, ow = conv_out(eh, k, s, p), conv_out(ew, k, s, p)
        nxt = V(buf(oh, ow, cout), 0, cout)
        conv(f"extras.{name}", e, nxt, eh, ew, L.ACT_BIAS_LINEAR)
        e, eh, ew = nxt, oh, ow
        if j % 2 == 1:
            sources.append((e, eh, ew, cout))
    g.level_hw = [(hh, ww) for _, hh, ww, _ in sources]
    g.anchors = sum(a * b for a, b in g.level_hw)
    pred = buf(g.anchors, 1, lay.pred_ld, L.BUF_PRED_F32)
    g.pred_buf = pred
    a_off = 0
    for i, (src, hh, ww, _) in enumerate(sources):
        conv(f"locs.{i}", src, V(pred, 0, lay.convs[f"locs.{i}"]["cout_eng"], a_off), hh, ww, L.ACT_BIAS)
        conv(f"confs.{i}", src, V(pred, LOC_COLS, lay.convs[f"confs.{i}"]["cout_eng"], a_off), hh, ww, L.ACT_BIAS)
        a_off += hh * ww
    return g


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise L.CvxError("parameter holder: the engine executes the whole graph (call the SSD model)")


class SSD300VGG(nn.Module):
    """``SSD(cfg)`` of the reference (ssd_model.py:131-191) on the engine: ``model(x)`` returns (loc (B, 8732, 4), conf (B, 8732, nc + 1))
    fp32.  In training mode (grad enabled) the two tensors are connected to the engine's backward pass (VGG16-BN with batch
    statistics behind biased convolutions, ceil-mode and 3x3 stride-1 max pools, L2Normalize and its weight, the bias-only extra
    layers and heads): any torch loss on them -- the reference's MultiBoxLossV2 is torch code on exactly these tensors -- trains it."""

    def __init__(self, num_classes: int = 20, loss_scale: float = 1024.0):
        super().__init__()
        self.layout = lay = SsdLayout(num_classes)
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
        lay = self.layout
        sd = dict(self.named_parameters())
        sd.update(dict(self.named_buffers()))
        with torch.no_grad():
            for key in lay.construction:
                sl = lay.slots[key + ".weight"]
                w = torch.empty(sl.shape)
                nn.init.kaiming_uniform_(w, a=math.sqrt(5))
                sd[key + ".weight"].copy_(w)
                bb = torch.empty(lay.slots[key + ".bias"].shape)
                bound = 1.0 / math.sqrt(sl.shape[1] * sl.shape[2] * sl.shape[3])
                nn.init.uniform_(bb, -bound, bound)
                sd[key + ".bias"].copy_(bb)
            for key, sl in lay.slots.items():
                stem, leaf = key.rsplit(".", 1)
                if (stem + ".running_mean") in lay.slots:
                    if leaf in ("weight", "running_var"):
                        sd[key].fill_(1.0)
                    elif leaf in ("bias", "running_mean"):
                        sd[key].zero_()
            sd["l2_norm.weight"].fill_(20.0)
            self._flat["nbt"].zero_()

    def engine_for(self, h: int, w: int) -> Engine:
        dev = self._flat["param"].device
        key = (h, w, dev)
        eng = self._engines.get(key)
        if eng is None:
            if dev.type != "cuda":
                raise L.CvxError("SSD300VGG runs on an MI355X only: move the model with .to('cuda') first (there is no CPU fallback)")
            eng = Engine(build_ssd_graph(self.layout, h, w), dev)
            eng.set_bn(BN_EPS, BN_MOMENTUM)
            self._engines[key] = eng
        eng.bind(self._flat["param"], self.flat_grads if self.training else self._flat["grad"], self._flat["stat"])
        return eng

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

    def _run_forward(self, x: torch.Tensor, training: bool) -> torch.Tensor:
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected images of shape (B, 3, 300, 300)")
        eng = self.engine_for(int(x.shape[2]), int(x.shape[3]))
        self._last_engine = eng
        rows = eng.forward(x, training)
        if training:
            self._flat["nbt"] += 1
        return rows

    def forward_rows(self, x: torch.Tensor) -> torch.Tensor:
        return self._run_forward(x, self.training)

    def _levels(self):
        """(a_off, hw, boxes per pixel, offset in the flattened loc tensor, offset in the flattened conf tensor) per source level."""
        g, nc1 = self._last_engine.graph, self.num_classes + 1
        out, a_off, lo, co = [], 0, 0, 0
        for (hh, ww), n in zip(g.level_hw, BOXES_PER_PIXEL):
            out.append((a_off, hh * ww, n, lo, co))
            a_off += hh * ww
            lo += hh * ww * n * 4
            co += hh * ww * n * nc1
        return out, lo // 4

    def _rows_to_outputs(self, rows: torch.Tensor):
        B = int(rows.shape[0])
        lay, lib, g = self.layout, L.load(), self._last_engine.graph
        nc1 = self.num_classes + 1
        levels, tot = self._levels()
        loc = torch.empty(B, tot * 4, dtype=torch.float32, device=rows.device)
        conf = torch.empty(B, tot * nc1, dtype=torch.float32, device=rows.device)
        st = L.stream_ptr(rows.device)
        for a_off, hw, n, lo, co in levels:                      # NCHW-order flattening per level, levels concatenated (ssd_model.py:177-183)
            L.check(lib.cvx_pred_cols_to_nchw(L.ptr(rows), lay.pred_ld, 0, n * 4, B, g.anchors, a_off, hw, L.ptr(loc), tot * 4, lo, st), "loc")
            L.check(lib.cvx_pred_cols_to_nchw(L.ptr(rows), lay.pred_ld, LOC_COLS, n * nc1, B, g.anchors, a_off, hw, L.ptr(conf), tot * nc1, co, st),
                    "conf")
        return loc, conf

    def _backward_outputs(self, g_loc, g_conf, scale: Optional[float] = None, run_backward=None):
        """Gradients w.r.t. the flattened (B, tot*4) / (B, tot*(nc+1)) outputs -> loss_scale * dLoss/drows in fp16 -> engine backward."""
        eng, lay, lib = self._last_engine, self.layout, L.load()
        scale = self.loss_scale if scale is None else float(scale)
        nc1 = self.num_classes + 1
        levels, tot = self._levels()
        ref = g_loc if g_loc is not None else g_conf
        B = int(ref.shape[0])
        dpred = torch.zeros(B, eng.graph.anchors, lay.pred_ld, dtype=torch.float16, device=ref.device)
        st = L.stream_ptr(ref.device)
        gl = None if g_loc is None else g_loc.contiguous().float()
        gc = None if g_conf is None else g_conf.contiguous().float()
        for a_off, hw, n, lo, co in levels:
            if gl is not None:
                L.check(lib.cvx_nchw_cols_grad_to_pred(L.ptr(gl), tot * 4, lo, n * 4, B, eng.graph.anchors, a_off, hw, scale, L.ptr(dpred),
                                                       lay.pred_ld, 0, st), "loc grad")
            if gc is not None:
                L.check(lib.cvx_nchw_cols_grad_to_pred(L.ptr(gc), tot * nc1, co, n * nc1, B, eng.graph.anchors, a_off, hw, scale,
                                                       L.ptr(dpred), lay.pred_ld, LOC_COLS, st), "conf grad")
        first = next(p for p in self.parameters() if p.requires_grad)
        if first.grad is None:               # optimizer.zero_grad(set_to_none=True) happened (or first step)
            self.flat_grads.zero_()
            self._grads_attached = False
        self.last_dpred = dpred
        if run_backward is not None:                              # (the data-parallel step runs it bucket by bucket)
            run_backward(eng, dpred, scale)
        else:
            eng.backward(dpred, scale)
        if not self._grads_attached or first.grad is None:
            self.attach_grads()

    def forward(self, x: torch.Tensor):
        if self.training and torch.is_grad_enabled():
            loc, conf = _SsdFn.apply(x, self._anchor, self)
        else:
            self.last_rows = self._run_forward(x, self.training)
            loc, conf = self._rows_to_outputs(self.last_rows)
        B, nc1 = int(x.shape[0]), self.num_classes + 1
        return loc.view(B, -1, 4), conf.view(B, -1, nc1)


class _SsdFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, anchor, model):
        ctx.model = model
        ctx.set_materialize_grads(False)
        model.last_rows = model._run_forward(images, training=True)
        return model._rows_to_outputs(model.last_rows)

    @staticmethod
    def backward(ctx, g_loc, g_conf):
        if g_loc is not None or g_conf is not None:
            ctx.model._backward_outputs(g_loc, g_conf)
        return None, None, None


class _MbLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loc, conf, owner, y_true):
        items, dloc, dconf = owner.op(loc.detach(), conf.detach(), y_true)
        ctx.save_for_backward(dloc, dconf)
        owner.last_items = items
        return items[0].reshape(())

    @staticmethod
    def backward(ctx, gout):
        dloc, dconf = ctx.saved_tensors
        return dloc * gout, dconf * gout, None, None


class MultiBoxLoss:
    """``MultiBoxLossV2(neg_pos_ratio, num_classes)`` of the reference (core/loss/multi_box_loss.py:77-192) on the engine:
    ``total, loc_loss, conf_loss = criterion(y_true, y_pred)`` with ``y_pred = model(images) = (loc, conf)`` and ``y_true`` (B, 8732,
    4 + (nc + 1) + 1) as ``ssd_collate`` encodes it.  Values and the gradient w.r.t. (loc, conf) come from ``cvx_multibox_loss``
    (radix-select hard-negative mining, no sort); ``total.backward()`` hands them to the engine's backward pass."""

    def __init__(self, neg_pos_ratio: float, num_classes: int, alpha: float = 0.5):
        self.neg_pos_ratio, self.nc1, self.alpha = float(neg_pos_ratio), int(num_classes) + 1, float(alpha)
        self._ws = None
        self.last_items = None

    def op(self, loc: torch.Tensor, conf: torch.Tensor, y_true: torch.Tensor, grad_scale: float = 1.0):
        """-> (items (3,): total, loc, conf; dloc, dconf = grad_scale * gradients, fp32)"""
        if loc.device.type != "cuda":
            raise L.CvxError("MultiBoxLoss runs on an MI355X only (there is no CPU path)")
        lib = L.load()
        B, A = int(loc.shape[0]), int(loc.shape[1])
        if tuple(conf.shape) != (B, A, self.nc1) or tuple(y_true.shape) != (B, A, 4 + self.nc1 + 1):
            raise ValueError(f"expected conf {(B, A, self.nc1)} and y_true {(B, A, 4 + self.nc1 + 1)}")
        loc, conf = loc.contiguous().float(), conf.contiguous().float()
        y_true = y_true.to(loc.device).contiguous().float()
        need = int(lib.cvx_multibox_loss_workspace_bytes(B, A))
        if self._ws is None or self._ws.numel() < need or self._ws.device != loc.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=loc.device)
        items, dloc, dconf = torch.empty(3, device=loc.device), torch.empty_like(loc), torch.empty_like(conf)
        L.check(lib.cvx_multibox_loss(L.ptr(loc), L.ptr(conf), L.ptr(y_true), B, A, self.nc1, self.neg_pos_ratio, self.alpha, float(grad_scale),
                                      L.ptr(items), L.ptr(dloc), L.ptr(dconf), L.ptr(self._ws), L.stream_ptr(loc.device)), "cvx_multibox_loss")
        return items, dloc, dconf

    def __call__(self, y_true, y_pred):
        loc, conf = y_pred
        if torch.is_grad_enabled() and (loc.requires_grad or conf.requires_grad):
            total = _MbLossFn.apply(loc, conf, self, y_true)
            return total, self.last_items[1], self.last_items[2]
        items = self.op(loc, conf, y_true)[0]
        return items[0], items[1], items[2]


class SsdTrainStep:
    """One optimisation step of the reference's ``SsdTrainer.train_loop`` (core/trainer/ssd_train.py: zero_grad -> forward ->
    MultiBoxLossV2 -> backward -> Adam under AMP) as C-ABI calls: engine forward (training), the NCHW-order flattening, ``cvx_multibox_loss``,
    its adjoint onto the prediction rows, engine backward, [gradient sum over the ranks], fused Adam with GradScaler's inf/nan check."""

    def __init__(self, model: SSD300VGG, criterion: MultiBoxLoss, optimizer, scaler=None, process_group=None, n_buckets: int = 4):
        self.model, self.criterion, self.optimizer, self.scaler = model, criterion, optimizer, scaler
        self.pg, self.n_buckets = process_group, n_buckets
        self.world, self.distributed = 1, False
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
            self.distributed = True
        self._side = None

    def __call__(self, images: torch.Tensor, y_true: torch.Tensor) -> torch.Tensor:
        from .engine import check_finite
        m, crit = self.model, self.criterion
        if not m.training:
            raise L.CvxError("SsdTrainStep: call model.train() first")
        dev = m.flat_params.device
        self.optimizer.sync_lr()
        B, nc1 = int(images.shape[0]), m.num_classes + 1
        m.last_rows = m._run_forward(images, True)
        loc, conf = m._rows_to_outputs(m.last_rows)
        scale = self.scaler.begin_step() if self.scaler is not None else m.loss_scale
        items, dloc, dconf = crit.op(loc.view(B, -1, 4), conf.view(B, -1, nc1), y_true)
        if self.distributed and dev.type == "cuda":               # gradient exchange overlapped with the backward pass, bucket by bucket
            if self._side is None:
                from .train import OverlappedExchange
                self._side = OverlappedExchange(self.pg, self.n_buckets)
            m._backward_outputs(dloc.view(B, -1), dconf.view(B, -1), scale,
                                run_backward=lambda eng, dpred, sc: self._side.backward(eng, m.flat_grads, dpred, sc))
        else:
            m._backward_outputs(dloc.view(B, -1), dconf.view(B, -1), scale)
        if self.scaler is not None:
            check_finite(m.flat_grads, self.scaler.found_inf)
            self.optimizer.found_inf = self.scaler.found_inf
        self.optimizer.step(zero_grad=True, grad_scale=1.0 / self.world)
        if self.scaler is not None:
            self.scaler.end_step()
        return items
