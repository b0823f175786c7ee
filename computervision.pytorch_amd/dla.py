"""CenterNet (DLA-34) on the MI355X engine -- INFERENCE path (BASELINE.json configs[3]).

Mirrors ``core/models/centernet_model.py:9-379`` of the reference as an engine graph:

* every Conv + BatchNorm (+ ReLU) is one convolution launch with the running statistics folded into its epilogue; the
  BasicBlock's ``relu(bn(conv) + residual)`` is the same launch (residual added before the activation);
* ``Root`` concatenations are gone: both BasicBlocks of a tree write straight into channel slices of the root's input
  buffer; the IDAUp nodes' two inputs are produced into the halves of one buffer (the depthwise transposed convolution
  writes its half directly);
* the three heads' 3x3 convolutions read the same tensor and run as ONE convolution with 3 x 256 output channels (their
  weights are adjacent in the arena, separate ``state_dict`` entries); the three 1x1 outputs land in one fp32 tensor
  (B, H/4 * W/4, [heatmap padded to 8 | wh 8 | reg 8]) that the decode kernel reads in place;
* ``Tree.project`` of a two-level tree is dead code in the reference (its result is overwritten before use,
  centernet_model.py:141-146) and is not executed; ``base.final`` (the unused ImageNet classifier) is kept as parameters only.

All parameters live in one flat fp32 arena, BN statistics in a second one; ``state_dict`` has the reference's 326 keys
and shapes in its order and is bit-identical to ``CenterNet(cfg)`` under the same global seed.  Training mode is not
built this round: ``model.train()`` forward raises.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib as L
from .engine import Engine
from .graph import Graph, TensorSlot

LEVELS = (1, 1, 1, 2, 2, 1)                      # dla34 (centernet_model.py:342-344)
CHANNELS = (16, 32, 64, 128, 256, 512)
HEAD_CONV = 256
BN_EPS, BN_MOMENTUM = 1e-5, 0.1                  # nn.BatchNorm2d defaults (the reference sets nothing else)
HEADS = ("heatmap", "wh", "reg")


def _tree_plan(prefix, levels, cin, cout, stride, level_root, root_dim=0):
    """Tree.__init__ (centernet_model.py:98-136) as data."""
    if root_dim == 0:
        root_dim = 2 * cout
    if level_root:
        root_dim += cin
    t = dict(kind="tree", prefix=prefix, levels=levels, cin=cin, cout=cout, stride=stride, level_root=level_root, root_dim=root_dim,
             project=cin != cout)
    if levels == 1:
        t["tree1"] = dict(kind="block", prefix=prefix + ".tree1", cin=cin, cout=cout, stride=stride)
        t["tree2"] = dict(kind="block", prefix=prefix + ".tree2", cin=cout, cout=cout, stride=1)
    else:
        t["tree1"] = _tree_plan(prefix + ".tree1", levels - 1, cin, cout, stride, False, 0)
        t["tree2"] = _tree_plan(prefix + ".tree2", levels - 1, cout, cout, 1, False, root_dim + cout)
    return t


def _trees():
    b, c = "backbone.base.", CHANNELS
    return [_tree_plan(b + "level_2", LEVELS[2], c[1], c[2], 2, False), _tree_plan(b + "level_3", LEVELS[3], c[2], c[3], 2, True),
            _tree_plan(b + "level_4", LEVELS[4], c[3], c[4], 2, True), _tree_plan(b + "level_5", LEVELS[5], c[4], c[5], 2, True)]


def _dla_up_plan():
    """DLAUp.__init__ with its in-place list updates (centernet_model.py:282-296) -> [(out_dim, in_channels, up_factors)]."""
    channels, in_ch, scales, idas = list(CHANNELS[2:]), list(CHANNELS[2:]), [1, 2, 4, 8], []
    for i in range(len(channels) - 1):
        j = -i - 2
        idas.append((channels[j], list(in_ch[j:]), [s // scales[j] for s in scales[j:]]))
        scales[j + 1:] = [scales[j]] * len(scales[j + 1:])
        in_ch[j + 1:] = [channels[j]] * len(in_ch[j + 1:])
    return idas


class DlaLayout:
    """Arena offsets for every tensor of the reference's CenterNet ``state_dict`` (same keys, shapes, order)."""

    def __init__(self, nc: int = 80):
        self.nc = nc
        self.nc_pad = (nc + 7) & ~7
        self.pred_ld = self.nc_pad + 16                 # [heatmap | wh (2 of 8) | reg (2 of 8)]
        self.slots: "OrderedDict[str, TensorSlot]" = OrderedDict()
        self.nbt_keys: List[str] = []
        self.convs: Dict[str, dict] = {}               # conv key -> offsets for the engine op
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

    # conv (+ optional bias) whose weights the engine reads as [cout][kh][kw][cin]
    def conv(self, key, cout, cin, k, bias=False, w_off=None, b_off=None):
        ce = (cout + 7) & ~7
        spec = dict(cout=cout, cout_eng=ce, cin=cin, k=k,
                    w_off=self._take("param", ce * k * k * cin) if w_off is None else w_off)
        self.slots[key + ".weight"] = TensorSlot("param", spec["w_off"], (cout, cin, k, k), (k * k * cin, 1, k * cin, cin))
        if bias:
            spec["bias_off"] = self._take("param", ce) if b_off is None else b_off
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

    def _block(self, p):
        self.conv_bn(p["prefix"] + ".conv1", p["prefix"] + ".bn1", p["cout"], p["cin"], 3)
        self.conv_bn(p["prefix"] + ".conv2", p["prefix"] + ".bn2", p["cout"], p["cout"], 3)

    def _tree(self, t):
        for sub in (t["tree1"], t["tree2"]):
            (self._block if sub["kind"] == "block" else self._tree)(sub)
        if t["levels"] == 1:
            self.conv_bn(t["prefix"] + ".root.conv", t["prefix"] + ".root.bn", t["cout"], t["root_dim"], 1)
        if t["project"]:
            self.conv_bn(t["prefix"] + ".project.0", t["prefix"] + ".project.1", t["cout"], t["cin"], 1)

    def _plan(self):
        b, c = "backbone.base.", CHANNELS
        self.conv_bn(b + "base_layer.0", b + "base_layer.1", c[0], 3, 7)
        self.conv_bn(b + "level_0.0", b + "level_0.1", c[0], c[0], 3)
        self.conv_bn(b + "level_1.0", b + "level_1.1", c[1], c[0], 3)
        for t in _trees():
            self._tree(t)
        # the ImageNet classifier of the DLA constructor: parameters only (never executed by DLASeg)
        off = self._take("param", 1000 * c[5])
        self.slots[b + "final.weight"] = TensorSlot("param", off, (1000, c[5], 1, 1), (c[5], 1, 1, 1))
        self.slots[b + "final.bias"] = TensorSlot("param", self._take("param", 1000), (1000,), (1,))
        for i, (out_dim, in_ch, ups) in enumerate(_dla_up_plan()):
            p = f"backbone.dla_up.ida_{i}."
            for k, (ci, f) in enumerate(zip(in_ch, ups)):
                if ci != out_dim:
                    self.conv_bn(p + f"proj_{k}.0", p + f"proj_{k}.1", out_dim, ci, 1)
                if f != 1:
                    kk = 2 * f
                    off = self._take("param", out_dim * kk * kk)
                    self.slots[p + f"up_{k}.weight"] = TensorSlot("param", off, (out_dim, 1, kk, kk), (kk * kk, kk * kk, kk, 1))
                    self.convs[p + f"up_{k}"] = dict(w_off=off, f=f, c=out_dim)
            for k in range(1, len(in_ch)):
                self.conv_bn(p + f"node_{k}.0", p + f"node_{k}.1", out_dim, 2 * out_dim, 3)
        # heads: the three 3x3 convs share their input -> adjacent arena blocks, one engine conv with 3 x 256 channels;
        # the slots keep the reference's key order (heatmap.0, heatmap.2, wh.0, wh.2, reg.0, reg.2)
        w0 = self._take("param", 3 * HEAD_CONV * 9 * c[2])
        b0 = self._take("param", 3 * HEAD_CONV)
        for n, (head, classes) in enumerate(zip(HEADS, (self.nc, 2, 2))):
            self.conv(f"backbone.{head}.0", HEAD_CONV, c[2], 3, True, w_off=w0 + n * HEAD_CONV * 9 * c[2], b_off=b0 + n * HEAD_CONV)
            self.conv(f"backbone.{head}.2", classes, HEAD_CONV, 1, True)
        self.head_first = dict(w_off=w0, bias_off=b0, cout=3 * HEAD_CONV, cin=c[2], k=3)

    # -- reference-keyed dicts <-> flat arenas -------------------------------------------------------
    def views(self, arena, which="param"):
        return {k: torch.as_strided(arena, sl.shape, sl.strides, sl.offset) for k, sl in self.slots.items() if sl.arena == which}


# --------------------------------------------------------------------------------------------------
def build_dla_graph(lay: DlaLayout, H: int, W: int) -> Graph:
    """Buffer plan + op list for an (H, W) input (multiples of 32)."""
    if H % 32 or W % 32:
        raise ValueError("input height/width must be multiples of 32")
    g = Graph()
    c = CHANNELS

    def buf(h, w, ch, kind=L.BUF_ACT_F16):
        g.bufs.append((h, w, ch, kind))
        return len(g.bufs) - 1

    def V(b, off, ch, pix=0):
        return (b, off, ch, pix)

    def conv(ckey, vin, vout, hin, win, stride, act, res=None, res_pre=False, spec=None, name=None):
        s = spec or lay.convs[ckey]
        k = s["k"]
        ho, wo = (hin + 2 * (k // 2) - k) // stride + 1, (win + 2 * (k // 2) - k) // stride + 1
        op = dict(type=L.OP_CONV, name=name or ckey, out=vout, ih=hin, iw=win, oh=ho, ow=wo, k=k, stride=stride, pad=k // 2, dil=1, act=act,
                  needs_dgrad=0 if vin[0] == img else 1, w_cin=s["cin"], w_off=s["w_off"], gamma_off=s.get("gamma_off", 0), beta_off=s.get("beta_off", 0),
                  bias_off=s.get("bias_off", 0), rmean_off=s.get("rmean_off", 0), rvar_off=s.get("rvar_off", 0),
                  flags=L.OPF_RES_PRE_ACT if res_pre else 0)
        op["in"] = vin
        if res is not None:
            op["res"] = res
        g.ops.append(op)
        return ho, wo

    def simple(kind, name, vin, vout, ih, iw, oh, ow, **kw):
        op = dict(type=kind, name=name, out=vout, ih=ih, iw=iw, oh=oh, ow=ow, **kw)
        op["in"] = vin
        g.ops.append(op)

    def block(p, vin, h, w, residual, vout):
        """BasicBlock (centernet_model.py:9-27): relu(bn2(conv2(relu(bn1(conv1(x))))) + residual)."""
        s = p["stride"]
        tmp = buf(h // s, w // s, p["cout"])
        conv(p["prefix"] + ".conv1", vin, V(tmp, 0, p["cout"]), h, w, s, L.ACT_BN_RELU)
        conv(p["prefix"] + ".conv2", V(tmp, 0, p["cout"]), vout, h // s, w // s, 1, L.ACT_BN_RELU, res=residual, res_pre=True)

    def tree(t, vin, h, w, vout, children=()):
        """Tree.forward (centernet_model.py:138-152); returns nothing, the result is written to `vout`."""
        s, co = t["stride"], t["cout"]
        ho, wo = h // s, w // s
        children = list(children)
        if t["levels"] > 1:
            if t["level_root"]:
                bottom = buf(ho, wo, t["cin"])
                simple(L.OP_MAXPOOL2, t["prefix"] + ".downsample", vin, V(bottom, 0, t["cin"]), h, w, ho, wo)
                children.append(V(bottom, 0, t["cin"]))
            x1 = buf(ho, wo, co)
            tree(t["tree1"], vin, h, w, V(x1, 0, co))
            children.append(V(x1, 0, co))
            tree(t["tree2"], V(x1, 0, co), ho, wo, vout, children)
            return
        cat = buf(ho, wo, t["root_dim"])                        # [x2 | x1 | children...]   (Root: torch.cat(inputs, 1))
        off = 2 * co
        if s > 1:
            if t["level_root"]:                                   # the pooled input is a child: pool straight into its slice
                bottom = V(cat, off, t["cin"])
                off += t["cin"]
            else:
                bottom = V(buf(ho, wo, t["cin"]), 0, t["cin"])
            simple(L.OP_MAXPOOL2, t["prefix"] + ".downsample", vin, bottom, h, w, ho, wo)
        else:
            bottom = vin
            assert not t["level_root"]
        if t["project"]:
            res = V(buf(ho, wo, co), 0, co)
            conv(t["prefix"] + ".project.0", bottom, res, ho, wo, 1, L.ACT_BN_LINEAR)
        else:
            res = bottom
        x1v, x2v = V(cat, co, co), V(cat, 0, co)
        block(t["tree1"], vin, h, w, res, x1v)
        block(t["tree2"], x1v, ho, wo, x1v, x2v)
        for ch in children:
            simple(L.OP_COPY, t["prefix"] + ".root.cat", ch, V(cat, off, ch[2]), ho, wo, ho, wo)
            off += ch[2]
        assert off == t["root_dim"], (t["prefix"], off, t["root_dim"])
        conv(t["prefix"] + ".root.conv", V(cat, 0, t["root_dim"]), vout, ho, wo, 1, L.ACT_BN_RELU)

    b = "backbone.base."
    img = buf(H, W, 8)
    g.image_buf = img
    base, l0, l1 = buf(H, W, c[0]), buf(H, W, c[0]), buf(H // 2, W // 2, c[1])
    conv(b + "base_layer.0", V(img, 0, 8), V(base, 0, c[0]), H, W, 1, L.ACT_BN_RELU)
    conv(b + "level_0.0", V(base, 0, c[0]), V(l0, 0, c[0]), H, W, 1, L.ACT_BN_RELU)
    conv(b + "level_1.0", V(l0, 0, c[0]), V(l1, 0, c[1]), H, W, 2, L.ACT_BN_RELU)
    levels, cur, h, w = [], V(l1, 0, c[1]), H // 2, W // 2
    for t in _trees():
        out = V(buf(h // 2, w // 2, t["cout"]), 0, t["cout"])
        tree(t, cur, h, w, out)
        cur, h, w = out, h // 2, w // 2
        levels.append((out, h, w))

    # DLAUp (centernet_model.py:298-305) over IDAUp (:268-279)
    layers = list(levels)
    for i, (out_dim, in_ch, ups) in enumerate(_dla_up_plan()):
        p = f"backbone.dla_up.ida_{i}."
        ins = layers[-i - 2:]
        (x, hh, ww) = ins[0]                                     # proj_0 / up_0 are Identity
        ys = []
        for k in range(1, len(ins)):
            cat = buf(hh, ww, 2 * out_dim)
            simple(L.OP_COPY, p + f"node_{k}.cat", x, V(cat, 0, out_dim), hh, ww, hh, ww)
            (src, sh_, sw_), ci, f = ins[k], in_ch[k], ups[k]
            if ci != out_dim:
                pr = V(buf(sh_, sw_, out_dim), 0, out_dim)
                conv(p + f"proj_{k}.0", src, pr, sh_, sw_, 1, L.ACT_BN_RELU)
                src = pr
            up = lay.convs[p + f"up_{k}"]
            simple(L.OP_DWCONVT, p + f"up_{k}", src, V(cat, out_dim, out_dim), sh_, sw_, hh, ww, stride=f, w_off=up["w_off"])
            x = V(buf(hh, ww, out_dim), 0, out_dim)
            conv(p + f"node_{k}.0", V(cat, 0, 2 * out_dim), x, hh, ww, 1, L.ACT_BN_RELU)
            ys.append((x, hh, ww))
        layers[-i - 1:] = ys
    feat, fh, fw = x, H // 4, W // 4
    assert (hh, ww) == (fh, fw)

    # heads (centernet_model.py:311-327, 371-379): fused 3x3 (bias + ReLU), three 1x1 (+bias) into the fp32 output tensor
    hid = buf(fh, fw, 3 * HEAD_CONV)
    conv(None, feat, V(hid, 0, 3 * HEAD_CONV), fh, fw, 1, L.ACT_BIAS_RELU, spec=lay.head_first, name="backbone.heads.0")
    pred = buf(fh * fw, 1, lay.pred_ld, L.BUF_PRED_F32)
    g.pred_buf = pred
    col = 0
    for n, head in enumerate(HEADS):
        s = lay.convs[f"backbone.{head}.2"]
        conv(f"backbone.{head}.2", V(hid, n * HEAD_CONV, HEAD_CONV), V(pred, col, s["cout_eng"], 0), fh, fw, 1, L.ACT_BIAS)
        col += s["cout_eng"]
    g.level_hw = [(fh, fw)]
    g.anchors = fh * fw
    return g


# --------------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise L.CvxError("parameter holder: the engine executes the whole graph (call the CenterNet model)")


class CenterNetDLA34(nn.Module):
    """``CenterNet(cfg)`` of the reference (centernet_model.py:365-379) on the engine: ``model(x)`` returns the (B, H/4, W/4, nc + 4)
    NHWC tensor [heatmap | wh | reg].  In training mode (grad enabled) the tensor is connected to the engine's backward pass
    (batch-statistics BatchNorm + ReLU, the BasicBlock residual inside the ReLU, 2x2 max pools, the depthwise transposed
    convolutions' data and weight gradients, the concat copies, the biased head convolutions): any torch loss on it -- the
    reference's CombinedLoss is torch code on exactly this tensor -- trains the network."""

    def __init__(self, num_classes: int = 80, loss_scale: float = 1024.0):
        super().__init__()
        self.layout = lay = DlaLayout(num_classes)
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
        self.last_raw = None

    # ---- module tree with the reference's names (state_dict keys / order) --------------------------
    def _build_tree(self):
        for key in self.layout.slots:
            mod = self
            parts = key.split(".")
            for name in parts[:-1]:
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
        """torch's default Conv2d / ConvTranspose2d / BatchNorm2d initialisation, drawn from the global RNG in the reference's
        construction order (= its state_dict order: a conv's weight, then its bias if it has one)."""
        sd = dict(self.named_parameters())
        sd.update(dict(self.named_buffers()))
        slots = self.layout.slots
        with torch.no_grad():
            for key, sl in slots.items():
                stem, leaf = key.rsplit(".", 1)
                is_bn = (stem + ".running_mean") in slots
                if leaf == "weight" and not is_bn:
                    w = torch.empty(sl.shape)
                    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
                    sd[key].copy_(w)
                    if (stem + ".bias") in slots:
                        bound = 1.0 / math.sqrt(sl.shape[1] * sl.shape[2] * sl.shape[3])
                        bb = torch.empty(slots[stem + ".bias"].shape)
                        nn.init.uniform_(bb, -bound, bound)
                        sd[stem + ".bias"].copy_(bb)
                elif is_bn and leaf in ("weight", "running_var"):
                    sd[key].fill_(1.0)
                elif is_bn and leaf in ("bias", "running_mean"):
                    sd[key].zero_()
            self._flat["nbt"].zero_()

    # ---- engine plumbing ---------------------------------------------------------------------------------
    def engine_for(self, h: int, w: int) -> Engine:
        dev = self._flat["param"].device
        key = (h, w, dev)
        eng = self._engines.get(key)
        if eng is None:
            if dev.type != "cuda":
                raise L.CvxError("CenterNetDLA34 runs on an MI355X only: move the model with .to('cuda') first (there is no CPU fallback)")
            eng = Engine(build_dla_graph(self.layout, h, w), dev)
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
            raise ValueError("expected images of shape (B, 3, H, W)")
        eng = self.engine_for(int(x.shape[2]), int(x.shape[3]))
        self._last_engine = eng
        raw = eng.forward(x, training)
        if training:
            self._flat["nbt"] += 1
        return raw

    def forward_raw(self, x: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) -> the engine's fp32 head tensor (B, H/4 * W/4, nc_pad + 16): what ``centernet_decode`` reads."""
        return self._run_forward(x, self.training)

    def _backward_rows(self, g: torch.Tensor):
        """Gradient w.r.t. the fp32 head rows -> loss_scale * g in fp16 -> engine backward (parameter gradients accumulate)."""
        first = next(p for p in self.parameters() if p.requires_grad)
        if first.grad is None:               # optimizer.zero_grad(set_to_none=True) happened (or first step)
            self.flat_grads.zero_()
            self._grads_attached = False
        self.last_dpred = (g * self.loss_scale).to(torch.float16).contiguous()
        self._last_engine.backward(self.last_dpred, self.loss_scale)
        if not self._grads_attached or first.grad is None:
            self.attach_grads()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raw = _RowsFn.apply(x, self._anchor, self) if (self.training and torch.is_grad_enabled()) else self.forward_raw(x)
        self.last_raw = raw.detach()
        B, _, H, W = x.shape
        lay = self.layout
        out = torch.cat((raw[..., :lay.nc], raw[..., lay.nc_pad:lay.nc_pad + 2], raw[..., lay.nc_pad + 8:lay.nc_pad + 10]), -1)
        out = out.reshape(B, H // 4, W // 4, lay.nc + 4)
        out.model = self                       # the fused CenterNetLoss starts from the head rows (last_raw)
        return out


class _RowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, anchor, model):
        ctx.model = model
        ctx.set_materialize_grads(False)
        return model._run_forward(images, training=True)

    @staticmethod
    def backward(ctx, g):
        if g is not None:
            ctx.model._backward_rows(g)
        return None, None, None


class _CnLossFn(torch.autograd.Function):
    """loss = CombinedLoss(rows, targets) with the gradient taken by the fused kernel; ``.backward()`` runs the engine's backward."""

    @staticmethod
    def forward(ctx, out, owner, model, targets):
        items, dpred = owner.op(model.last_raw.detach(), targets, model._last_engine.graph.level_hw[0], model.loss_scale)
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
        return None, None, None, None


class CenterNetLoss:
    """``CombinedLoss(num_classes, hm_weight, wh_weight, off_weight)`` of the reference (core/loss/centernet_loss.py:46-67) on the engine:
    ``loss = criterion(preds, targets)`` with ``preds = model(images)`` and ``targets = [heatmap_true (B,h,w,nc), reg_true (B,K,2),
    wh_true (B,K,2), reg_mask (B,K), indices (B,K)]`` as ``centernet_collate`` builds them.  Value and gradient come from
    ``cvx_centernet_loss`` on the head rows behind ``preds``; ``loss.backward()`` then runs the engine's backward pass.  The
    reference applies its "reg" term to the output's columns nc, nc+1 and its "wh" term to the last two -- kept as is."""

    def __init__(self, num_classes: int, hm_weight: float = 1.0, wh_weight: float = 0.1, off_weight: float = 1.0, check_targets: bool = True):
        self.num_classes = int(num_classes)
        self.hm_weight, self.wh_weight, self.off_weight = float(hm_weight), float(wh_weight), float(off_weight)
        self.check_targets = check_targets
        self._ws = self._bad = None
        self.last_items = None

    def bad_targets(self) -> bool:
        return self._bad is not None and int(self._bad.item()) != 0

    def op(self, rows: torch.Tensor, targets: Sequence[torch.Tensor], hw, loss_scale: float, dpred: Optional[torch.Tensor] = None,
           check: Optional[bool] = None):
        """rows (B, h*w, ld) fp32 -> (loss items (4,): total, heat-map, L1 "reg", L1 "wh"; dpred (B, h*w, ld) fp16)."""
        if rows.device.type != "cuda":
            raise L.CvxError("CenterNetLoss runs on an MI355X only (there is no CPU path)")
        lib = L.load()
        B, A, ld = rows.shape
        nc = self.num_classes
        nc_pad = (nc + 7) & ~7
        heat, reg_t, wh_t, mask, idx = (t.to(rows.device) for t in targets)
        if tuple(heat.shape) != (B, hw[0], hw[1], nc):
            raise ValueError(f"heatmap_true must have shape {(B, hw[0], hw[1], nc)}")
        K = int(mask.shape[1])
        heat, reg_t, wh_t, mask = (t.float().contiguous() for t in (heat, reg_t, wh_t, mask))
        idx = idx.long().contiguous()
        need = int(lib.cvx_centernet_loss_workspace_bytes(B, A))
        if self._ws is None or self._ws.numel() < need or self._ws.device != rows.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=rows.device)
            self._bad = torch.zeros(1, dtype=torch.int32, device=rows.device)
        if dpred is None:
            dpred = torch.empty(B, A, ld, dtype=torch.float16, device=rows.device)
        items = torch.empty(4, device=rows.device)
        # model output = [heatmap | wh head | reg head]; the loss's "reg" = output[..., nc:nc+2] (rows' columns nc_pad..), "wh" = the last two
        L.check(lib.cvx_centernet_loss(L.ptr(rows), ld, B, A, nc, nc_pad, nc_pad + 8, L.ptr(heat), L.ptr(reg_t), L.ptr(wh_t), L.ptr(mask), L.ptr(idx), K,
                                       self.hm_weight, self.off_weight, self.wh_weight, float(loss_scale), L.ptr(items), L.ptr(dpred),
                                       L.ptr(self._bad), L.ptr(self._ws), L.stream_ptr(rows.device)), "cvx_centernet_loss")
        if (self.check_targets if check is None else check) and self.bad_targets():
            raise L.CvxError("CenterNetLoss: a masked object's index lies outside the feature map")
        return items, dpred

    def __call__(self, preds: torch.Tensor, targets):
        model = getattr(preds, "model", None)
        if model is None:
            raise L.CvxError("CenterNetLoss needs the output of CenterNetDLA34.forward (it carries the head rows the loss starts from)")
        if model.training and torch.is_grad_enabled():
            return _CnLossFn.apply(preds, self, model, targets)
        return self.op(model.last_raw, targets, model._last_engine.graph.level_hw[0], model.loss_scale)[0][0].reshape(())


class CenterNetTrainStep:
    """One optimisation step of the reference's ``CenterNetTrainer.train_loop`` (core/trainer/centernet_train.py:104-121) as C-ABI calls:
    engine forward (training), ``cvx_centernet_loss``, engine backward, [gradient sum over the ranks], fused Adam with GradScaler's
    inf/nan check.  Returns the loss items (4,)."""

    def __init__(self, model: "CenterNetDLA34", criterion: CenterNetLoss, optimizer, scaler=None, process_group=None, n_buckets: int = 4):
        self.model, self.criterion, self.optimizer, self.scaler = model, criterion, optimizer, scaler
        self.pg, self.n_buckets = process_group, n_buckets
        self.world, self.distributed = 1, False
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
            self.distributed = True
        self._dpred = self._side = None

    def __call__(self, images: torch.Tensor, targets) -> torch.Tensor:
        from .engine import check_finite
        m, crit = self.model, self.criterion
        if not m.training:
            raise L.CvxError("CenterNetTrainStep: call model.train() first")
        dev = m.flat_params.device
        self.optimizer.sync_lr()
        rows = m._run_forward(images, True)
        m.last_raw = rows
        eng = m._last_engine
        if self._dpred is None or self._dpred.shape != rows.shape:
            self._dpred = torch.empty(rows.shape, device=dev, dtype=torch.float16)
        scale = self.scaler.begin_step() if self.scaler is not None else m.loss_scale
        items, dpred = crit.op(rows, targets, eng.graph.level_hw[0], scale, self._dpred, check=False)
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
