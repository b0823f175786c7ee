"""YOLOv8 as an engine graph: concat-free NHWC buffer plan, op list and flat parameter arenas.

Mirrors the hand-unrolled 23-module network of the reference (core/models/yolov8/yolo_v8.py:26-107;
blocks in core/models/yolov8/modules.py:19-33,124-135,189-202,304-318,407-455) but lays it out for
the MI355X engine:

* every ``torch.cat`` of the reference disappears: producers write straight into channel slices of
  the consumer's NHWC buffer (C2f split/concat, SPPF pyramid, the four FPN/PAN concats);
* the two 3x3 convs that open each Detect level (box branch ``cv2[i][0]`` and class branch
  ``cv3[i][0]``) read the same input, so they run as ONE conv with 64+c3 output channels -- their
  weights / BN parameters are laid out adjacently in the arena, and stay separate ``state_dict`` entries;
* all trainable tensors live in one flat fp32 arena (conv weights in [cout][kh][kw][cin] order), all
  BN running statistics in a second one.  ``state_dict`` tensors are strided views of these arenas
  with the reference's keys and logical (cout, cin, kh, kw) shapes, gradients are one flat buffer
  (one RCCL all-reduce, one fused Adam launch).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

from . import _lib as L

SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768), "l": (1.00, 1.00, 512),
          "x": (1.00, 1.25, 512)}                              # yolo_v8.py:110-132
REG_MAX = 16                                                   # modules.py:419
STRIDES = (8, 16, 32)


def _ch(c: int, width: float, max_ch: int) -> int:
    """make_divisible(min(c, max_channels) * width, 8)   (yolo_v8.py:67-76)."""
    return int(math.ceil(min(c, max_ch) * width / 8) * 8)


def _rep(n: int, depth: float) -> int:
    return max(round(n * depth), 1) if n > 1 else n            # yolo_v8.py:64-65


@dataclass
class TensorSlot:
    arena: str                  # "param" | "stat"
    offset: int
    shape: Tuple[int, ...]      # logical (state_dict) shape
    strides: Tuple[int, ...]    # element strides inside the arena
    trainable: bool = True


@dataclass
class ConvSpec:
    """One engine conv = one or two reference conv modules sharing an input."""
    prefixes: List[str]         # e.g. ["model.22.cv2.0.0", "model.22.cv3.0.0"]
    couts: List[int]
    cin: int                    # stored weight input channels
    k: int
    bn: bool                    # True: Conv2d(no bias)+BN+SiLU, False: Conv2d+bias (head output)
    w_off: int = 0
    gamma_off: int = 0
    beta_off: int = 0
    bias_off: int = 0
    rmean_off: int = 0
    rvar_off: int = 0
    cout_eng: int = 0           # output channels the ENGINE computes: cout rounded up to 8 for the bias convs (zero rows)

    @property
    def cout(self):
        return sum(self.couts)


class ParamLayout:
    """Arena offsets for every tensor of the reference's state_dict (same keys, same shapes)."""

    def __init__(self, model_type: str = "n", nc: int = 80):
        if model_type not in SCALES:
            raise ValueError(f"model_type: {model_type} is not supported")
        self.model_type, self.nc = model_type, nc
        depth, width, max_ch = SCALES[model_type]
        c = lambda x: _ch(x, width, max_ch)  # noqa: E731
        self.c64, self.c128, self.c256, self.c512, self.c1024 = c(64), c(128), c(256), c(512), c(1024)
        self.n3, self.n6 = _rep(3, depth), _rep(6, depth)
        self.head_in = (self.c256, self.c512, self.c1024)
        self.c_box = max(16, self.head_in[0] // 4, REG_MAX * 4)        # modules.py:422
        self.c_cls = max(self.head_in[0], nc)
        self.c_cls_eng = (self.c_cls + 7) & ~7        # hidden width of the class branch as the engine runs it (zero-padded)
        self.no = nc + 4 * REG_MAX
        self.nc_pad = (nc + 7) & ~7                    # class columns of the engine's pred buffer
        self.no_pad = self.nc_pad + 4 * REG_MAX
        self.slots: "OrderedDict[str, TensorSlot]" = OrderedDict()
        self.convs: Dict[str, ConvSpec] = {}
        self._p = 0
        self._s = 0
        self.n_bn = 0
        self._plan()
        self.n_params = (self._p + 3) & ~3
        self.n_stats = (self._s + 3) & ~3

    # -- arena bookkeeping ---------------------------------------------------------------------
    def _take(self, arena: str, n: int) -> int:
        if arena == "param":
            off = self._p
            self._p = (self._p + n + 3) & ~3
        else:
            off = self._s
            self._s = (self._s + n + 3) & ~3
        return off

    def _add_conv(self, name: str, prefixes, couts, cin, k, bn=True) -> ConvSpec:
        spec = ConvSpec(list(prefixes), list(couts), cin, k, bn)
        ct = spec.cout
        # the engine's kernels work on 8-channel groups: an arena block whose channel count is not a multiple of 8 (the head's
        # class convs: nc outputs, and their c3 = max(ch0, nc) wide hidden layers when nc > ch0) is padded with zero rows.
        # Padded channels are inert: zero weights -> zero conv output -> zero normalised value -> silu(beta = 0) = 0, the next
        # layer's weights for them are zero as well, and every gradient that reaches them is zero, so Adam never moves them.
        # The state_dict views cover the real rows only.
        ce = (ct + 7) & ~7
        spec.cout_eng = ce
        spec.w_off = self._take("param", ce * k * k * cin)
        if bn:
            spec.gamma_off = self._take("param", ce)
            spec.beta_off = self._take("param", ce)
            spec.rmean_off = self._take("stat", ce)
            spec.rvar_off = self._take("stat", ce)
        else:
            spec.bias_off = self._take("param", ce)
        self.convs[name] = spec
        # state_dict views, segment by segment (segments are adjacent inside the op's arena block)
        c0 = 0
        for pre, co in zip(prefixes, couts):
            wk = pre + (".conv.weight" if bn else ".weight")
            self.slots[wk] = TensorSlot("param", spec.w_off + c0 * k * k * cin, (co, cin, k, k), (k * k * cin, 1, k * cin, cin))
            if bn:
                self.slots[pre + ".bn.weight"] = TensorSlot("param", spec.gamma_off + c0, (co,), (1,))
                self.slots[pre + ".bn.bias"] = TensorSlot("param", spec.beta_off + c0, (co,), (1,))
                self.slots[pre + ".bn.running_mean"] = TensorSlot("stat", spec.rmean_off + c0, (co,), (1,), False)
                self.slots[pre + ".bn.running_var"] = TensorSlot("stat", spec.rvar_off + c0, (co,), (1,), False)
                self.n_bn += 1
            else:
                self.slots[pre + ".bias"] = TensorSlot("param", spec.bias_off + c0, (co,), (1,))
            c0 += co
        return spec

    # -- reference-keyed dicts <-> flat arenas (checkpoints, tests) ------------------------------------
    def views(self, arena, which: str = "param"):
        """{state_dict key: strided view of `arena`} for the slots of one arena ("param" or "stat")."""
        import torch
        return {k: torch.as_strided(arena, sl.shape, sl.strides, sl.offset) for k, sl in self.slots.items() if sl.arena == which}

    def scatter(self, tensors, which: str = "param"):
        """Flat arena (zero padding) holding `tensors` (reference keys, logical shapes) in the engine's layout."""
        import torch
        arena = torch.zeros(self.n_params if which == "param" else self.n_stats)
        for k, v in self.views(arena, which).items():
            if k in tensors:
                v.copy_(tensors[k])
        return arena

    def _plan(self):
        A = self._add_conv
        c64, c128, c256, c512, c1024 = self.c64, self.c128, self.c256, self.c512, self.c1024

        def c2f(idx, c1, c2, n):
            c = c2 // 2
            A(f"{idx}.cv1", [f"model.{idx}.cv1"], [2 * c], c1, 1)
            A(f"{idx}.cv2", [f"model.{idx}.cv2"], [c2], (2 + n) * c, 1)
            for j in range(n):
                A(f"{idx}.m{j}.cv1", [f"model.{idx}.m.{j}.cv1"], [c], c, 3)
                A(f"{idx}.m{j}.cv2", [f"model.{idx}.m.{j}.cv2"], [c], c, 3)

        A("0", ["model.0"], [c64], 3, 3)
        A("1", ["model.1"], [c128], c64, 3)
        c2f(2, c128, c128, self.n3)
        A("3", ["model.3"], [c256], c128, 3)
        c2f(4, c256, c256, self.n6)
        A("5", ["model.5"], [c512], c256, 3)
        c2f(6, c512, c512, self.n6)
        A("7", ["model.7"], [c1024], c512, 3)
        c2f(8, c1024, c1024, self.n3)
        A("9.cv1", ["model.9.cv1"], [c1024 // 2], c1024, 1)
        A("9.cv2", ["model.9.cv2"], [c1024], (c1024 // 2) * 4, 1)
        c2f(12, c1024 + c512, c512, self.n3)
        c2f(15, c512 + c256, c256, self.n3)
        A("16", ["model.16"], [c256], c256, 3)
        c2f(18, c256 + c512, c512, self.n3)
        A("19", ["model.19"], [c512], c512, 3)
        c2f(21, c512 + c1024, c1024, self.n3)
        for lvl, cin in enumerate(self.head_in):
            A(f"22.{lvl}.0", [f"model.22.cv2.{lvl}.0", f"model.22.cv3.{lvl}.0"], [self.c_box, self.c_cls], cin, 3)
            A(f"22.{lvl}.1b", [f"model.22.cv2.{lvl}.1"], [self.c_box], self.c_box, 3)
            A(f"22.{lvl}.1c", [f"model.22.cv3.{lvl}.1"], [self.c_cls], self.c_cls, 3)
            A(f"22.{lvl}.2b", [f"model.22.cv2.{lvl}.2"], [4 * REG_MAX], self.c_box, 1, bn=False)
            A(f"22.{lvl}.2c", [f"model.22.cv3.{lvl}.2"], [self.nc], self.c_cls, 1, bn=False)


# --------------------------------------------------------------------------------------------------
@dataclass
class Graph:
    bufs: List[Tuple[int, int, int, int]] = field(default_factory=list)
    ops: List[dict] = field(default_factory=list)
    image_buf: int = -1
    pred_buf: int = -1
    level_hw: List[Tuple[int, int]] = field(default_factory=list)
    anchors: int = 0
    taps: Dict[int, Tuple[int, int, int]] = field(default_factory=dict)   # reference layer idx -> (buf, coff, c)

    def c_arrays(self):
        bufs = (L.BufDesc * len(self.bufs))(*[L.BufDesc(*b) for b in self.bufs])
        ops = (L.OpDesc * len(self.ops))()
        for i, o in enumerate(self.ops):
            d = ops[i]
            d.type = o["type"]
            d.in_ = L.View(*o["in"])
            d.out = L.View(*o["out"])
            d.res = L.View(*o.get("res", (-1, 0, 0, 0)))
            d.ih, d.iw, d.oh, d.ow = o["ih"], o["iw"], o["oh"], o["ow"]
            d.k, d.stride, d.pad, d.dil = o.get("k", 1), o.get("stride", 1), o.get("pad", 0), o.get("dil", 1)
            d.act = o.get("act", 0)
            d.needs_dgrad = o.get("needs_dgrad", 1)
            d.w_cin = o.get("w_cin", 0)
            for f in ("w_off", "gamma_off", "beta_off", "bias_off", "rmean_off", "rvar_off"):
                setattr(d, f, o.get(f, 0))
            d.lane = o.get("lane", 0)
            d.flags = o.get("flags", 0)
        return bufs, ops


# Which convs keep their raw output in fp16 during training (CVX_OPF_RAW_F16: 6 instead of 12 bytes per element around the forward BatchNorm):
# * every Conv of the neck and the head (top-level modules 12 .. 22, 62 % of the BatchNorm elements of YOLOv8-n): their rounding does not
#   reach the logits (oracle/fp16_raw_study.py: the per-level error against the fp32 reference moves by <= 0.5 %);
# * backbone Convs outside the Bottlenecks whose output is at least 80 x 80 per image -- where the bytes are (at 640 x 640: 1, 2.cv1, 2.cv2,
#   3, 4.cv1, 4.cv2; 0.13 G elements at batch 32).  Each of those costs 2-6 % of the logits' error (P5 6.98e-4 -> 8.1e-4 at 640 x 640, 7.6e-4 -> 8.3e-4 at 320 x 320, under the 9e-4 line
#   the round set); the backbone layers below 80 x 80 would add as much again for a tenth of the bytes and stay fp32, and so does everything at
#   the small shapes of the fixtures (128 x 128, 96 x 160), whose P5 statistics over 32-45 samples leave no margin.
# Layers with a residual input and the fp32 stem never take the flag.  None / 0 switch a rule off.
RAW_F16_FROM = 12
RAW_F16_MIN_PIXELS = 80 * 80


def build_yolov8_graph(lay: ParamLayout, H: int, W: int) -> Graph:
    """Buffer plan + op list for an (H, W) input (both multiples of 32)."""
    if H % 32 or W % 32:
        raise ValueError("input height/width must be multiples of 32")
    g = Graph()
    View = lambda b, off, c, pix=0: (b, off, c, pix)  # noqa: E731

    def buf(h, w, c, kind=L.BUF_ACT_F16):
        g.bufs.append((h, w, c, kind))
        return len(g.bufs) - 1

    def conv(name, vin, vout, hin, win, stride=1, res=None, needs_dgrad=1):
        s = lay.convs[name]
        k = s.k
        ho, wo = (hin + 2 * (k // 2) - k) // stride + 1, (win + 2 * (k // 2) - k) // stride + 1
        assert vout[2] == s.cout_eng, (name, vout, s.cout_eng)
        op = dict(type=L.OP_CONV, name=name, out=vout, ih=hin, iw=win, oh=ho, ow=wo, k=k, stride=stride, pad=k // 2, dil=1,
                  act=L.ACT_BN_SILU if s.bn else L.ACT_BIAS, needs_dgrad=needs_dgrad, w_cin=s.cin, w_off=s.w_off,
                  gamma_off=s.gamma_off, beta_off=s.beta_off, bias_off=s.bias_off, rmean_off=s.rmean_off, rvar_off=s.rvar_off)
        op["in"] = vin
        if res is not None:
            op["res"] = res
        if s.bn and res is None and name != "0" and ((RAW_F16_FROM is not None and int(name.split(".")[0]) >= RAW_F16_FROM) or
                                                     (RAW_F16_MIN_PIXELS and ".m" not in name and ho * wo >= RAW_F16_MIN_PIXELS)):
            op["flags"] = L.OPF_RAW_F16
        g.ops.append(op)
        return ho, wo

    def c2f(idx, vin, vout, h, w, c2, n, shortcut):
        """cv1 -> [a | b] in place, bottlenecks append slices, cv2 reads the whole buffer (modules.py:199-202)."""
        c = c2 // 2
        cat = buf(h, w, (2 + n) * c)
        conv(f"{idx}.cv1", vin, View(cat, 0, 2 * c), h, w)
        for j in range(n):
            tmp = buf(h, w, c)             # one per bottleneck: its content is an operand of the backward pass
            src = View(cat, (1 + j) * c, c)
            conv(f"{idx}.m{j}.cv1", src, View(tmp, 0, c), h, w)
            conv(f"{idx}.m{j}.cv2", View(tmp, 0, c), View(cat, (2 + j) * c, c), h, w, res=src if shortcut else None)
        conv(f"{idx}.cv2", View(cat, 0, (2 + n) * c), vout, h, w)

    c64, c128, c256, c512, c1024 = lay.c64, lay.c128, lay.c256, lay.c512, lay.c1024
    n3, n6 = lay.n3, lay.n6
    h2, w2, h4, w4, h8, w8, h16, w16, h32, w32 = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8, H // 16, W // 16, H // 32, W // 32

    img = buf(H, W, 8)
    g.image_buf = img
    b0 = buf(h2, w2, c64)
    b1 = buf(h4, w4, c128)
    b2 = buf(h4, w4, c128)
    b3 = buf(h8, w8, c256)
    cat14 = buf(h8, w8, c512 + c256)       # [up(L12) | L4]            yolo_v8.py:93-94
    b5 = buf(h16, w16, c512)
    cat11 = buf(h16, w16, c1024 + c512)    # [up(L9) | L6]             yolo_v8.py:91-92
    b7 = buf(h32, w32, c1024)
    b8 = buf(h32, w32, c1024)
    sppf = buf(h32, w32, (c1024 // 2) * 4)
    cat20 = buf(h32, w32, c512 + c1024)    # [L19 | L9]                yolo_v8.py:97-98
    cat17 = buf(h16, w16, c256 + c512)     # [L16 | L12]               yolo_v8.py:95-96
    b15 = buf(h8, w8, c256)
    b18 = buf(h16, w16, c512)
    b21 = buf(h32, w32, c1024)

    conv("0", View(img, 0, 8), View(b0, 0, c64), H, W, stride=2, needs_dgrad=0)
    conv("1", View(b0, 0, c64), View(b1, 0, c128), h2, w2, stride=2)
    c2f(2, View(b1, 0, c128), View(b2, 0, c128), h4, w4, c128, n3, True)
    conv("3", View(b2, 0, c128), View(b3, 0, c256), h4, w4, stride=2)
    L4 = View(cat14, c512, c256)
    c2f(4, View(b3, 0, c256), L4, h8, w8, c256, n6, True)
    conv("5", L4, View(b5, 0, c512), h8, w8, stride=2)
    L6 = View(cat11, c1024, c512)
    c2f(6, View(b5, 0, c512), L6, h16, w16, c512, n6, True)
    conv("7", L6, View(b7, 0, c1024), h16, w16, stride=2)
    c2f(8, View(b7, 0, c1024), View(b8, 0, c1024), h32, w32, c1024, n3, True)
    # SPPF (modules.py:314-318): cv1 -> slice 0, three chained 5x5 pools -> slices 1..3, cv2 reads all
    ch = c1024 // 2
    conv("9.cv1", View(b8, 0, c1024), View(sppf, 0, ch), h32, w32)
    for j in range(3):
        g.ops.append(dict(type=L.OP_MAXPOOL5, name=f"9.pool{j}", out=View(sppf, (j + 1) * ch, ch), ih=h32, iw=w32, oh=h32, ow=w32))
        g.ops[-1]["in"] = View(sppf, j * ch, ch)
    L9 = View(cat20, c512, c1024)
    conv("9.cv2", View(sppf, 0, 4 * ch), L9, h32, w32)
    g.ops.append(dict(type=L.OP_UPSAMPLE2, name="10", out=View(cat11, 0, c1024), ih=h32, iw=w32, oh=h16, ow=w16))
    g.ops[-1]["in"] = L9
    L12 = View(cat17, c256, c512)
    c2f(12, View(cat11, 0, c1024 + c512), L12, h16, w16, c512, n3, False)
    g.ops.append(dict(type=L.OP_UPSAMPLE2, name="13", out=View(cat14, 0, c512), ih=h16, iw=w16, oh=h8, ow=w8))
    g.ops[-1]["in"] = L12
    c2f(15, View(cat14, 0, c512 + c256), View(b15, 0, c256), h8, w8, c256, n3, False)
    conv("16", View(b15, 0, c256), View(cat17, 0, c256), h8, w8, stride=2)
    c2f(18, View(cat17, 0, c256 + c512), View(b18, 0, c512), h16, w16, c512, n3, False)
    conv("19", View(b18, 0, c512), View(cat20, 0, c512), h16, w16, stride=2)
    c2f(21, View(cat20, 0, c512 + c1024), View(b21, 0, c1024), h32, w32, c1024, n3, False)

    # Detect (modules.py:428-433): per level one fused 3x3 (box|cls), two 3x3, two 1x1+bias into pred
    g.level_hw = [(h8, w8), (h16, w16), (h32, w32)]
    g.anchors = sum(a * b for a, b in g.level_hw)
    pred = buf(g.anchors, 1, lay.no_pad, L.BUF_PRED_F32)
    g.pred_buf = pred
    a_off = 0
    cb, cc = lay.c_box, lay.c_cls_eng
    for lvl, (src, (hh, ww)) in enumerate(zip((View(b15, 0, c256), View(b18, 0, c512), View(b21, 0, c1024)), g.level_hw)):
        h1 = buf(hh, ww, cb + cc)
        hb = buf(hh, ww, cb)
        hc = buf(hh, ww, cc)
        first = len(g.ops)
        conv(f"22.{lvl}.0", src, View(h1, 0, cb + cc), hh, ww)
        conv(f"22.{lvl}.1b", View(h1, 0, cb), View(hb, 0, cb), hh, ww)
        conv(f"22.{lvl}.1c", View(h1, cb, cc), View(hc, 0, cc), hh, ww)
        conv(f"22.{lvl}.2b", View(hb, 0, cb), View(pred, 0, 4 * REG_MAX, a_off), hh, ww)
        conv(f"22.{lvl}.2c", View(hc, 0, cc), View(pred, 4 * REG_MAX, lay.nc_pad, a_off), hh, ww)
        for op in g.ops[first:]:
            op["lane"] = 1 + lvl          # the three Detect levels are independent: own HIP stream each (cvx_op_desc.lane)
        a_off += hh * ww
    g.taps = {0: (b0, 0, c64), 1: (b1, 0, c128), 2: (b2, 0, c128), 3: (b3, 0, c256), 4: L4[:3], 5: (b5, 0, c512), 6: L6[:3],
              7: (b7, 0, c1024), 8: (b8, 0, c1024), 9: L9[:3], 12: L12[:3], 15: (b15, 0, c256), 18: (b18, 0, c512),
              21: (b21, 0, c1024)}
    return g


def grad_buckets(g: Graph, lay: ParamLayout, n_buckets: int, tail_modules: int = 2):
    """Op ranges, in backward order, whose parameters are contiguous slices of the flat arenas:
    ``[(op_hi, op_lo, p_start, p_end), ...]`` -- the units of the overlapped gradient exchange
    (``cvx_engine_backward_range`` / ``cvx_engine_grads_ready``).  Cuts fall between top-level modules (model.N),
    balanced by parameter count; the ranges tile the op list from the last op down to op 0 and the slices tile the arena.
    The first ``tail_modules`` modules (the stem: tiny parameters, but the largest images, i.e. the longest weight
    gradients, and the last to finish) form a bucket of their own when there is more than one bucket, so that the
    exchange left after the end of the backward pass is a few kilobytes."""
    mods = []                                          # [op_lo, op_hi, p_start, p_end] per top-level module, op order
    last = None
    for i, o in enumerate(g.ops):
        m = o["name"].split(".")[0]
        if m != last:
            mods.append([i, i, None, None])
            last = m
        mods[-1][1] = i
        if o["type"] == L.OP_CONV:
            spec = lay.convs[o["name"]]
            start = spec.w_off
            end = ((spec.beta_off if spec.bn else spec.bias_off) + spec.cout_eng + 3) & ~3
            mods[-1][2] = start if mods[-1][2] is None else min(mods[-1][2], start)
            mods[-1][3] = end if mods[-1][3] is None else max(mods[-1][3], end)
    merged = []
    for m in mods:                                     # parameter-free modules (upsample) ride with their predecessor
        if m[2] is None and merged:
            merged[-1][1] = m[1]
        else:
            merged.append(m)
    if merged[0][2] is None:
        merged[1][0] = merged[0][0]
        merged = merged[1:]
    tail = []
    if int(n_buckets) > 1 and 0 < tail_modules < len(merged) - 1:
        tail, merged = merged[:tail_modules], merged[tail_modules:]
        n_buckets = int(n_buckets) - 1
    total = sum(m[3] - m[2] for m in merged)
    n = max(1, min(int(n_buckets), len(merged)))
    groups, cur, acc = [], [], 0
    for k in range(len(merged) - 1, -1, -1):           # backward order: the head first
        cur.append(merged[k])
        acc += merged[k][3] - merged[k][2]
        groups_left = n - len(groups) - 1              # groups still to be opened after the current one
        if groups_left > 0 and (acc >= total / n or k == groups_left):
            groups.append(cur)
            cur, acc = [], 0
    if cur:
        groups.append(cur)
    if tail:
        groups.append(tail)
    out = [(max(m[1] for m in b), min(m[0] for m in b), min(m[2] for m in b), max(m[3] for m in b)) for b in groups]
    assert out[0][0] == len(g.ops) - 1 and out[-1][1] == 0 and out[-1][2] == 0
    for a, b in zip(out, out[1:]):
        assert b[0] == a[1] - 1 and b[3] == a[2], (a, b)  # ops and parameters both tile without gaps
    return out


def op_param_interval(o):
    """[start, end) of the flat parameter arena an op's trainable tensors occupy (weights, BatchNorm affine, bias), or None."""
    t = o["type"]
    iv = []
    if t == L.OP_CONV:
        cout = o["out"][2]
        iv.append((o["w_off"], o["w_off"] + cout * o["k"] * o["k"] * o["w_cin"]))
        act = o.get("act", 0)
        if act in (L.ACT_BN_SILU, L.ACT_BN_RELU, L.ACT_BN_LINEAR):
            iv += [(o["gamma_off"], o["gamma_off"] + cout), (o["beta_off"], o["beta_off"] + cout)]
        if act in (L.ACT_BIAS, L.ACT_BIAS_RELU, L.ACT_BIAS_LINEAR) or (o.get("flags", 0) & L.OPF_CONV_BIAS):
            iv.append((o["bias_off"], o["bias_off"] + cout))
    elif t == L.OP_DWCONVT:
        f = o["stride"]
        iv.append((o["w_off"], o["w_off"] + o["in"][2] * 4 * f * f))
    elif t == L.OP_L2NORM:
        iv.append((o["gamma_off"], o["gamma_off"] + o["in"][2]))
    if not iv:
        return None
    return min(a for a, _ in iv), max(b for _, b in iv)


def generic_grad_buckets(g: Graph, n_buckets: int):
    """``grad_buckets`` for any graph, from the ops' own parameter offsets: op ranges, in backward order, whose parameters form disjoint,
    ordered slices of the flat arena -- ``[(op_hi, op_lo, p_start, p_end), ...]``, the units of the overlapped gradient exchange.  A cut
    between op k and k + 1 is allowed where every parameter of ops <= k lies below every parameter of ops > k (layouts follow the
    reference's state_dict order, which is not always op order: a Bottleneck's downsample runs before conv3 but is stored after it);
    the atomic units between allowed cuts are grouped from the head backwards into ``n_buckets`` of similar parameter count.
    Parameters no op touches (the reference's unused modules) have zero gradients on every rank and need no exchange."""
    nops = len(g.ops)
    iv = [op_param_interval(o) for o in g.ops]
    INF = 1 << 62
    pre_end, suf_start = [0] * nops, [INF] * (nops + 1)
    run = 0
    for k in range(nops):
        if iv[k] is not None:
            run = max(run, iv[k][1])
        pre_end[k] = run
    for k in range(nops - 1, -1, -1):
        suf_start[k] = min(suf_start[k + 1], iv[k][0] if iv[k] is not None else INF)
    units, lo = [], 0                                    # atomic units [op_lo, op_hi, p_start, p_end] in op order
    for k in range(nops):
        if k == nops - 1 or pre_end[k] <= suf_start[k + 1]:
            ivs = [iv[q] for q in range(lo, k + 1) if iv[q] is not None]
            if ivs:
                units.append([lo, k, min(a for a, _ in ivs), max(b for _, b in ivs)])
            elif units:
                units[-1][1] = k                         # parameter-free ops ride with their predecessor
            else:
                units.append([lo, k, 0, 0])
            lo = k + 1
    units[0][0] = 0
    total = sum(u[3] - u[2] for u in units)
    n = max(1, min(int(n_buckets), len(units)))
    groups, cur, acc = [], [], 0
    for idx in range(len(units) - 1, -1, -1):            # backward order: the head first
        cur.append(units[idx])
        acc += units[idx][3] - units[idx][2]
        left = n - len(groups) - 1
        if left > 0 and (acc >= total / n or idx == left):
            groups.append(cur)
            cur, acc = [], 0
    if cur:
        groups.append(cur)
    out = [(max(u[1] for u in b), min(u[0] for u in b), min(u[2] for u in b), max(u[3] for u in b)) for b in groups]
    assert out[0][0] == nops - 1 and out[-1][1] == 0
    for a, b in zip(out, out[1:]):
        assert b[0] == a[1] - 1 and b[3] <= a[2], (a, b)  # ops tile without gaps; parameter slices are disjoint and ordered
    return out
