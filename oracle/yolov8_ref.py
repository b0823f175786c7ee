"""CPU oracle for the YOLOv8 hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A torch-CPU fp32 restatement of the reference's algorithm for SURVEY.md section 8 rows a1-a15.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product path (``computervision.pytorch_amd``) never does and fails loudly without its
HIP library.

Parity pin: ``oracle/make_golden.py`` imports the real reference from ``/root/reference`` in the
build container, asserts this restatement agrees with it (weights bit-exact, forward / loss /
gradients / Adam to fp32 round-off) and writes the fixtures under ``tests/golden/`` that
``tests/test_oracle_golden.py`` re-checks wherever the reference is absent (the GPU box).

The network is written functionally over a flat ``state_dict`` whose keys and shapes are the
reference's (``model.<i>....``, 355 tensors for scale "n"), so reference checkpoints load and
autograd gives per-parameter gradients.  Citations are ``file:line`` under ``/root/reference``.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

# (depth multiple, width multiple, max channels)        core/models/yolov8/yolo_v8.py:110-132
SCALES = {
    "n": (0.33, 0.25, 1024),
    "s": (0.33, 0.50, 1024),
    "m": (0.67, 0.75, 768),
    "l": (1.00, 1.00, 512),
    "x": (1.00, 1.25, 512),
}
BN_EPS = 1e-3          # core/models/yolov8/torch_utils.py:17-19
BN_MOMENTUM = 0.03
REG_MAX = 16           # core/models/yolov8/modules.py:419


# ----------------------------------------------------------------------------------------------
# architecture description
# ----------------------------------------------------------------------------------------------
def _round_channels(c: int, width: float, max_ch: int) -> int:
    """Width scaling, rounded up to a multiple of 8 (yolo_v8.py:67-76, ultralytics_ops.py:115-128)."""
    return int(math.ceil(min(c, max_ch) * width / 8) * 8)


def _repeats(n: int, depth: float) -> int:
    """Depth scaling (yolo_v8.py:64-65)."""
    return max(round(n * depth), 1) if n > 1 else n


def arch(model_type: str = "n", nc: int = 80) -> dict:
    """Channel / repeat plan of the 23-module graph (yolo_v8.py:26-49)."""
    depth, width, max_ch = SCALES[model_type]
    ch = lambda c: _round_channels(c, width, max_ch)  # noqa: E731
    c64, c128, c256, c512, c1024 = ch(64), ch(128), ch(256), ch(512), ch(1024)
    n3, n6 = _repeats(3, depth), _repeats(6, depth)
    head_in = (c256, c512, c1024)
    return dict(
        nc=nc,
        stem=[(3, c64), (c64, c128)],                       # layers 0, 1
        # (layer index, kind, args)
        layers=[
            (0, "conv", dict(c1=3, c2=c64, k=3, s=2)),
            (1, "conv", dict(c1=c64, c2=c128, k=3, s=2)),
            (2, "c2f", dict(c1=c128, c2=c128, n=n3, shortcut=True)),
            (3, "conv", dict(c1=c128, c2=c256, k=3, s=2)),
            (4, "c2f", dict(c1=c256, c2=c256, n=n6, shortcut=True)),
            (5, "conv", dict(c1=c256, c2=c512, k=3, s=2)),
            (6, "c2f", dict(c1=c512, c2=c512, n=n6, shortcut=True)),
            (7, "conv", dict(c1=c512, c2=c1024, k=3, s=2)),
            (8, "c2f", dict(c1=c1024, c2=c1024, n=n3, shortcut=True)),
            (9, "sppf", dict(c1=c1024, c2=c1024)),
            (10, "up", {}), (11, "cat", dict(src=6)),
            (12, "c2f", dict(c1=c1024 + c512, c2=c512, n=n3, shortcut=False)),
            (13, "up", {}), (14, "cat", dict(src=4)),
            (15, "c2f", dict(c1=c512 + c256, c2=c256, n=n3, shortcut=False)),
            (16, "conv", dict(c1=c256, c2=c256, k=3, s=2)), (17, "cat", dict(src=12)),
            (18, "c2f", dict(c1=c256 + c512, c2=c512, n=n3, shortcut=False)),
            (19, "conv", dict(c1=c512, c2=c512, k=3, s=2)), (20, "cat", dict(src=9)),
            (21, "c2f", dict(c1=c512 + c1024, c2=c1024, n=n3, shortcut=False)),
            (22, "detect", dict(ch=head_in)),
        ],
        head_in=head_in,
        # Detect hidden widths (modules.py:422)
        c_box=max(16, head_in[0] // 4, REG_MAX * 4),
        c_cls=max(head_in[0], nc),
        strides=(8.0, 16.0, 32.0),
    )


def _conv_unit_entries(prefix: str, c1: int, c2: int, k: int):
    """Conv2d(no bias)+BN parameter/buffer shapes in registration order (modules.py:23-27)."""
    return [
        (prefix + ".conv.weight", (c2, c1, k, k), "conv"),
        (prefix + ".bn.weight", (c2,), "ones"),
        (prefix + ".bn.bias", (c2,), "zeros"),
        (prefix + ".bn.running_mean", (c2,), "zeros"),
        (prefix + ".bn.running_var", (c2,), "rvar"),
        (prefix + ".bn.num_batches_tracked", (), "nbt"),
    ]


def param_plan(model_type: str = "n", nc: int = 80):
    """[(key, shape, kind)] in the reference's construction == state_dict order."""
    a = arch(model_type, nc)
    plan = []
    for idx, kind, kw in a["layers"]:
        p = f"model.{idx}"
        if kind == "conv":
            plan += _conv_unit_entries(p, kw["c1"], kw["c2"], kw["k"])
        elif kind == "c2f":                                 # modules.py:192-197
            c = kw["c2"] // 2
            plan += _conv_unit_entries(p + ".cv1", kw["c1"], 2 * c, 1)
            plan += _conv_unit_entries(p + ".cv2", (2 + kw["n"]) * c, kw["c2"], 1)
            for j in range(kw["n"]):
                plan += _conv_unit_entries(f"{p}.m.{j}.cv1", c, c, 3)
                plan += _conv_unit_entries(f"{p}.m.{j}.cv2", c, c, 3)
        elif kind == "sppf":                                # modules.py:307-312
            c_ = kw["c1"] // 2
            plan += _conv_unit_entries(p + ".cv1", kw["c1"], c_, 1)
            plan += _conv_unit_entries(p + ".cv2", c_ * 4, kw["c2"], 1)
        elif kind == "detect":                              # modules.py:415-426
            for branch, width, cout in (("cv2", a["c_box"], 4 * REG_MAX), ("cv3", a["c_cls"], nc)):
                for lvl, cin in enumerate(kw["ch"]):
                    q = f"{p}.{branch}.{lvl}"
                    plan += _conv_unit_entries(q + ".0", cin, width, 3)
                    plan += _conv_unit_entries(q + ".1", width, width, 3)
                    plan.append((q + ".2.weight", (cout, width, 1, 1), "conv"))
                    plan.append((q + ".2.bias", (cout,), "convbias:" + str(width) + ":" + branch + ":" + str(lvl)))
            plan.append((p + ".dfl.conv.weight", (1, REG_MAX, 1, 1), "dfl"))
    return plan


def init_state_dict(model_type: str = "n", nc: int = 80, seed: int | None = 0) -> "OrderedDict[str, torch.Tensor]":
    """Random-init weights exactly as the reference constructor leaves them.

    torch's default Conv2d init (kaiming-uniform, a=sqrt(5); bias U(+-1/sqrt(fan_in))) drawn in the
    reference's module construction order from the global RNG, then the constructor's side effects:
    the stride-probe forward on zeros (yolo_v8.py:53-58) leaves every BN with running_mean 0,
    running_var 0.9, num_batches_tracked 1, and ``bias_init`` (modules.py:448-455) overwrites the
    head biases.  ``make_golden.py`` asserts bit-equality with ``get_yolo8_n`` under the same seed.
    """
    if seed is not None:
        torch.manual_seed(seed)
    a = arch(model_type, nc)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for key, shape, kind in param_plan(model_type, nc):
        if kind == "conv" or kind == "dfl":
            w = torch.empty(shape)
            torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
            if kind == "dfl":                               # modules.py:75-78: weights 0..15, frozen
                w = torch.arange(REG_MAX, dtype=torch.float32).view(1, REG_MAX, 1, 1)
            sd[key] = w
        elif kind.startswith("convbias"):
            _, fan_in, branch, lvl = kind.split(":")
            b = torch.empty(shape)
            bound = 1.0 / math.sqrt(int(fan_in))
            torch.nn.init.uniform_(b, -bound, bound)
            if branch == "cv2":
                b.fill_(1.0)
            else:
                b[:nc] = math.log(5 / nc / (640 / a["strides"][int(lvl)]) ** 2)
            sd[key] = b
        elif kind == "ones":
            sd[key] = torch.ones(shape)
        elif kind == "zeros":
            sd[key] = torch.zeros(shape)
        elif kind == "rvar":
            sd[key] = torch.full(shape, 0.9)
        elif kind == "nbt":
            sd[key] = torch.tensor(1, dtype=torch.long)
    return sd


def trainable_keys(sd: Dict[str, torch.Tensor]) -> List[str]:
    """Keys Adam updates: everything except BN buffers and the frozen DFL projection."""
    out = []
    for k in sd:
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")) or k.endswith("dfl.conv.weight"):
            continue
        out.append(k)
    return out


# ----------------------------------------------------------------------------------------------
# forward graph
# ----------------------------------------------------------------------------------------------
# fp16-STORAGE emulation.  The reference's GPU path runs under torch.cuda.amp.autocast()
# (core/trainer/yolo8_train.py:98-104): conv operands and activations are fp16, accumulation fp32.
# With FP16_STORAGE[0] = True the oracle rounds exactly the tensors the MI355X engine rounds -- the weights of the MFMA
# convolutions and every stored activation -- and nothing else: the stem (model.0) runs in fp32 from the fp32 image, and
# the BatchNorm normalises the un-rounded fp32 conv output (the engine keeps it in fp32 until the normalisation pass).
# The plain fp32 run is the reference's CPU path.  Casts are differentiable (straight-through).
FP16_STORAGE = [False]


def _q(t: torch.Tensor) -> torch.Tensor:
    return t.half().float() if FP16_STORAGE[0] else t


# Layers (state_dict prefixes, e.g. "model.2.cv1") whose raw conv output the engine stores in fp16 between the convolution and its
# normalisation pass.  The batch statistics still come from the fp32 accumulators (the conv epilogue sums them before it rounds); only the
# normalised VALUE starts from the rounded number.  None: the engine's own rule (graph.py: RAW_F16_FROM, RAW_F16_MIN_PIXELS) -- every Conv of
# the neck and the head (top-level modules 12 .. 22), plus the backbone's Convs outside the Bottlenecks with outputs of at least 80 x 80 per
# image; never the stem.  A set overrides the rule (oracle/fp16_raw_study.py).
FP16_RAW_LAYERS = None
ENGINE_RAW_F16_FROM = 12
ENGINE_RAW_F16_MIN_PIXELS = 80 * 80


def _raw_f16(p: str, pixels: int) -> bool:
    if FP16_RAW_LAYERS is not None:
        return p in FP16_RAW_LAYERS
    part = p.split(".")
    if int(part[1]) >= ENGINE_RAW_F16_FROM:
        return True
    return p != "model.0" and "m" not in part[2:] and pixels >= ENGINE_RAW_F16_MIN_PIXELS   # (not the Bottlenecks' convs)


def _unit(x, sd, p, k, s, training):
    """Conv2d(no bias, 'same' pad) -> BatchNorm -> SiLU   (modules.py:29-30)."""
    w = sd[p + ".conv.weight"]
    y = F.conv2d(x, w if p == "model.0" else _q(w), None, s, k // 2)
    if FP16_STORAGE[0] and training and _raw_f16(p, y.shape[2] * y.shape[3]):
        mean, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=False)
        n = y.numel() // y.shape[1]
        with torch.no_grad():  # running statistics exactly as F.batch_norm updates them
            sd[p + ".bn.running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean)
            sd[p + ".bn.running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var * (n / max(n - 1, 1)))
            sd[p + ".bn.num_batches_tracked"] += 1
        yq = _q(y)
        y = (yq - mean.view(1, -1, 1, 1)) * torch.rsqrt(var.view(1, -1, 1, 1) + BN_EPS) * sd[p + ".bn.weight"].view(1, -1, 1, 1) + sd[p + ".bn.bias"].view(1, -1, 1, 1)
        return _q(F.silu(y))
    y = F.batch_norm(y, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"],
                     sd[p + ".bn.bias"], training, BN_MOMENTUM, BN_EPS)
    if training:
        sd[p + ".bn.num_batches_tracked"] += 1
    return _q(F.silu(y))


def _c2f(x, sd, p, n, shortcut, training):
    """modules.py:199-202 -- split, chain of bottlenecks on the last chunk, concat, 1x1."""
    parts = list(_unit(x, sd, p + ".cv1", 1, 1, training).chunk(2, 1))
    for j in range(n):
        h = _unit(parts[-1], sd, f"{p}.m.{j}.cv1", 3, 1, training)
        h = _unit(h, sd, f"{p}.m.{j}.cv2", 3, 1, training)
        parts.append(_q(parts[-1] + h) if shortcut else h)  # modules.py:134-135
    return _unit(torch.cat(parts, 1), sd, p + ".cv2", 1, 1, training)


def _sppf(x, sd, p, training):
    """modules.py:314-318."""
    x = _unit(x, sd, p + ".cv1", 1, 1, training)
    y1 = F.max_pool2d(x, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    y3 = F.max_pool2d(y2, 5, 1, 2)
    return _unit(torch.cat((x, y1, y2, y3), 1), sd, p + ".cv2", 1, 1, training)


def make_anchors(shapes: List[Tuple[int, int]], strides, offset: float = 0.5):
    """Anchor centres (A,2) in grid units and per-anchor stride (A,1)   (anchor.py:126-145)."""
    pts, st = [], []
    for (h, w), s in zip(shapes, strides):
        sy, sx = torch.meshgrid(torch.arange(h, dtype=torch.float32) + offset,
                                torch.arange(w, dtype=torch.float32) + offset, indexing="ij")
        pts.append(torch.stack((sx, sy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s)))
    return torch.cat(pts), torch.cat(st)


def decode_eval(feats: List[torch.Tensor], strides, nc: int) -> torch.Tensor:
    """Detect eval tail: DFL expectation -> cxcywh * stride, sigmoid(cls)   (modules.py:434-446)."""
    b = feats[0].shape[0]
    x_cat = torch.cat([f.reshape(b, 4 * REG_MAX + nc, -1) for f in feats], 2)
    box, cls = x_cat.split((4 * REG_MAX, nc), 1)
    anchors, st = make_anchors([f.shape[2:] for f in feats], strides)
    prob = box.view(b, 4, REG_MAX, -1).softmax(2)
    dist = (prob * torch.arange(REG_MAX, dtype=torch.float32).view(1, 1, -1, 1)).sum(2)   # (b,4,A)
    lt, rb = dist[:, :2], dist[:, 2:]
    a = anchors.t().unsqueeze(0)
    x1y1, x2y2 = a - lt, a + rb
    dbox = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * st.t()
    return torch.cat((dbox, cls.sigmoid()), 1)


def forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, model_type: str = "n", nc: int = 80,
            training: bool = True, taps: dict | None = None):
    """Yolo8.forward (yolo_v8.py:78-107).  train -> [3 x (B,144,H,W)]; eval -> (y (B,84,A), same list).

    In training mode the BN running statistics inside ``sd`` are updated in place, as torch does.
    ``taps`` (optional dict) receives the output of every top-level module, keyed by layer index.
    """
    a = arch(model_type, nc)
    saved = {}
    for idx, kind, kw in a["layers"]:
        p = f"model.{idx}"
        if kind == "conv":
            x = _unit(x, sd, p, kw["k"], kw["s"], training)
        elif kind == "c2f":
            x = _c2f(x, sd, p, kw["n"], kw["shortcut"], training)
        elif kind == "sppf":
            x = _sppf(x, sd, p, training)
        elif kind == "up":
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        elif kind == "cat":
            x = torch.cat((x, saved[kw["src"]]), 1)
        elif kind == "detect":
            outs = []
            for lvl, src in enumerate((15, 18, 21)):
                f = saved[src]
                branch_out = []
                for br in ("cv2", "cv3"):
                    q = f"{p}.{br}.{lvl}"
                    h = _unit(f, sd, q + ".0", 3, 1, training)
                    h = _unit(h, sd, q + ".1", 3, 1, training)
                    branch_out.append(F.conv2d(h, _q(sd[q + ".2.weight"]), sd[q + ".2.bias"]))
                outs.append(torch.cat(branch_out, 1))
            x = outs
        saved[idx] = x
        if taps is not None:
            taps[idx] = x
    if training:
        return x
    return decode_eval(x, a["strides"], nc), x


# ----------------------------------------------------------------------------------------------
# loss (core/algorithms/yolo_v8.py:25-124) and task-aligned assigner (core/utils/bboxes.py:231-470)
# ----------------------------------------------------------------------------------------------
def ciou(box1: torch.Tensor, box2: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """Complete-IoU of xyxy boxes, broadcasting over leading dims (ultralytics_iou.py:64-117).

    Quirks kept: eps is added to the heights only, and ``alpha`` carries no gradient.
    """
    ax1, ay1, ax2, ay2 = box1.unbind(-1)
    bx1, by1, bx2, by2 = box2.unbind(-1)
    w1, h1 = ax2 - ax1, ay2 - ay1 + eps
    w2, h2 = bx2 - bx1, by2 - by1 + eps
    inter = (torch.minimum(ax2, bx2) - torch.maximum(ax1, bx1)).clamp(0) * \
            (torch.minimum(ay2, by2) - torch.maximum(ay1, by1)).clamp(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.maximum(ax2, bx2) - torch.minimum(ax1, bx1)
    chh = torch.maximum(ay2, by2) - torch.minimum(ay1, by1)
    c2 = cw ** 2 + chh ** 2 + eps
    rho2 = ((bx1 + bx2 - ax1 - ax2) ** 2 + (by1 + by2 - ay1 - ay2) ** 2) / 4
    v = (4 / math.pi ** 2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)).pow(2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def build_targets(batch: dict, batch_size: int, img_h: float, img_w: float):
    """Loss.preprocess (yolo_v8.py:51-65): flat (N,6) labels -> (B,Gmax,5) [cls, xyxy pixels]."""
    bi = batch["batch_idx"].view(-1).float()
    cls = batch["cls"].view(-1).float()
    boxes = batch["bboxes"].view(-1, 4).float()
    if bi.numel() == 0:
        return torch.zeros(batch_size, 0, 5)
    counts = [(bi == j).sum().item() for j in range(batch_size)]
    gmax = int(max(counts))
    out = torch.zeros(batch_size, gmax, 5)
    for j in range(batch_size):
        sel = bi == j
        n = int(sel.sum())
        if n:
            out[j, :n, 0] = cls[sel]
            out[j, :n, 1:] = boxes[sel]
    scale = torch.tensor([img_w, img_h, img_w, img_h])
    cxcywh = out[..., 1:5] * scale
    xy, wh = cxcywh[..., :2], cxcywh[..., 2:]
    out[..., 1:5] = torch.cat((xy - wh / 2, xy + wh / 2), -1)
    return out


@torch.no_grad()
def task_aligned_assign(pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt,
                        topk: int = 10, alpha: float = 0.5, beta: float = 6.0, eps: float = 1e-9):
    """TaskAlignedAssigner.forward (bboxes.py:299-345).

    pd_scores (B,A,nc) sigmoid scores, pd_bboxes (B,A,4) xyxy px, anc_points (A,2) px,
    gt_labels (B,G,1), gt_bboxes (B,G,4) xyxy px, mask_gt (B,G,1) in {0,1}.
    Returns target_bboxes (B,A,4), target_scores (B,A,nc), fg_mask (B,A) bool, target_gt_idx (B,A).
    """
    B, A, nc = pd_scores.shape
    G = gt_bboxes.shape[1]
    if G == 0:                                              # bboxes.py:322-327
        return (torch.zeros_like(pd_bboxes), torch.zeros_like(pd_scores),
                torch.zeros(B, A, dtype=torch.bool), torch.zeros(B, A, dtype=torch.long))
    # anchors strictly inside a gt (bboxes.py:231-246)
    lt = anc_points.view(1, 1, A, 2) - gt_bboxes[:, :, None, :2]
    rb = gt_bboxes[:, :, None, 2:] - anc_points.view(1, 1, A, 2)
    in_gt = (torch.cat((lt, rb), -1).amin(-1) > eps).float()            # (B,G,A)
    valid = (in_gt * mask_gt).bool()                                    # (B,G,A)
    # metric = score^alpha * CIoU^beta on the valid (gt, anchor) pairs (bboxes.py:369-396)
    lab = gt_labels.long().squeeze(-1).clamp(0, nc - 1)                 # (B,G)
    cls_score = pd_scores.gather(2, lab[:, None, :].expand(B, A, G)).permute(0, 2, 1)   # (B,G,A)
    iou = ciou(gt_bboxes[:, :, None, :], pd_bboxes[:, None, :, :]).clamp(0)             # (B,G,A)
    overlaps = torch.where(valid, iou, torch.zeros_like(iou))
    scores = torch.where(valid, cls_score, torch.zeros_like(cls_score))
    metric = scores.pow(alpha) * overlaps.pow(beta)
    # top-k anchors per gt; an index hit more than once (padding trick) is dropped (bboxes.py:398-429)
    _, top_idx = torch.topk(metric, topk, dim=-1)
    top_idx = torch.where(mask_gt.bool().expand(B, G, topk), top_idx, torch.zeros_like(top_idx))
    hits = torch.zeros(B, G, A, dtype=torch.long).scatter_add_(2, top_idx, torch.ones_like(top_idx))
    in_topk = torch.where(hits > 1, torch.zeros_like(hits), hits).float()
    mask_pos = in_topk * in_gt * mask_gt
    # an anchor claimed by several gts keeps the one with the largest overlap (bboxes.py:249-272)
    fg = mask_pos.sum(1)
    if fg.max() > 1:
        multi = (fg.unsqueeze(1) > 1).expand(B, G, A)
        best = F.one_hot(overlaps.argmax(1), G).permute(0, 2, 1).float()
        mask_pos = torch.where(multi, best, mask_pos)
        fg = mask_pos.sum(1)
    gt_idx = mask_pos.argmax(1)                                         # (B,A)
    # gather targets (bboxes.py:431-470)
    flat_idx = gt_idx + torch.arange(B).view(B, 1) * G
    t_labels = gt_labels.long().flatten()[flat_idx]
    t_boxes = gt_bboxes.reshape(-1, 4)[flat_idx]
    t_scores = F.one_hot(t_labels.clamp(0), nc).float() * (fg > 0).unsqueeze(-1)
    # rescale by the normalised alignment metric (bboxes.py:338-343)
    metric = metric * mask_pos
    pos_metric = metric.amax(-1, keepdim=True)
    pos_overlap = (overlaps * mask_pos).amax(-1, keepdim=True)
    norm = (metric * pos_overlap / (pos_metric + eps)).amax(1).unsqueeze(-1)
    return t_boxes, t_scores * norm, fg > 0, gt_idx


def v8_loss(feats: List[torch.Tensor], batch: dict, nc: int = 80, strides=(8.0, 16.0, 32.0),
            gains=(7.5, 0.5, 1.5), aux: dict | None = None):
    """Loss.__call__ (yolo_v8.py:75-124): returns (sum(box,cls,dfl) * B, detached (3,) items)."""
    B = feats[0].shape[0]
    no = nc + 4 * REG_MAX
    cat = torch.cat([f.reshape(B, no, -1) for f in feats], 2)
    pred_dist, pred_scores = cat.split((4 * REG_MAX, nc), 1)
    pred_scores = pred_scores.permute(0, 2, 1).contiguous()            # (B,A,nc)
    pred_dist = pred_dist.permute(0, 2, 1).contiguous()                # (B,A,64)
    img_h, img_w = feats[0].shape[2] * strides[0], feats[0].shape[3] * strides[0]
    anchors, stride_t = make_anchors([f.shape[2:] for f in feats], strides)

    targets = build_targets(batch, B, img_h, img_w)
    gt_labels, gt_bboxes = targets.split((1, 4), 2)
    mask_gt = (gt_bboxes.sum(2, keepdim=True) > 0).float()

    # expectation over the 16 DFL bins -> ltrb -> xyxy in grid units (yolo_v8.py:67-73)
    A = pred_dist.shape[1]
    ltrb = pred_dist.view(B, A, 4, REG_MAX).softmax(3).matmul(torch.arange(REG_MAX, dtype=torch.float32))
    pred_boxes = torch.cat((anchors - ltrb[..., :2], anchors + ltrb[..., 2:]), -1)

    t_boxes, t_scores, fg, gt_idx = task_aligned_assign(
        pred_scores.detach().sigmoid(), pred_boxes.detach() * stride_t, anchors * stride_t,
        gt_labels, gt_bboxes, mask_gt)
    t_boxes = t_boxes / stride_t
    score_sum = max(t_scores.sum(), 1)

    loss = torch.zeros(3)
    loss_cls = F.binary_cross_entropy_with_logits(pred_scores, t_scores, reduction="none").sum() / score_sum
    loss_box = torch.zeros(())
    loss_dfl = torch.zeros(())
    if fg.sum():
        w = t_scores.sum(-1)[fg].unsqueeze(-1)
        iou = ciou(pred_boxes[fg], t_boxes[fg]).unsqueeze(-1)
        loss_box = ((1.0 - iou) * w).sum() / score_sum                 # ultralytics_loss.py:34-36
        # DFL: two-bin cross entropy around the continuous target (ultralytics_loss.py:39-57)
        t_ltrb = torch.cat((anchors - t_boxes[..., :2], t_boxes[..., 2:] - anchors), -1).clamp(0, REG_MAX - 1 - 0.01)
        logits = pred_dist[fg].view(-1, REG_MAX)
        t = t_ltrb[fg]
        lo = t.long()
        w_lo = (lo + 1) - t
        ce_lo = F.cross_entropy(logits, lo.view(-1), reduction="none").view(lo.shape)
        ce_hi = F.cross_entropy(logits, (lo + 1).view(-1), reduction="none").view(lo.shape)
        dfl = (ce_lo * w_lo + ce_hi * (1 - w_lo)).mean(-1, keepdim=True)
        loss_dfl = (dfl * w).sum() / score_sum
    items = torch.stack((loss_box * gains[0], loss_cls * gains[1], loss_dfl * gains[2]))
    if aux is not None:
        aux.update(target_scores=t_scores, target_bboxes=t_boxes, fg_mask=fg, target_gt_idx=gt_idx,
                   score_sum=float(score_sum), pred_boxes=pred_boxes.detach())
    return items.sum() * B, items.detach()


# ----------------------------------------------------------------------------------------------
# optimiser step (torch.optim.Adam defaults; core/trainer/lr_scheduler.py:37-43)
# ----------------------------------------------------------------------------------------------
def adam_step(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], state: dict, lr: float = 1e-3,
              betas=(0.9, 0.999), eps: float = 1e-8):
    """In-place Adam over every key of ``grads``; ``state`` holds step/exp_avg/exp_avg_sq."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    bc1, bc2 = 1 - betas[0] ** t, 1 - betas[1] ** t
    for k, g in grads.items():
        m = state.setdefault("m:" + k, torch.zeros_like(g))
        v = state.setdefault("v:" + k, torch.zeros_like(g))
        m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
        v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        params[k].addcdiv_(m, denom, value=-lr / bc1)


def train_step(sd, images, batch, state, model_type="n", nc=80, lr=1e-3):
    """zero_grad -> forward -> loss -> backward -> Adam over all trainable tensors
    (core/trainer/yolo8_train.py:93-111, non-AMP branch: the CPU reference computes in fp32)."""
    keys = trainable_keys(sd)
    leaves = {k: sd[k].detach().requires_grad_(True) for k in keys}
    work = dict(sd)
    work.update(leaves)
    feats = forward(work, images, model_type, nc, training=True)
    loss, items = v8_loss(feats, batch, nc)
    grads = torch.autograd.grad(loss, [leaves[k] for k in keys], allow_unused=True)
    gd = {k: (g if g is not None else torch.zeros_like(sd[k])) for k, g in zip(keys, grads)}
    for k in sd:                        # BN buffers were updated in `work` (same tensor objects)
        if k not in leaves:
            sd[k] = work[k]
    with torch.no_grad():
        adam_step(sd, gd, state, lr)
    return loss.detach(), items, gd, feats
