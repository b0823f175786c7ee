"""CPU oracle for CenterNet (DLA-34): network forward (eval / train mode) and backward, heat-map decode -- TEST INFRASTRUCTURE,
NOT PRODUCT CODE.

SURVEY.md section 8 row a18 / (f)1, BASELINE.json configs[3].  A torch-CPU fp32 restatement, functional over a flat
``state_dict`` with the reference's keys (``backbone.base...``, ``backbone.dla_up...``, ``backbone.<head>...``), of

* the DLA-34 network (core/models/centernet_model.py:9-379): 7x7 stem, BasicBlock trees with Root aggregation, 2x2 max-pool
  down-sampling, IDAUp / DLAUp with depthwise ConvTranspose2d up-sampling, three heads, NHWC output (:371-379);
* the decode tail (core/algorithms/centernet.py:271-338): sigmoid, the 3x3 max-pool that the reference applies to the
  NHWC tensor as if it were NCHW -- the window spans (x, class) for a fixed row y, reproduced as is --, global top-K,
  gather, clamp, score mask, class-agnostic DIoU-NMS (core/utils/nms.py:9-31, core/utils/iou.py:8-64), letterbox
  inverse (core/utils/image_process.py:100-129).

Training (``forward(training=True)``, ``projection_loss``, ``loss_and_grads``): batch-statistics BatchNorm (running statistics
updated in place); the backward pass is torch autograd over this restatement, driven by a fixed linear functional of the output
tensor (the reference's CombinedLoss, core/loss/centernet_loss.py, is torch code on that tensor and is not restated); pinned
against the real model's autograd in ``make_golden.py`` (section 9b).

Parity pin: ``oracle/make_golden.py`` imports the real reference in the build container and asserts that this file
reproduces its seed-0 initialisation bit for bit, its forward to fp32 round-off and its decode exactly, then writes
``tests/golden/centernet_*.npz``.  ``torch.topk``'s order among equal scores is implementation-defined; the oracle defines
it as (score descending, flat index ascending) -- the fixtures contain no ties among the kept scores.
"""
from __future__ import annotations

import math

import numpy as np
from collections import OrderedDict

import torch
import torch.nn.functional as F

LEVELS = (1, 1, 1, 2, 2, 1)                      # dla34 (centernet_model.py:342-344)
CHANNELS = (16, 32, 64, 128, 256, 512)
HEAD_CONV = 256                                   # centernet_model.py:312
BN_EPS = 1e-5                                     # nn.BatchNorm2d defaults
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------------------------
# architecture walk: one description drives the initialiser, the forward and (in the product) the engine graph
# ----------------------------------------------------------------------------------------------
def tree_plan(prefix, levels, cin, cout, stride, level_root, root_dim=0):
    """Tree.__init__ (centernet_model.py:98-136) as data."""
    if root_dim == 0:
        root_dim = 2 * cout
    if level_root:
        root_dim += cin
    t = dict(prefix=prefix, levels=levels, cin=cin, cout=cout, stride=stride, level_root=level_root, root_dim=root_dim,
             project=cin != cout)
    if levels == 1:
        t["tree1"] = dict(kind="block", prefix=prefix + ".tree1", cin=cin, cout=cout, stride=stride)
        t["tree2"] = dict(kind="block", prefix=prefix + ".tree2", cin=cout, cout=cout, stride=1)
    else:
        t["tree1"] = tree_plan(prefix + ".tree1", levels - 1, cin, cout, stride, False, 0)
        t["tree2"] = tree_plan(prefix + ".tree2", levels - 1, cout, cout, 1, False, root_dim + cout)
    t["kind"] = "tree"
    return t


def dla_up_plan():
    """DLAUp.__init__ with its in-place list updates (centernet_model.py:282-296) -> [(out_dim, in_channels, up_factors)]."""
    channels = list(CHANNELS[2:])
    in_ch = list(CHANNELS[2:])
    scales = [1, 2, 4, 8]
    idas = []
    for i in range(len(channels) - 1):
        j = -i - 2
        idas.append((channels[j], list(in_ch[j:]), [s // scales[j] for s in scales[j:]]))
        scales[j + 1:] = [scales[j]] * len(scales[j + 1:])
        in_ch[j + 1:] = [channels[j]] * len(in_ch[j + 1:])
    return idas


def _emit_conv(out, key, cout, cin_per_group, k, bias, gen):
    w = torch.empty(cout, cin_per_group, k, k)
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5), generator=gen)
    out[key + ".weight"] = w
    if bias:
        bound = 1.0 / math.sqrt(cin_per_group * k * k)
        b = torch.empty(cout)
        torch.nn.init.uniform_(b, -bound, bound, generator=gen)
        out[key + ".bias"] = b


def _emit_bn(out, key, c):
    out[key + ".weight"] = torch.ones(c)
    out[key + ".bias"] = torch.zeros(c)
    out[key + ".running_mean"] = torch.zeros(c)
    out[key + ".running_var"] = torch.ones(c)
    out[key + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def _emit_block(out, p, gen):
    _emit_conv(out, p["prefix"] + ".conv1", p["cout"], p["cin"], 3, False, gen)
    _emit_bn(out, p["prefix"] + ".bn1", p["cout"])
    _emit_conv(out, p["prefix"] + ".conv2", p["cout"], p["cout"], 3, False, gen)
    _emit_bn(out, p["prefix"] + ".bn2", p["cout"])


def _emit_tree(out, t, gen):
    for sub in (t["tree1"], t["tree2"]):
        (_emit_block if sub["kind"] == "block" else _emit_tree)(out, sub, gen)
    if t["levels"] == 1:
        _emit_conv(out, t["prefix"] + ".root.conv", t["cout"], t["root_dim"], 1, False, gen)
        _emit_bn(out, t["prefix"] + ".root.bn", t["cout"])
    if t["project"]:
        _emit_conv(out, t["prefix"] + ".project.0", t["cout"], t["cin"], 1, False, gen)
        _emit_bn(out, t["prefix"] + ".project.1", t["cout"])


def trees():
    b = "backbone.base."
    c = CHANNELS
    return [tree_plan(b + "level_2", LEVELS[2], c[1], c[2], 2, False), tree_plan(b + "level_3", LEVELS[3], c[2], c[3], 2, True),
            tree_plan(b + "level_4", LEVELS[4], c[3], c[4], 2, True), tree_plan(b + "level_5", LEVELS[5], c[4], c[5], 2, True)]


def init_state_dict(nc: int = 80, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """The reference's ``CenterNet(cfg)`` under ``torch.manual_seed(seed)``: same draws in the same (construction) order --
    torch's default Conv2d / ConvTranspose2d initialisation; the unused classifier ``base.final`` (512 -> 1000, with bias)
    draws too and is part of the ``state_dict``."""
    gen = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    b = "backbone.base."
    c = CHANNELS
    _emit_conv(sd, b + "base_layer.0", c[0], 3, 7, False, gen)
    _emit_bn(sd, b + "base_layer.1", c[0])
    _emit_conv(sd, b + "level_0.0", c[0], c[0], 3, False, gen)
    _emit_bn(sd, b + "level_0.1", c[0])
    _emit_conv(sd, b + "level_1.0", c[1], c[0], 3, False, gen)
    _emit_bn(sd, b + "level_1.1", c[1])
    for t in trees():
        _emit_tree(sd, t, gen)
    _emit_conv(sd, b + "final", 1000, c[5], 1, True, gen)
    for i, (out_dim, in_ch, ups) in enumerate(dla_up_plan()):
        p = f"backbone.dla_up.ida_{i}."
        for k, (ci, f) in enumerate(zip(in_ch, ups)):
            if ci != out_dim:
                _emit_conv(sd, p + f"proj_{k}.0", out_dim, ci, 1, False, gen)
                _emit_bn(sd, p + f"proj_{k}.1", out_dim)
            if f != 1:
                _emit_conv(sd, p + f"up_{k}", out_dim, 1, 2 * f, False, gen)        # ConvTranspose2d weight (C, 1, 2f, 2f)
        for k in range(1, len(in_ch)):
            _emit_conv(sd, p + f"node_{k}.0", out_dim, 2 * out_dim, 3, False, gen)
            _emit_bn(sd, p + f"node_{k}.1", out_dim)
    for head, classes in (("heatmap", nc), ("wh", 2), ("reg", 2)):
        _emit_conv(sd, f"backbone.{head}.0", HEAD_CONV, c[2], 3, True, gen)
        _emit_conv(sd, f"backbone.{head}.2", classes, HEAD_CONV, 1, True, gen)
    return sd


# ----------------------------------------------------------------------------------------------
# forward
# ----------------------------------------------------------------------------------------------
# fp16-STORAGE emulation (as in oracle/yolov8_ref.py): with FP16_STORAGE[0] = True every convolution sees its input and its
# weights rounded to fp16 -- exactly what the MI355X engine's MFMA kernels read -- and accumulates in fp32.
FP16_STORAGE = [False]


def _q(t):
    if not FP16_STORAGE[0]:
        return t
    if t.requires_grad:                       # straight-through: the value is rounded, the gradient passes unrounded
        return t + (t.detach().half().float() - t.detach())
    return t.half().float()


def _conv(x, w, b=None, stride=1, pad=0):
    return F.conv2d(_q(x), _q(w), b, stride, pad)


def _bn(x, sd, key, training):
    if training:
        sd[key + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"],
                        training, BN_MOMENTUM, BN_EPS)


def _block(x, sd, p, residual, training):
    """BasicBlock.forward (centernet_model.py:20-27): the residual is added BEFORE the ReLU."""
    pre = p["prefix"]
    y = F.relu(_bn(_conv(x, sd[pre + ".conv1.weight"], None, p["stride"], 1), sd, pre + ".bn1", training))
    y = _bn(_conv(y, sd[pre + ".conv2.weight"], None, 1, 1), sd, pre + ".bn2", training)
    return F.relu(y + residual)


def _tree(x, sd, t, training, residual=None, children=None):
    """Tree.forward (centernet_model.py:138-152)."""
    children = [] if children is None else children
    pre = t["prefix"]
    bottom = F.max_pool2d(x, 2, 2) if t["stride"] > 1 else x
    if t["project"]:
        residual = _bn(_conv(bottom, sd[pre + ".project.0.weight"]), sd, pre + ".project.1", training)
    else:
        residual = bottom
    if t["level_root"]:
        children.append(bottom)
    if t["levels"] == 1:
        x1 = _block(x, sd, t["tree1"], residual, training)
        x2 = _block(x1, sd, t["tree2"], x1, training)
        cat = torch.cat([x2, x1, *children], 1)
        return F.relu(_bn(_conv(cat, sd[pre + ".root.conv.weight"]), sd, pre + ".root.bn", training))   # root_residual False
    x1 = _tree(x, sd, t["tree1"], training, residual=residual)
    children.append(x1)
    return _tree(x1, sd, t["tree2"], training, children=children)


def _ida(layers, sd, p, out_dim, in_ch, ups, training):
    """IDAUp.forward (centernet_model.py:268-279)."""
    layers = list(layers)
    for k, (ci, f) in enumerate(zip(in_ch, ups)):
        l = layers[k]
        if ci != out_dim:
            l = F.relu(_bn(_conv(l, sd[p + f"proj_{k}.0.weight"]), sd, p + f"proj_{k}.1", training))
        if f != 1:
            l = F.conv_transpose2d(_q(l), sd[p + f"up_{k}.weight"], None, stride=f, padding=f // 2, groups=out_dim)
        layers[k] = l
    x, ys = layers[0], []
    for k in range(1, len(layers)):
        x = F.relu(_bn(_conv(torch.cat([x, layers[k]], 1), sd[p + f"node_{k}.0.weight"], None, 1, 1), sd, p + f"node_{k}.1", training))
        ys.append(x)
    return x, ys


def forward(sd, x: torch.Tensor, nc: int = 80, training: bool = False) -> torch.Tensor:
    """CenterNet.forward (centernet_model.py:371-379): (B,3,H,W) -> (B, H/4, W/4, nc + 4) = [heatmap | wh | reg] heads, NHWC."""
    b = "backbone.base."
    x = F.relu(_bn(_conv(x, sd[b + "base_layer.0.weight"], None, 1, 3), sd, b + "base_layer.1", training))
    x = F.relu(_bn(_conv(x, sd[b + "level_0.0.weight"], None, 1, 1), sd, b + "level_0.1", training))
    x = F.relu(_bn(_conv(x, sd[b + "level_1.0.weight"], None, 2, 1), sd, b + "level_1.1", training))
    ys = []
    for t in trees():
        x = _tree(x, sd, t, training)
        ys.append(x)
    layers = list(ys)                                              # DLAUp.forward (centernet_model.py:298-305)
    for i, (out_dim, in_ch, ups) in enumerate(dla_up_plan()):
        x, y = _ida(layers[-i - 2:], sd, f"backbone.dla_up.ida_{i}.", out_dim, in_ch, ups, training)
        layers[-i - 1:] = y
    outs = []
    for head in ("heatmap", "wh", "reg"):
        h = F.relu(_conv(x, sd[f"backbone.{head}.0.weight"], sd[f"backbone.{head}.0.bias"], 1, 1))
        outs.append(_conv(h, sd[f"backbone.{head}.2.weight"], sd[f"backbone.{head}.2.bias"]))
    return torch.cat(outs, 1).permute(0, 2, 3, 1)


# ----------------------------------------------------------------------------------------------
# decode
# ----------------------------------------------------------------------------------------------
IOU_EPS = 1e-6                                    # core/utils/iou.py:5


def box_diou(b1: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """core/utils/iou.py:8-64, xyxy boxes, fp32, the reference's operation order."""
    a1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    a2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    wh = torch.clamp(torch.minimum(b1[..., 2:4], b2[..., 2:4]) - torch.maximum(b1[..., 0:2], b2[..., 0:2]), min=0)
    inter = wh[..., 0] * wh[..., 1]
    iou = inter / torch.clamp(a1 + a2 - inter, min=IOU_EPS)
    c1, c2 = (b1[..., 0:2] + b1[..., 2:4]) / 2, (b2[..., 0:2] + b2[..., 2:4]) / 2
    enc = torch.clamp(torch.maximum(b1[..., 2:4], b2[..., 2:4]) - torch.minimum(b1[..., 0:2], b2[..., 0:2]), min=0)
    c_sq = torch.sum(torch.pow(enc, 2), dim=-1)
    d_sq = torch.sum(torch.pow(c1 - c2, 2), dim=-1)
    return iou - d_sq / torch.clamp(c_sq, min=IOU_EPS)


def diou_nms(boxes: torch.Tensor, scores: torch.Tensor, thr: float) -> torch.Tensor:
    """core/utils/nms.py:9-31: greedy, class-agnostic; a box survives a kept box when DIoU <= thr.  Ties in the score sort:
    lower index first."""
    order = torch.sort(scores, descending=True, stable=True).indices
    alive = torch.ones(order.numel(), dtype=torch.bool)
    keep = []
    for pos in range(order.numel()):
        if not alive[pos]:
            continue
        i = int(order[pos])
        keep.append(i)
        rest = order[pos + 1:]
        if rest.numel():
            alive[pos + 1:] &= box_diou(boxes[i], boxes[rest]) <= thr
    return torch.tensor(keep, dtype=torch.long)


def suppress_and_topk(pred: torch.Tensor, nc: int, k: int):
    """sigmoid -> the reference's (x, class)-window 3x3 max-pool -> top-k over H*W*C per image
    (centernet.py:279-281,313-336).  Returns scores (B,k), flat indices (B,k) into (H, W, C)."""
    heat = torch.sigmoid(pred[..., :nc])                                             # (B, H, W, C)
    hmax = F.max_pool2d(heat, 3, 1, 1)      # on a (B,H,W,C) tensor: "channels" = H, the window runs over (W, C)
    heat = heat * (heat == hmax).float()
    B = heat.shape[0]
    flat = heat.reshape(B, -1)
    order = torch.sort(flat, dim=1, descending=True, stable=True).indices[:, :k]   # (score desc, index asc)
    return flat.gather(1, order), order


def decode(pred: torch.Tensor, nc: int, input_hw, image_hw, k: int = 100, conf: float = 0.1, nms_thr: float = 0.5,
           use_nms: bool = True):
    """CenterNetA.decode_boxes (centernet.py:271-311) for ONE image (the reference flattens the batch before the score mask
    and the NMS, so per-image semantics hold for B = 1 only): -> boxes (n,4) xyxy in original-image pixels, scores, classes,
    and the positions (0..k-1) of the survivors in the top-k list."""
    assert pred.shape[0] == 1
    _, H, W, _ = pred.shape
    scores, inds = suppress_and_topk(pred, nc, k)
    cls = inds % nc
    pixel = torch.div(inds, nc, rounding_mode="floor")
    ys, xs = torch.div(pixel, W, rounding_mode="floor"), pixel % W
    pix = (ys * W + xs).long()
    feat = pred.reshape(1, H * W, nc + 4)
    reg = feat[..., nc:nc + 2].gather(1, pix.unsqueeze(2).expand(-1, -1, 2))         # the reference reads offsets from the "wh" head
    wh = feat[..., nc + 2:].gather(1, pix.unsqueeze(2).expand(-1, -1, 2))            # ... and sizes from the "reg" head
    xs = xs.float() + reg[..., 0]
    ys = ys.float() + reg[..., 1]
    bb = torch.cat((xs.unsqueeze(-1), ys.unsqueeze(-1), wh), -1)
    bb[..., ::2] /= W
    bb[..., 1::2] /= H
    bb = torch.clamp(bb, min=0, max=1)
    bb = torch.cat((bb[..., 0:1] - bb[..., 2:3] / 2, bb[..., 1:2] - bb[..., 3:4] / 2, bb[..., 0:1] + bb[..., 2:3] / 2,
                    bb[..., 1:2] + bb[..., 3:4] / 2), -1)
    mask = scores >= conf
    pos = torch.nonzero(mask[0]).flatten()
    bb, sc, cl = bb[mask], scores[mask], cls[mask]
    if use_nms and bb.shape[0] > 0:
        keep = diou_nms(bb, sc, nms_thr)
        bb, sc, cl, pos = bb[keep], sc[keep], cl[keep], pos[keep]
    h, w = image_hw                                                 # reverse_letter_box (image_process.py:100-129), xyxy input
    out = bb.clone()
    out[..., ::2] *= input_hw[1]
    out[..., 1::2] *= input_hw[0]
    scale = max(h / input_hw[0], w / input_hw[1])
    top = (input_hw[0] - h / scale) // 2
    left = (input_hw[1] - w / scale) // 2
    out[..., 0] -= left
    out[..., 2] -= left
    out[..., 1] -= top
    out[..., 3] -= top
    out *= scale
    return out, sc, cl, pos


def projection_weights(shape, seed: int = 9):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def projection_loss(out, weights):
    """mean(out * w): a fixed linear functional of the (B, H/4, W/4, nc + 4) output the backward parity runs on."""
    return (out * weights).mean()


def loss_and_grads(sd, x, nc: int = 80, weights=None, seed: int = 9):
    """Train-mode forward + backward of ``projection_loss``: (loss, {key: grad}, out); running statistics in ``sd`` are updated."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()
              if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var"))}
    work = dict(sd)
    work.update(params)
    out = forward(work, x, nc, training=True)
    if weights is None:
        weights = projection_weights(out.shape, seed)
    loss = projection_loss(out, weights)
    loss.backward()
    return loss.detach(), {k: p.grad for k, p in params.items() if p.grad is not None}, out.detach()


def combined_loss(pred, targets, nc: int, hm_weight: float = 1.0, wh_weight: float = 0.1, off_weight: float = 1.0):
    """CombinedLoss.__call__ (core/loss/centernet_loss.py:46-67) with FocalLoss (:5-26) and RegL1Loss (:29-43): pred (B, h, w, nc + 4),
    targets = [heatmap_true (B,h,w,nc), reg_true (B,K,2), wh_true (B,K,2), reg_mask (B,K), indices (B,K)].  The "reg" term reads
    pred[..., nc:nc+2], the "wh" term pred[..., -2:] (the model emits [heatmap | wh head | reg head]: the names are swapped in the
    reference, kept).  Returns (total, heat-map, L1 reg, L1 wh)."""
    heat_t, reg_t, wh_t, mask, idx = targets
    heat = torch.clamp(torch.sigmoid(pred[..., :nc]), min=1e-4, max=1.0 - 1e-4)
    pos, neg = torch.eq(heat_t, 1).float(), torch.lt(heat_t, 1).float()
    num_pos = pos.sum()
    pos_loss = (torch.log(heat) * torch.pow(1 - heat, 2) * pos).sum()
    neg_loss = (torch.log(1 - heat) * torch.pow(heat, 2) * torch.pow(1 - heat_t, 4) * neg).sum()
    hm = -neg_loss if float(num_pos) == 0 else -(pos_loss + neg_loss) / num_pos

    def l1(feat, true):
        f = feat.reshape(feat.shape[0], -1, feat.shape[3])
        g = torch.gather(f, 1, idx.unsqueeze(2).long().expand(idx.shape[0], idx.shape[1], f.shape[2]))
        m = mask.unsqueeze(2).expand_as(g).float()
        return F.l1_loss(g * m, true * m, reduction="sum") / (m.sum() + 1e-4)

    off, wh = l1(pred[..., nc:nc + 2], reg_t), l1(pred[..., -2:], wh_t)
    return hm_weight * hm + off_weight * off + wh_weight * wh, hm, off, wh


def synth_targets(B: int, h: int, w: int, nc: int, K: int = 30, seed: int = 3):
    """Seeded targets in the format of CenterNet.generate_targets (core/algorithms/centernet.py:66-120): Gaussian bumps with an exact 1
    at each centre (the later object wins the maximum), sub-pixel offsets, sizes, mask and flat indices.  NOT a restatement of the
    reference's radius rule -- synthetic inputs for the loss kernel only."""
    g = torch.Generator().manual_seed(seed)
    heat = torch.zeros(B, h, w, nc)
    reg, wh, mask, idx = torch.zeros(B, K, 2), torch.zeros(B, K, 2), torch.zeros(B, K), torch.zeros(B, K, dtype=torch.long)
    ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    for b in range(B):
        n = int(torch.randint(1, max(2, min(K, 8)), (1,), generator=g))
        for k in range(n):
            cx, cy = float(torch.rand(1, generator=g)) * (w - 1), float(torch.rand(1, generator=g)) * (h - 1)
            bw, bh = 2 + float(torch.rand(1, generator=g)) * w / 3, 2 + float(torch.rand(1, generator=g)) * h / 3
            c = int(torch.randint(0, nc, (1,), generator=g))
            ix, iy = int(cx), int(cy)
            sigma = max(1.0, min(bw, bh) / 6)
            bump = torch.exp(-((xs - ix) ** 2 + (ys - iy) ** 2) / (2 * sigma * sigma))
            heat[b, :, :, c] = torch.maximum(heat[b, :, :, c], bump)
            reg[b, k] = torch.tensor([cx - ix, cy - iy])
            wh[b, k] = torch.tensor([bw, bh])
            mask[b, k] = 1.0
            idx[b, k] = iy * w + ix
    return [heat, reg, wh, mask, idx]


def gaussian_radius(det_size, min_overlap=0.7):
    """core/utils/gaussian.py:5-25 (the CornerNet radius), float64."""
    height, width = det_size
    b1, c1 = (height + width), width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + np.sqrt(b1 ** 2 - 4 * 1 * c1)) / 2
    b2, c2 = 2 * (height + width), (1 - min_overlap) * width * height
    r2 = (b2 + np.sqrt(b2 ** 2 - 4 * 4 * c2)) / 2
    a3, b3, c3 = 4 * min_overlap, -2 * min_overlap * (height + width), (min_overlap - 1) * width * height
    r3 = (b3 + np.sqrt(b3 ** 2 - 4 * a3 * c3)) / 2
    return min(r1, r2, r3)


def generate_targets(label, feature_hw, nc: int, max_num_boxes: int = 30):
    """CenterNet.generate_targets (core/algorithms/centernet.py:66-112) with gaussian2D / draw_umich_gaussian (gaussian.py:28-57), numpy,
    the reference's dtypes: label (N, 6) float32 [_, class id, cx, cy, w, h] -> (heatmap (h, w, nc), reg (K, 2), wh (K, 2), reg_mask (K,),
    ind (K,)) float32."""
    H, W = feature_hw
    label = np.array(label, dtype=np.float32, copy=True)
    c = label[:, 2:]
    rows = np.concatenate((c[:, 0:1] - c[:, 2:3] / 2, c[:, 1:2] - c[:, 3:4] / 2, c[:, 0:1] + c[:, 2:3] / 2, c[:, 1:2] + c[:, 3:4] / 2, label[:, 1:2]), -1)
    rows = rows[:max_num_boxes]
    hm = np.zeros((H, W, nc), dtype=np.float32)
    reg, wh = np.zeros((max_num_boxes, 2), dtype=np.float32), np.zeros((max_num_boxes, 2), dtype=np.float32)
    mask, ind = np.zeros((max_num_boxes,), dtype=np.float32), np.zeros((max_num_boxes,), dtype=np.float32)
    for j, item in enumerate(rows):
        item[:4:2] = item[:4:2] * W
        item[1:4:2] = item[1:4:2] * H
        xmin, ymin, xmax, ymax, cid = item
        cid = cid.astype(np.int32)
        h, w = int(ymax - ymin), int(xmax - xmin)
        radius = max(0, int(gaussian_radius((h, w))))
        ctr = np.array([(xmin + xmax) / 2, (ymin + ymax) / 2], dtype=np.float32)
        ci = ctr.astype(np.int32)
        d = 2 * radius + 1
        m = (d - 1.0) / 2.0
        yy, xx = np.ogrid[-m:m + 1, -m:m + 1]
        gk = np.exp(-(xx * xx + yy * yy) / (2 * (d / 6) * (d / 6)))
        gk[gk < np.finfo(gk.dtype).eps * gk.max()] = 0
        x, y = int(ci[0]), int(ci[1])
        left, right, top, bottom = min(x, radius), min(W - x, radius + 1), min(y, radius), min(H - y, radius + 1)
        mh = hm[:, :, cid][y - top:y + bottom, x - left:x + right]
        mg = gk[radius - top:radius + bottom, radius - left:radius + right]
        if min(mg.shape) > 0 and min(mh.shape) > 0:
            plane = hm[:, :, cid].copy()
            np.maximum(plane[y - top:y + bottom, x - left:x + right], mg, out=plane[y - top:y + bottom, x - left:x + right])
            hm[:, :, cid] = plane
        reg[j] = ctr - ci
        wh[j] = np.array([w, h], dtype=np.float32)
        mask[j] = 1
        ind[j] = ci[1] * W + ci[0]
    return hm, reg, wh, mask, ind
