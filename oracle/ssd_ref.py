"""CPU oracle for SSD300-VGG16(BN): network forward (eval / train mode) and backward, prior boxes, decode, per-class NMS -- TEST
INFRASTRUCTURE, NOT PRODUCT CODE.

SURVEY.md section 8 row a17 / (f)4.  A torch-CPU fp32 restatement, functional over a flat ``state_dict`` with the
reference's keys, of

* ``SSD`` (core/models/ssd_model.py:6-191): VGG16 with BatchNorm (Conv2d(bias) + BN + ReLU; 2x2 pools, the third with
  ceil_mode; 3x3/1 pool5; dilated conv6; conv7), ``L2Normalize`` on conv4_3, ``ExtraLayer`` (eight convolutions WITHOUT
  activations, ssd_model.py:90-110), six (loc, conf) 3x3 heads.  Quirk reproduced as is: the head outputs are flattened in
  NCHW order (no permute, :177-183), so a "box" of the (B, 8732, 4) tensor is four consecutive elements of the
  channel-major map, not the four regressions of one prior;
* the prior boxes (core/algorithms/ssd.py:482-535), ``_parse_mbox_loc`` (:285-325) and ``decode_boxes`` (:236-283) up to
  the letterbox inverse: softmax, per class c = 1..num_classes (column 0 is the background), score > threshold, greedy NMS.  ``torchvision.ops.nms`` is third party and
  absent: restated as the standard greedy algorithm (oracle/nms_ref.py) -- parity unpinned upstream, as for the YOLO tails.

Training (``forward(training=True)``, ``projection_loss``, ``loss_and_grads``): batch-statistics BatchNorm (running statistics
updated in place); the backward pass is torch autograd over this restatement, driven by a fixed linear functional of (loc, conf)
(the reference's MultiBoxLossV2, core/loss/multi_box_loss.py, is torch code on those tensors and is not restated).

Parity pin: ``oracle/make_golden.py`` section 12 imports the real reference and asserts init bit for bit, forward to fp32
round-off, priors and decoded boxes exactly; section 12b asserts the train-mode outputs and every parameter gradient of
``projection_loss`` against the real model's autograd.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

VGG_PARAMS = (64, 64, "M", 128, 128, "M", 256, 256, 256, "C", 512, 512, 512, "M", 512, 512, 512)
BN_EPS = 1e-5
ASPECT_RATIOS = ([1, 2, 0.5], [1, 2, 0.5, 3, 1.0 / 3], [1, 2, 0.5, 3, 1.0 / 3], [1, 2, 0.5, 3, 1.0 / 3], [1, 2, 0.5], [1, 2, 0.5])
FEATURE_SHAPES = (38, 19, 10, 5, 3, 1)
FEATURE_CHANNELS = (512, 1024, 512, 256, 256, 256)
ANCHOR_SIZES = (30, 60, 111, 162, 213, 264, 315)
BOXES_PER_PIXEL = tuple(len(a) + 1 for a in ASPECT_RATIOS)
EXTRAS = (("conv1", 256, 1024, 1, 1, 0), ("conv2", 512, 256, 3, 2, 1), ("conv3", 128, 512, 1, 1, 0), ("conv4", 256, 128, 3, 2, 1),
          ("conv5", 128, 256, 1, 1, 0), ("conv6", 256, 128, 3, 1, 0), ("conv7", 128, 256, 1, 1, 0), ("conv8", 256, 128, 3, 1, 0))


def vgg_plan():
    """backbone.layers as data: [(index, kind, ...)] -- kind 'conv' (cout, cin, k, pad, dil, bn_index), 'M', 'C', 'P5'."""
    plan, idx, cin = [], 0, 3
    for v in VGG_PARAMS:
        if v in ("M", "C"):
            plan.append((idx, v))
            idx += 1
        else:
            plan.append((idx, "conv", v, cin, 3, 1, 1, idx + 1))
            idx += 3                                              # conv, bn, relu
            cin = v
    plan.append((idx, "P5"))
    plan.append((idx + 1, "conv", 1024, cin, 3, 6, 6, None))     # dilated conv6 (+ ReLU at idx + 2)
    plan.append((idx + 3, "conv", 1024, 1024, 1, 0, 1, None))    # conv7 (+ ReLU at idx + 4)
    return plan


def init_state_dict(nc: int = 20, seed: int = 0):
    """torch's default initialisation in the reference's construction order (ssd_model.py:9-38, 66-88, 131-162): VGG
    layers, extras conv1..8, then loc_i / conf_i alternately; state_dict order lists locs before confs."""
    torch.manual_seed(seed)
    sd = OrderedDict()

    def conv(key, cout, cin, k):
        w = torch.empty(cout, cin, k, k)
        torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        b = torch.empty(cout)
        bound = 1.0 / math.sqrt(cin * k * k)
        torch.nn.init.uniform_(b, -bound, bound)
        return w, b

    for item in vgg_plan():
        if item[1] != "conv":
            continue
        idx, _, cout, cin, k, _, _, bn = item
        sd[f"backbone.layers.{idx}.weight"], sd[f"backbone.layers.{idx}.bias"] = conv(None, cout, cin, k)
        if bn is not None:
            sd[f"backbone.layers.{bn}.weight"], sd[f"backbone.layers.{bn}.bias"] = torch.ones(cout), torch.zeros(cout)
            sd[f"backbone.layers.{bn}.running_mean"], sd[f"backbone.layers.{bn}.running_var"] = torch.zeros(cout), torch.ones(cout)
            sd[f"backbone.layers.{bn}.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    sd["l2_norm.weight"] = torch.full((512,), 20.0)
    for name, cout, cin, k, _, _ in EXTRAS:
        sd[f"extras.{name}.weight"], sd[f"extras.{name}.bias"] = conv(None, cout, cin, k)
    heads = {}
    for i, (c, n) in enumerate(zip(FEATURE_CHANNELS, BOXES_PER_PIXEL)):
        heads[f"locs.{i}"] = conv(None, n * 4, c, 3)
        heads[f"confs.{i}"] = conv(None, n * (nc + 1), c, 3)
    for grp in ("locs", "confs"):
        for i in range(6):
            sd[f"{grp}.{i}.weight"], sd[f"{grp}.{i}.bias"] = heads[f"{grp}.{i}"]
    return sd


FP16_STORAGE = [False]


def _q(t):
    if not FP16_STORAGE[0]:
        return t
    if t.requires_grad:                       # straight-through: the value is rounded, the gradient passes unrounded
        return t + (t.detach().half().float() - t.detach())
    return t.half().float()


def forward(sd, x, nc: int = 20, training: bool = False, return_maps: bool = False):
    """SSD.forward (ssd_model.py:164-191) -> (loc (B, 8732, 4), conf (B, 8732, nc + 1)); ``return_maps``: also the 12 head maps."""
    y, x1 = x, None
    for item in vgg_plan():
        kind = item[1]
        if kind == "M":
            y = F.max_pool2d(y, 2, 2)
        elif kind == "C":
            y = F.max_pool2d(y, 2, 2, ceil_mode=True)
        elif kind == "P5":
            y = F.max_pool2d(y, 3, 1, 1)
        else:
            idx, _, cout, cin, k, pad, dil, bn = item
            y = F.conv2d(_q(y), _q(sd[f"backbone.layers.{idx}.weight"]), sd[f"backbone.layers.{idx}.bias"], 1, pad, dil)
            if bn is not None:
                p = f"backbone.layers.{bn}"
                y = F.batch_norm(y, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], training, 0.1, BN_EPS)
            y = _q(F.relu(y))
            if bn == 31:                                          # extract_index 32 = the ReLU after conv4_3's BatchNorm (ssd_model.py:52)
                x1 = y
    norm = x1.pow(2).sum(1, keepdim=True).sqrt() + 1e-10
    sources = [_q(sd["l2_norm.weight"].view(1, -1, 1, 1) * (x1 / norm)), y]
    e = y
    for j, (name, cout, cin, k, s, p) in enumerate(EXTRAS):
        e = _q(F.conv2d(_q(e), _q(sd[f"extras.{name}.weight"]), sd[f"extras.{name}.bias"], s, p))
        if j % 2 == 1:
            sources.append(e)
    locs = [F.conv2d(_q(s_), _q(sd[f"locs.{i}.weight"]), sd[f"locs.{i}.bias"], 1, 1) for i, s_ in enumerate(sources)]
    confs = [F.conv2d(_q(s_), _q(sd[f"confs.{i}.weight"]), sd[f"confs.{i}.bias"], 1, 1) for i, s_ in enumerate(sources)]
    B = x.shape[0]
    loc = torch.cat([o.reshape(B, -1) for o in locs], 1).reshape(B, -1, 4)
    conf = torch.cat([o.reshape(B, -1) for o in confs], 1).reshape(B, -1, nc + 1)
    return (loc, conf, locs, confs) if return_maps else (loc, conf)


def priors(input_hw=(300, 300)):
    """Ssd._get_ssd_anchors (core/algorithms/ssd.py:482-535): (8732, 4) corner boxes in [0, 1], float32."""
    image_h, image_w = input_hw
    out = []
    for i, fh in enumerate(FEATURE_SHAPES):
        mn, mx = ANCHOR_SIZES[i], ANCHOR_SIZES[i + 1]
        ws, hs = [], []
        for ar in ASPECT_RATIOS[i]:
            if ar == 1:
                ws += [mn, np.sqrt(mn * mx)]
                hs += [mn, np.sqrt(mn * mx)]
            else:
                ws.append(mn * np.sqrt(ar))
                hs.append(mn / np.sqrt(ar))
        hw_, hh_ = np.array(ws) / 2.0, np.array(hs) / 2.0
        pl = [image_h / fh, image_w / fh]
        cx = np.linspace(0.5 * pl[1], image_w - 0.5 * pl[1], fh)
        cy = np.linspace(0.5 * pl[0], image_h - 0.5 * pl[0], fh)
        gx, gy = np.meshgrid(cx, cy)
        a = np.tile(np.concatenate((gx.reshape(-1, 1), gy.reshape(-1, 1)), 1), (1, (len(ASPECT_RATIOS[i]) + 1) * 2))
        a[:, ::4] -= hw_
        a[:, 1::4] -= hh_
        a[:, 2::4] += hw_
        a[:, 3::4] += hh_
        a[:, ::2] /= image_w
        a[:, 1::2] /= image_h
        out.append(np.clip(a, 0.0, 1.0).reshape(-1, 4))
    return np.concatenate(out, 0).astype(np.float32)


def parse_loc(mbox_loc, anchors, variances=(0.1, 0.2)):
    """Ssd._parse_mbox_loc (ssd.py:285-325): (A, 4) regressions -> clipped corner boxes."""
    a = torch.from_numpy(anchors).float()
    aw, ah = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
    acx, acy = 0.5 * (a[:, 2] + a[:, 0]), 0.5 * (a[:, 3] + a[:, 1])
    cx = mbox_loc[:, 0] * aw * variances[0] + acx
    cy = mbox_loc[:, 1] * ah * variances[0] + acy
    w = torch.exp(mbox_loc[:, 2] * variances[1]) * aw
    h = torch.exp(mbox_loc[:, 3] * variances[1]) * ah
    box = torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), -1)
    return torch.min(torch.max(box, torch.zeros_like(box)), torch.ones_like(box))


def nms_per_class(boxes, prob, nc: int, conf_threshold: float, nms_threshold: float):
    """The class loop of Ssd.decode_boxes (ssd.py:252-274) on decoded corner boxes (A, 4) and softmax scores (A, nc + 1) of ONE
    image: (rows (n, 6) [x1, y1, x2, y2, label, conf], pairs (n, 2) kept (prior, class column))."""
    from . import nms_ref
    rows, pairs = [], []
    for c in range(1, nc + 1):
        sc = prob[:, c]
        idx = torch.nonzero(sc > conf_threshold).flatten()
        if idx.numel() == 0:
            continue
        s = sc[idx].numpy()
        order = np.argsort(-s, kind="stable")
        k = order[nms_ref._greedy(boxes[idx].numpy()[order], None, nms_threshold)]
        kept = idx[k]
        rows.append(torch.cat((boxes[kept], torch.full((len(k), 1), float(c - 1)), sc[kept, None]), 1))
        pairs.append(torch.stack((kept, torch.full_like(kept, c)), 1))
    if not rows:
        return np.zeros((0, 6), np.float32), np.zeros((0, 2), np.int64)
    return torch.cat(rows).numpy(), torch.cat(pairs).numpy()


def decode(loc, conf, anchors, nc: int, conf_threshold: float, nms_threshold: float):
    """Ssd.decode_boxes before the letterbox inverse (ssd.py:236-276): per image (rows, pairs) of ``nms_per_class`` --
    classes ascending, scores descending inside a class."""
    prob = torch.softmax(conf, -1)
    return [nms_per_class(parse_loc(loc[i], anchors), prob[i], nc, conf_threshold, nms_threshold) for i in range(loc.shape[0])]


def projection_weights(shapes, seed: int = 9):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*sh, generator=g) for sh in shapes]


def projection_loss(outs, weights):
    """mean(loc * w0) + mean(conf * w1): a fixed linear functional of the two outputs the backward parity runs on."""
    return sum((o * w).mean() for o, w in zip(outs, weights))


def loss_and_grads(sd, x, nc: int = 20, weights=None, seed: int = 9):
    """Train-mode forward + backward of ``projection_loss``: (loss, {key: grad}, (loc, conf)); running statistics in ``sd`` are updated."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()
              if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var"))}
    work = dict(sd)
    work.update(params)
    outs = forward(work, x, nc, training=True)
    if weights is None:
        weights = projection_weights([o.shape for o in outs], seed)
    loss = projection_loss(outs, weights)
    loss.backward()
    return loss.detach(), {k: p.grad for k, p in params.items() if p.grad is not None}, [o.detach() for o in outs]


def multibox_loss(y_true, loc, conf, neg_pos_ratio: float = 3.0, alpha: float = 0.5):
    """MultiBoxLossV2.__call__ (core/loss/multi_box_loss.py:117-192): y_true (B, A, 4 + nc1 + 1), loc (B, A, 4), conf (B, A, nc1) logits ->
    (total, loc_loss, conf_loss).  Hard negatives: the k anchors of the WHOLE batch with the largest non-background probability mass
    among the non-positives, k = sum_b min(ratio * num_pos_b, A - num_pos_b) (100 when that is 0 for every image)."""
    A = y_true.shape[1]
    nc1 = conf.shape[-1]
    p = torch.softmax(conf, -1)
    conf_loss = -(y_true[:, :, 4:-1] * torch.log(torch.clamp(p, min=1e-7))).sum(-1)
    d = y_true[:, :, :4] - loc
    loc_loss = torch.where(d.abs() < 1.0, 0.5 * d * d, d.abs() - 0.5).sum(-1)
    pos = y_true[:, :, -1]
    num_pos = pos.sum(-1)
    num_neg = torch.minimum(neg_pos_ratio * num_pos, A - num_pos)
    k = int(num_neg.sum()) if int((num_neg > 0).sum()) > 0 else 100
    hard = (p[:, :, 1:nc1].sum(2) * (1 - pos)).reshape(-1)
    _, idx = torch.topk(hard, k=k)
    neg = conf_loss.reshape(-1)[idx]
    norm = torch.where(num_pos != 0, num_pos, torch.ones_like(num_pos)).sum()
    c = ((conf_loss * pos).sum() + neg.sum()) / norm
    l = (loc_loss * pos).sum() / norm
    return c * (1 - alpha) + l * alpha, l, c


def synth_y_true(B: int, A: int, nc: int, n_pos: int = 12, seed: int = 5):
    """Seeded encoded targets in the format of Ssd.generate_targets (core/algorithms/ssd.py:327-480): box regression targets, one-hot
    class incl. the background column, positive flag -- synthetic inputs for the loss kernel (not the reference's prior matching)."""
    g = torch.Generator().manual_seed(seed)
    y = torch.zeros(B, A, 4 + nc + 1 + 1)
    y[:, :, 4] = 1.0                                              # background
    for b in range(B):
        k = int(torch.randint(0 if b else 1, n_pos + 1, (1,), generator=g))
        idx = torch.randperm(A, generator=g)[:k]
        y[b, idx, :4] = torch.randn(k, 4, generator=g) * 1.5
        y[b, idx, 4] = 0.0
        y[b, idx, 5 + torch.randint(0, nc, (k,), generator=g)] = 1.0
        y[b, idx, -1] = 1.0
    return y


def generate_targets(label, anchors, nc: int, overlap_threshold: float = 0.5, variance=(0.1, 0.1, 0.2, 0.2)):
    """Ssd.generate_targets + _encode_box (core/algorithms/ssd.py:327-480), numpy, the reference's dtypes: label (N, 6) float32
    [_, class id, cx, cy, w, h] -> corners in float32; then float64 arithmetic against the float32 anchors (the corners ride in a float64
    array with the one-hot labels, :343); result (A, 4 + (nc + 1) + 1) float32."""
    label = np.array(label, dtype=np.float32, copy=True)
    anchors = np.asarray(anchors, dtype=np.float32)
    variance = np.asarray(variance, dtype=np.float32)
    A = anchors.shape[0]
    label[:, 1] += 1
    cls = label[:, 1].astype(np.int32)
    c = label[:, 2:]
    coord = np.concatenate((c[:, 0:1] - c[:, 2:3] / 2, c[:, 1:2] - c[:, 3:4] / 2, c[:, 0:1] + c[:, 2:3] / 2, c[:, 1:2] + c[:, 3:4] / 2), -1)
    assignment = np.zeros((A, 4 + 1 + nc + 1), dtype=np.float32)
    assignment[:, 4] = 1.0
    if len(coord) == 0:
        return assignment
    enc = []
    for box in coord.astype(np.float64):                          # the reference concatenates the float32 corners with a float64 one-hot: float64 from here
        wh = np.maximum(np.minimum(anchors[:, 2:4], box[2:]) - np.maximum(anchors[:, :2], box[:2]), 0)
        inter = wh[:, 0] * wh[:, 1]
        iou = inter / ((box[2] - box[0]) * (box[3] - box[1]) + (anchors[:, 2] - anchors[:, 0]) * (anchors[:, 3] - anchors[:, 1]) - inter)
        e = np.zeros((A, 5))
        mask = iou > overlap_threshold
        if not mask.any():
            mask[iou.argmax()] = True
        e[:, -1][mask] = iou[mask]
        aa = anchors[mask]
        a_center, a_wh = (aa[:, 0:2] + aa[:, 2:4]) * 0.5, aa[:, 2:4] - aa[:, 0:2]
        e[:, :2][mask] = 0.5 * (box[:2] + box[2:]) - a_center
        e[:, :2][mask] /= a_wh
        e[:, :2][mask] /= variance[:2]
        e[:, 2:4][mask] = np.log((box[2:] - box[:2]) / a_wh)
        e[:, 2:4][mask] /= variance[2:4]
        enc.append(e)
    enc = np.stack(enc)
    best = enc[:, :, -1].max(0)
    idx = enc[:, :, -1].argmax(0)
    m = best > 0
    idx = idx[m]
    assignment[:, :4][m] = enc[:, m, :][idx, np.arange(len(idx)), :4]
    assignment[:, 4][m] = 0
    assignment[:, 5:-1][m] = np.eye(nc + 1)[cls][idx, 1:]
    assignment[:, -1][m] = 1
    return assignment
