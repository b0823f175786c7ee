"""CPU oracle for YOLOv7-l: network forward (eval / train mode) and backward, anchor decode, per-class NMS -- TEST INFRASTRUCTURE,
NOT PRODUCT CODE.

SURVEY.md section 8 row a16 / (f)3.  A torch-CPU fp32 restatement, functional over a flat ``state_dict`` with the
reference's keys, of

* ``Yolo7`` (core/models/yolov7_model.py:14-525, phi = 'l'): ConvBNSiLU (BatchNorm eps 1e-3, momentum 0.03), ELAN
  ``Multi_Concat_Block``, ``Transition_Block`` (2x2 max pool | stride-2 conv), ``SPPCSPC`` (5/9/13 max pools), nearest 2x
  up-sampling, PANet, ``RepConv`` in its training form (3x3+BN plus 1x1+BN, summed before SiLU; no identity branch since
  c1 != c2), three 1x1 heads with bias -> (out0 20x20, out1 40x40, out2 80x80) at 640 input, each (B, 3*(5+nc), H, W);
* ``YOLOv7.decode_box`` (core/algorithms/yolo_v7.py:234-346): sigmoid, grid / anchor decode to normalised (cx, cy, w, h),
  levels concatenated in the order of the outputs (coarsest first), anchors of a level anchor-major;
* ``YOLOv7._nms`` (:348-424) up to the letterbox inverse: corner boxes, score = objectness * best class probability >= the
  threshold, greedy NMS per class in ascending class order.  The suppression itself is ``torchvision.ops.nms`` -- third party,
  absent here (parity unpinned upstream, as for YOLOv8): restated as the standard greedy algorithm, ties broken by the
  lower index (oracle/nms_ref.py).

Training (``forward(training=True)``, ``projection_loss``, ``loss_and_grads``): batch-statistics BatchNorm with the running
statistics updated in place; the backward pass is torch autograd over this restatement, driven by a fixed linear functional of
the three outputs (the reference's Yolo7Loss, core/loss/yolo7_loss.py, is torch code on those tensors and is not restated).

Parity pin: ``oracle/make_golden.py`` section 11 imports the real reference and asserts the seed-0 ``state_dict`` bit for
bit, the forward to fp32 round-off and the decoded tensor (before NMS) exactly; section 11b asserts the train-mode outputs and
every parameter gradient of ``projection_loss`` against the real model's autograd.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS, BN_MOMENTUM = 1e-3, 0.03
TC, BC, PC, E, N = 32, 32, 32, 2, 4             # transition / block / panet channels, e, n for phi = 'l' (yolov7_model.py:366-371)
IDS_BACKBONE = (-1, -3, -5, -6)
IDS_NECK = (-1, -2, -3, -4, -5, -6)
ANCHORS = (12, 16, 19, 36, 40, 28, 36, 75, 76, 55, 72, 146, 142, 110, 192, 243, 459, 401)   # configs/yolo7_cfg.py
ANCHORS_MASK = ((6, 7, 8), (3, 4, 5), (0, 1, 2))


# ----------------------------------------------------------------------------------------------
# architecture as data: a list of modules in REGISTRATION order, each a list of (key, cout, cin, k, stride, bias)
# ----------------------------------------------------------------------------------------------
def _mcb(prefix, c1, c2, c3, n, e, ids):
    c_ = int(c2 * e)
    convs = [(prefix + ".cv1", c_, c1, 1, 1), (prefix + ".cv2", c_, c1, 1, 1)]
    convs += [(prefix + f".cv3.{i}", c2, c_ if i == 0 else c2, 3, 1) for i in range(n)]
    convs.append((prefix + ".cv4", c3, c_ * 2 + c2 * (len(ids) - 2), 1, 1))
    return convs


def _trans(prefix, c1, c2):
    return [(prefix + ".cv1", c2, c1, 1, 1), (prefix + ".cv2", c2, c1, 1, 1), (prefix + ".cv3", c2, c2, 3, 2)]


def conv_bn_specs():
    """Every ConvBNSiLU of the network, registration order: (key, cout, cin, k, stride)."""
    t, b = TC, BC
    s = [("backbone.stem.0", t, 3, 3, 1), ("backbone.stem.1", 2 * t, t, 3, 2), ("backbone.stem.2", 2 * t, 2 * t, 3, 1),
         ("backbone.dark2.0", 4 * t, 2 * t, 3, 2)]
    s += _mcb("backbone.dark2.1", 4 * t, 2 * b, 8 * t, N, 1, IDS_BACKBONE)
    s += _trans("backbone.dark3.0", 8 * t, 4 * t) + _mcb("backbone.dark3.1", 8 * t, 4 * b, 16 * t, N, 1, IDS_BACKBONE)
    s += _trans("backbone.dark4.0", 16 * t, 8 * t) + _mcb("backbone.dark4.1", 16 * t, 8 * b, 32 * t, N, 1, IDS_BACKBONE)
    s += _trans("backbone.dark5.0", 32 * t, 16 * t) + _mcb("backbone.dark5.1", 32 * t, 8 * b, 32 * t, N, 1, IDS_BACKBONE)
    c_ = 16 * t                                                   # SPPCSPC(32t, 16t): c_ = int(2 * c2 * 0.5)
    s += [("sppcspc.cv1", c_, 32 * t, 1, 1), ("sppcspc.cv2", c_, 32 * t, 1, 1), ("sppcspc.cv3", c_, c_, 3, 1), ("sppcspc.cv4", c_, c_, 1, 1),
          ("sppcspc.cv5", c_, 4 * c_, 1, 1), ("sppcspc.cv6", c_, c_, 3, 1), ("sppcspc.cv7", 16 * t, 2 * c_, 1, 1)]
    s += [("conv_for_P5", 8 * t, 16 * t, 1, 1), ("conv_for_feat2", 8 * t, 32 * t, 1, 1)]
    s += _mcb("conv3_for_upsample1", 16 * t, 4 * PC, 8 * t, N, E, IDS_NECK)
    s += [("conv_for_P4", 4 * t, 8 * t, 1, 1), ("conv_for_feat1", 4 * t, 16 * t, 1, 1)]
    s += _mcb("conv3_for_upsample2", 8 * t, 2 * PC, 4 * t, N, E, IDS_NECK)
    s += _trans("down_sample1", 4 * t, 4 * t) + _mcb("conv3_for_downsample1", 16 * t, 4 * PC, 8 * t, N, E, IDS_NECK)
    s += _trans("down_sample2", 8 * t, 8 * t) + _mcb("conv3_for_downsample2", 32 * t, 8 * PC, 16 * t, N, E, IDS_NECK)
    return s


REP = (("rep_conv_1", 4 * TC, 8 * TC), ("rep_conv_2", 8 * TC, 16 * TC), ("rep_conv_3", 16 * TC, 32 * TC))
HEADS = (("yolo_head_P3", 8 * TC), ("yolo_head_P4", 16 * TC), ("yolo_head_P5", 32 * TC))


def init_state_dict(nc: int = 20, seed: int = 0):
    """Construction (torch defaults, registration order) then ``init_weights`` (yolov7_model.py:449-458): conv weights
    N(0, 0.02), conv biases 0, BatchNorm weights N(1, 0.02), biases 0 -- all in ``modules()`` order, from the global RNG."""
    import math
    torch.manual_seed(seed)
    sd = OrderedDict()
    order = []                                                   # (kind, key) in registration order

    def new_conv(key, cout, cin, k, bias):
        w = torch.empty(cout, cin, k, k)
        torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        sd[key + ".weight"] = w
        if bias:
            bb = torch.empty(cout)
            bound = 1.0 / math.sqrt(cin * k * k)
            torch.nn.init.uniform_(bb, -bound, bound)
            sd[key + ".bias"] = bb
        order.append(("conv", key))

    def new_bn(key, c):
        sd[key + ".weight"], sd[key + ".bias"] = torch.ones(c), torch.zeros(c)
        sd[key + ".running_mean"], sd[key + ".running_var"] = torch.zeros(c), torch.ones(c)
        sd[key + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
        order.append(("bn", key))

    for key, cout, cin, k, _ in conv_bn_specs():
        new_conv(key + ".conv", cout, cin, k, False)
        new_bn(key + ".bn", cout)
    for key, c1, c2 in REP:                                       # RepConv.__init__: rbr_identity (None), rbr_dense, rbr_1x1
        new_conv(key + ".rbr_dense.0", c2, c1, 3, False)
        new_bn(key + ".rbr_dense.1", c2)
        new_conv(key + ".rbr_1x1.0", c2, c1, 1, False)
        new_bn(key + ".rbr_1x1.1", c2)
    for key, c in HEADS:
        new_conv(key, 3 * (5 + nc), c, 1, True)
    for kind, key in order:                                       # init_weights, modules() order == registration order
        if kind == "conv":
            torch.nn.init.normal_(sd[key + ".weight"], 0, 0.02)
            if key + ".bias" in sd:
                sd[key + ".bias"].zero_()
        else:
            torch.nn.init.normal_(sd[key + ".weight"], 1, 0.02)
            sd[key + ".bias"].zero_()
    return sd


# fp16-STORAGE emulation (see oracle/yolov8_ref.py): conv operands and stored activations rounded to fp16
FP16_STORAGE = [False]


def _q(t):
    if not FP16_STORAGE[0]:
        return t
    if t.requires_grad:                       # straight-through: the value is rounded, the gradient passes unrounded
        return t + (t.detach().half().float() - t.detach())
    return t.half().float()


def _bn(sd, key, y, training):
    if training:
        return F.batch_norm(y, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"], True, BN_MOMENTUM, BN_EPS)
    return F.batch_norm(y, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"], False, 0.0, BN_EPS)


def _cbs(sd, key, x, k=1, s=1, training=False):
    y = F.conv2d(_q(x), _q(sd[key + ".conv.weight"]), None, s, k // 2)
    return _q(F.silu(_bn(sd, key + ".bn", y, training)))


def _mcb_fwd(sd, p, x, ids, training):
    x1, x2 = _cbs(sd, p + ".cv1", x, training=training), _cbs(sd, p + ".cv2", x, training=training)
    xs = [x1, x2]
    for i in range(N):
        x2 = _cbs(sd, p + f".cv3.{i}", x2, 3, training=training)
        xs.append(x2)
    return _cbs(sd, p + ".cv4", torch.cat([xs[i] for i in ids], 1), training=training)


def _trans_fwd(sd, p, x, training):
    x1 = _cbs(sd, p + ".cv1", F.max_pool2d(x, 2, 2), training=training)
    x2 = _cbs(sd, p + ".cv3", _cbs(sd, p + ".cv2", x, training=training), 3, 2, training=training)
    return torch.cat([x2, x1], 1)


def _rep(sd, p, x, training):
    d = _bn(sd, p + ".rbr_dense.1", F.conv2d(_q(x), _q(sd[p + ".rbr_dense.0.weight"]), None, 1, 1), training)
    o = _bn(sd, p + ".rbr_1x1.1", F.conv2d(_q(x), _q(sd[p + ".rbr_1x1.0.weight"]), None, 1, 0), training)
    return _q(F.silu(_q(d) + o))                                  # (the engine stores the 3x3 branch in fp16 before the sum)


def forward(sd, x, training: bool = False):
    """Yolo7.forward (yolov7_model.py:472-525) -> (out0, out1, out2), NCHW."""
    t = training
    y = _cbs(sd, "backbone.stem.0", x, 3, 1, t)
    y = _cbs(sd, "backbone.stem.1", y, 3, 2, t)
    y = _cbs(sd, "backbone.stem.2", y, 3, 1, t)
    y = _mcb_fwd(sd, "backbone.dark2.1", _cbs(sd, "backbone.dark2.0", y, 3, 2, t), IDS_BACKBONE, t)
    feat1 = _mcb_fwd(sd, "backbone.dark3.1", _trans_fwd(sd, "backbone.dark3.0", y, t), IDS_BACKBONE, t)
    feat2 = _mcb_fwd(sd, "backbone.dark4.1", _trans_fwd(sd, "backbone.dark4.0", feat1, t), IDS_BACKBONE, t)
    feat3 = _mcb_fwd(sd, "backbone.dark5.1", _trans_fwd(sd, "backbone.dark5.0", feat2, t), IDS_BACKBONE, t)
    # SPPCSPC
    x1 = _cbs(sd, "sppcspc.cv4", _cbs(sd, "sppcspc.cv3", _cbs(sd, "sppcspc.cv1", feat3, training=t), 3, training=t), training=t)
    pools = [x1] + [F.max_pool2d(x1, k, 1, k // 2) for k in (5, 9, 13)]
    y1 = _cbs(sd, "sppcspc.cv6", _cbs(sd, "sppcspc.cv5", torch.cat(pools, 1), training=t), 3, training=t)
    P5 = _cbs(sd, "sppcspc.cv7", torch.cat((y1, _cbs(sd, "sppcspc.cv2", feat3, training=t)), 1), training=t)
    up = lambda v: F.interpolate(v, scale_factor=2.0, mode="nearest")  # noqa: E731
    P4 = torch.cat([_cbs(sd, "conv_for_feat2", feat2, training=t), up(_cbs(sd, "conv_for_P5", P5, training=t))], 1)
    P4 = _mcb_fwd(sd, "conv3_for_upsample1", P4, IDS_NECK, t)
    P3 = torch.cat([_cbs(sd, "conv_for_feat1", feat1, training=t), up(_cbs(sd, "conv_for_P4", P4, training=t))], 1)
    P3 = _mcb_fwd(sd, "conv3_for_upsample2", P3, IDS_NECK, t)
    P4 = _mcb_fwd(sd, "conv3_for_downsample1", torch.cat([_trans_fwd(sd, "down_sample1", P3, t), P4], 1), IDS_NECK, t)
    P5 = _mcb_fwd(sd, "conv3_for_downsample2", torch.cat([_trans_fwd(sd, "down_sample2", P4, t), P5], 1), IDS_NECK, t)
    outs = []
    for rep, head, feat in (("rep_conv_3", "yolo_head_P5", P5), ("rep_conv_2", "yolo_head_P4", P4), ("rep_conv_1", "yolo_head_P3", P3)):
        f = _rep(sd, rep, feat, t)
        outs.append(F.conv2d(_q(f), _q(sd[head + ".weight"]), sd[head + ".bias"]))
    return tuple(outs)                                            # (out0 = P5 head, out1, out2)


def decode(preds, nc: int, input_hw=(640, 640)):
    """YOLOv7.decode_box up to ``decoded_outputs`` (yolo_v7.py:246-343): (B, sum 3*H*W, 5 + nc), normalised cx, cy, w, h."""
    anchors = np.array(ANCHORS, dtype=np.float32).reshape(-1, 2)
    outs = []
    for i, pred in enumerate(preds):
        bs, _, h, w = pred.shape
        stride_h, stride_w = input_hw[0] / h, input_hw[1] / w
        sa = torch.tensor([(aw / stride_w, ah / stride_h) for aw, ah in anchors[list(ANCHORS_MASK[i])]], dtype=torch.float32)
        p = pred.reshape(bs, 3, 5 + nc, h, w).permute(0, 1, 3, 4, 2)
        sx, sy, sw, sh = (torch.sigmoid(p[..., k]) for k in range(4))
        gx = torch.linspace(0, w - 1, w).repeat(h, 1).repeat(bs * 3, 1, 1).view(sx.shape)
        gy = torch.linspace(0, h - 1, h).repeat(w, 1).t().repeat(bs * 3, 1, 1).view(sy.shape)
        aw = sa[:, 0:1].repeat(bs, 1).repeat(1, 1, h * w).view(sw.shape)
        ah = sa[:, 1:2].repeat(bs, 1).repeat(1, 1, h * w).view(sh.shape)
        boxes = torch.stack((sx * 2. - 0.5 + gx, sy * 2. - 0.5 + gy, (sw * 2) ** 2 * aw, (sh * 2) ** 2 * ah), -1)
        scale = torch.tensor([w, h, w, h], dtype=torch.float32)
        outs.append(torch.cat((boxes.reshape(bs, -1, 4) / scale, torch.sigmoid(p[..., 4]).reshape(bs, -1, 1),
                               torch.sigmoid(p[..., 5:]).reshape(bs, -1, nc)), -1))
    return torch.cat(outs, 1)


def nms(decoded, nc: int, conf_threshold: float, nms_threshold: float):
    """YOLOv7._nms before the letterbox inverse (yolo_v7.py:348-415): per image an (n, 7) array
    [x1, y1, x2, y2, obj_conf, class_conf, class_pred] (classes ascending, scores descending inside a class) and the indices of
    the kept rows of ``decoded``; None for an image without detections."""
    from . import nms_ref
    res = []
    for img in decoded:
        img = img.clone()
        xy, wh = img[:, 0:2].clone(), img[:, 2:4].clone()
        img[:, 0:2], img[:, 2:4] = xy - wh / 2, xy + wh / 2      # xywh_to_xyxy_torch(more=True), core/utils/bboxes.py
        cconf, cpred = torch.max(img[:, 5:5 + nc], 1)
        mask = img[:, 4] * cconf >= conf_threshold
        idx = torch.nonzero(mask).flatten()
        if idx.numel() == 0:
            res.append((None, None))
            continue
        det = torch.cat((img[idx, :5], cconf[idx, None], cpred[idx, None].float()), 1)
        rows, keep_idx = [], []
        for c in det[:, -1].unique():
            sel = torch.nonzero(det[:, -1] == c).flatten()
            sc = (det[sel, 4] * det[sel, 5]).numpy()
            order = np.argsort(-sc, kind="stable")                # descending score, ties: lower index first
            k = order[nms_ref._greedy(det[sel, :4].numpy()[order], None, nms_threshold)]
            rows.append(det[sel][k])
            keep_idx.append(idx[sel][k])
        res.append((torch.cat(rows).numpy(), torch.cat(keep_idx).numpy()))
    return res


def projection_weights(shapes, seed: int = 9):
    """Fixed N(0,1) tensors of the three output shapes: the linear functional the backward parity runs on."""
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*sh, generator=g) for sh in shapes]


def projection_loss(outs, weights):
    """sum_l mean(out_l * w_l): every output element receives the gradient w_l / numel_l."""
    return sum((o * w).mean() for o, w in zip(outs, weights))


def loss_and_grads(sd, x, weights=None, seed: int = 9):
    """Train-mode forward + backward of ``projection_loss``: (loss, {key: grad}, outs); running statistics in ``sd`` are updated."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()
              if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var"))}
    work = dict(sd)
    work.update(params)
    outs = forward(work, x, training=True)
    if weights is None:
        weights = projection_weights([o.shape for o in outs], seed)
    loss = projection_loss(outs, weights)
    loss.backward()
    return loss.detach(), {k: p.grad for k, p in params.items() if p.grad is not None}, [o.detach() for o in outs]


# ----------------------------------------------------------------------------------------------------------------------------------------
# Yolo7Loss (core/loss/yolo7_loss.py:14-444): candidate generation ("find_3_positive"), SimOTA assignment per image ("build_targets"),
# CIoU / objectness / class terms.  Restated with explicit loops over candidates; pinned against the real class in make_golden.py (11c).
STRIDES_LOSS = (32, 16, 8)
BALANCE = (0.4, 1.0, 4.0)
OFFSETS = ((0.0, 0.0), (0.5, 0.0), (0.0, 0.5), (-0.5, 0.0), (0.0, -0.5))


def _level_anchors(level: int):
    """anchors of a level in grid units, float32 (find_3_positive: torch.from_numpy(anchors[mask] / stride).type_as(pred))."""
    a = np.asarray(ANCHORS, dtype=np.float64).reshape(-1, 2)[list(ANCHORS_MASK[level])] / STRIDES_LOSS[level]
    return torch.from_numpy(a).float()


def loss_candidates(targets, level_hw, threshold: float = 4.0):
    """find_3_positive (:340-398): per level the ordered candidate list [(image, anchor, gj, gi, target row)] -- for each of the five cell
    offsets in turn, every (anchor, target) pair whose size ratio to the anchor is below `threshold` and whose centre lies in the half of
    its cell that faces the neighbour (and more than one cell away from the border on that side)."""
    out = []
    N = targets.shape[0]
    for lv, (h, w) in enumerate(level_hw):
        anc = _level_anchors(lv)
        gain = torch.tensor([w, h, w, h], dtype=torch.float32)
        t = targets[:, 2:6].float() * gain                                    # gx, gy, gw, gh in grid units
        keep = []
        for a in range(3):
            r = t[:, 2:4] / anc[a]
            ok = torch.max(r, 1.0 / r).max(1)[0] < threshold
            keep += [(a, n) for n in range(N) if bool(ok[n])]
        cands = []
        for oi, (ox, oy) in enumerate(OFFSETS):
            for a, n in keep:
                gx, gy = t[n, 0], t[n, 1]
                ix, iy = gain[0] - gx, gain[1] - gy
                cond = (True, bool((gx % 1.0 < 0.5) & (gx > 1.0)), bool((gy % 1.0 < 0.5) & (gy > 1.0)), bool((ix % 1.0 < 0.5) & (ix > 1.0)),
                        bool((iy % 1.0 < 0.5) & (iy > 1.0)))[oi]
                if not cond:
                    continue
                gi = min(max(int((gx - ox).to(torch.int64)), 0), w - 1)
                gj = min(max(int((gy - oy).to(torch.int64)), 0), h - 1)
                cands.append((int(targets[n, 0]), a, gj, gi, n))
        out.append(cands)
    return out


def simota_assign(preds, targets, cands, img_size: float, nc: int):
    """build_targets (:129-338): per image, among its candidates of all levels, the SimOTA assignment.  Returns per level the ordered list of
    matched (image, anchor, gj, gi, target row)."""
    matched = [[] for _ in preds]
    B = preds[0].shape[0]
    for b in range(B):
        rows = [n for n in range(targets.shape[0]) if int(targets[n, 0]) == b]
        if not rows:
            continue
        tgt = targets[rows]
        txywh = tgt[:, 2:6] * img_size                                        # imgs[b].shape[1] for all four coordinates
        txyxy = torch.cat((txywh[:, :2] - txywh[:, 2:] / 2, txywh[:, :2] + txywh[:, 2:] / 2), 1)
        entries, boxes, p_obj, p_cls = [], [], [], []
        for lv, p in enumerate(preds):
            anc = _level_anchors(lv)
            for (cb, a, gj, gi, n) in cands[lv]:
                if cb != b:
                    continue
                v = p[b, a, gj, gi]
                grid = torch.tensor([gi, gj], dtype=v.dtype)
                pxy = (v[:2].sigmoid() * 2.0 - 0.5 + grid) * STRIDES_LOSS[lv]
                pwh = (v[2:4].sigmoid() * 2) ** 2 * anc[a] * STRIDES_LOSS[lv]
                boxes.append(torch.cat((pxy - pwh / 2, pxy + pwh / 2)))
                p_obj.append(v[4:5])
                p_cls.append(v[5:])
                entries.append((lv, cb, a, gj, gi))
        if not entries:
            continue
        boxes, p_obj, p_cls = torch.stack(boxes), torch.stack(p_obj), torch.stack(p_cls)
        G, C = txyxy.shape[0], boxes.shape[0]
        lt, rb = torch.max(txyxy[:, None, :2], boxes[None, :, :2]), torch.min(txyxy[:, None, 2:], boxes[None, :, 2:])
        inter = (rb - lt).clamp(min=0).prod(2)
        area_t = ((txyxy[:, 2] - txyxy[:, 0]) * (txyxy[:, 3] - txyxy[:, 1]))[:, None]
        area_p = ((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))[None]
        iou = inter / (area_t + area_p - inter)
        topv, _ = torch.topk(iou, min(20, C), dim=1)
        dyn_k = torch.clamp(topv.sum(1).int(), min=1)
        onehot = F.one_hot(tgt[:, 1].long(), nc).float()[:, None, :].expand(G, C, nc)
        y = (p_cls.float().sigmoid()[None] * p_obj.sigmoid()[None]).expand(G, C, nc).sqrt()
        cls_cost = F.binary_cross_entropy_with_logits(torch.log(y / (1 - y)), onehot, reduction="none").sum(-1)
        cost = cls_cost + 3.0 * (-torch.log(iou + 1e-8))
        match = torch.zeros_like(cost)
        for g in range(G):
            _, pos = torch.topk(cost[g], k=int(dyn_k[g]), largest=False)
            match[g][pos] = 1.0
        multi = match.sum(0) > 1
        if int(multi.sum()) > 0:
            amin = torch.min(cost[:, multi], dim=0)[1]
            match[:, multi] *= 0.0
            match[amin, multi] = 1.0
        fg = match.sum(0) > 0
        gt_of = match[:, fg].argmax(0)
        k = 0
        for c in range(C):
            if bool(fg[c]):
                lv, cb, a, gj, gi = entries[c]
                matched[lv].append((cb, a, gj, gi, rows[int(gt_of[k])]))
                k += 1
    return matched


def _ciou_xywh(box, tbox, eps: float = 1e-7):
    """yolo7_bbox_iou(box.T, tbox, x1y1x2y2=False, CIoU=True) (core/utils/iou.py:184-218); rows of (cx, cy, w, h)."""
    b1x1, b1x2, b1y1, b1y2 = box[:, 0] - box[:, 2] / 2, box[:, 0] + box[:, 2] / 2, box[:, 1] - box[:, 3] / 2, box[:, 1] + box[:, 3] / 2
    b2x1, b2x2, b2y1, b2y2 = tbox[:, 0] - tbox[:, 2] / 2, tbox[:, 0] + tbox[:, 2] / 2, tbox[:, 1] - tbox[:, 3] / 2, tbox[:, 1] + tbox[:, 3] / 2
    inter = (torch.min(b1x2, b2x2) - torch.max(b1x1, b2x1)).clamp(0) * (torch.min(b1y2, b2y2) - torch.max(b1y1, b2y1)).clamp(0)
    w1, h1, w2, h2 = b1x2 - b1x1, b1y2 - b1y1 + eps, b2x2 - b2x1, b2y2 - b2y1 + eps
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw, ch = torch.max(b1x2, b2x2) - torch.min(b1x1, b2x1), torch.max(b1y2, b2y2) - torch.min(b1y1, b2y1)
    c2 = cw ** 2 + ch ** 2 + eps
    rho2 = ((b2x1 + b2x2 - b1x1 - b1x2) ** 2 + (b2y1 + b2y2 - b1y1 - b1y2) ** 2) / 4
    v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def yolo7_loss(outs, targets, img_size: float, nc: int, input_hw=(640, 640), label_smoothing: float = 0.0):
    """Yolo7Loss.__call__ (:38-127): outs = the three (B, 3*(5+nc), h, w) maps, coarsest first; targets (N, 6) [image, class, cx, cy, w, h]
    normalised; img_size = imgs[b].shape[1].  Returns (total, box, obj, cls) with the reference's ratios applied."""
    preds = [o.reshape(o.shape[0], 3, -1, o.shape[2], o.shape[3]).permute(0, 1, 3, 4, 2) for o in outs]
    level_hw = [(p.shape[2], p.shape[3]) for p in preds]
    with torch.no_grad():
        cands = loss_candidates(targets, level_hw)
        matched = simota_assign([p.detach() for p in preds], targets, cands, img_size, nc)
    cp, cn = 1.0 - 0.5 * label_smoothing, 0.5 * label_smoothing
    box_loss, obj_loss, cls_loss = torch.zeros(1), torch.zeros(1), torch.zeros(1)
    for lv, p in enumerate(preds):
        h, w = level_hw[lv]
        tobj = torch.zeros_like(p[..., 0])
        m = matched[lv]
        if m:
            b, a, gj, gi = (torch.tensor([e[k] for e in m]) for k in range(4))
            tt = targets[[e[4] for e in m]]
            pp = p[b, a, gj, gi]
            anc = _level_anchors(lv)[a]
            xy = pp[:, :2].sigmoid() * 2.0 - 0.5
            wh = (pp[:, 2:4].sigmoid() * 2) ** 2 * anc
            tbox = tt[:, 2:6] * torch.tensor([w, h, w, h], dtype=pp.dtype)
            tbox = torch.cat((tbox[:, :2] - torch.stack([gi, gj], 1).to(pp.dtype), tbox[:, 2:]), 1)
            iou = _ciou_xywh(torch.cat((xy, wh), 1), tbox)
            box_loss = box_loss + (1.0 - iou).mean()
            vals = iou.detach().clamp(0).to(tobj.dtype)
            for k in range(len(m)):                                           # duplicates: the last entry of the list wins (index_put on the CPU)
                tobj[b[k], a[k], gj[k], gi[k]] = vals[k]
            t = torch.full_like(pp[:, 5:], cn)
            t[range(len(m)), tt[:, 1].long()] = cp
            cls_loss = cls_loss + F.binary_cross_entropy_with_logits(pp[:, 5:], t)
        obj_loss = obj_loss + F.binary_cross_entropy_with_logits(p[..., 4], tobj) * BALANCE[lv]
    box_loss = box_loss * 0.05
    obj_loss = obj_loss * (1.0 * (input_hw[0] * input_hw[1]) / (640 ** 2))
    cls_loss = cls_loss * (0.5 * (nc / 80))
    return box_loss + obj_loss + cls_loss, box_loss, obj_loss, cls_loss
