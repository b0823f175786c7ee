"""CPU oracle for DeepLabv3+ (ResNet-101, output stride 16), inference and one training step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

SURVEY.md section 8 row a19 / (f)2.  A torch-CPU fp32 restatement, functional over a flat ``state_dict`` with the
reference's keys (``backbone.conv1...``, ``backbone.layerN.M...``, ``classifier.project...``, ``classifier.aspp...``,
``classifier.classifier...``), of

* the dilated ResNet-101 backbone (core/models/resnet.py:82-277, ``replace_stride_with_dilation=[False, False, True]``):
  7x7/2 stem, 3x3/2 max pool, 3 + 4 + 23 + 3 Bottlenecks (stride on the 3x3, resnet.py:110), layer4 at stride 1 with
  dilation 1 (first block) / 2, features "low_level" (layer1) and "out" (layer4);
* the DeepLabv3+ head (core/models/deeplabv3plus.py:10-149): ASPP (1x1, three atrous 3x3 at rates 6 / 12 / 18, image
  pooling; concat; 1x1 projection; Dropout = identity in eval), decoder (48-channel low-level projection, bilinear resize of
  the ASPP output, concat, 3x3, 1x1 + bias) and the final bilinear resize to the input size (``align_corners=False``).

Training (``forward(train=True)``, ``focal_loss``, ``loss_and_grads``): the same graph with batch-statistics BatchNorm
(running statistics updated in place, momentum 0.1, unbiased variance), ``Dropout(0.1)`` after the ASPP projection as a
caller-supplied keep mask (deeplabv3plus.py:67), FocalLoss (core/loss/focal_loss.py:6-22) / nn.CrossEntropyLoss
(core/algorithms/segmentation_2d.py:59-64); the gradients are torch autograd's over this restatement.

Parity pin: ``oracle/make_golden.py`` imports the real reference in the build container and asserts that
``init_state_dict`` reproduces ``DeeplabV3Plus(21, 16, pretrained_backbone=False)`` under seed 0 bit for bit, that
``forward`` matches its eval-mode output to fp32 round-off, and that ``loss_and_grads`` matches the loss and every parameter
gradient of one reference training step (model.train(), FocalLoss(), loss.backward()), then writes ``tests/golden/deeplab_*``.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

LAYERS = (3, 4, 23, 3)
PLANES = (64, 128, 256, 512)
ASPP_RATES = (6, 12, 18)
BN_EPS = 1e-5


def blocks():
    """ResNet._make_layer (resnet.py:190-232) as data, module order."""
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


def _conv_specs(nc):
    """(key, cout, cin, k, bias) in MODULE order (= state_dict order)."""
    specs = [("backbone.conv1", 64, 3, 7, False)]
    for b in blocks():
        p = b["prefix"]
        specs += [(p + ".conv1", b["width"], b["cin"], 1, False), (p + ".conv2", b["width"], b["width"], 3, False),
                  (p + ".conv3", b["cout"], b["width"], 1, False)]
        if b["down"]:
            specs.append((p + ".downsample.0", b["cout"], b["cin"], 1, False))
    c = "classifier."
    specs += [(c + "project.0", 48, 256, 1, False), (c + "aspp.convs.0.0", 256, 2048, 1, False)]
    specs += [(c + f"aspp.convs.{i + 1}.0", 256, 2048, 3, False) for i in range(3)]
    specs += [(c + "aspp.convs.4.1", 256, 2048, 1, False), (c + "aspp.project.0", 256, 1280, 1, False), (c + "classifier.0", 256, 304, 3, False),
              (c + "classifier.3", nc, 256, 1, True)]
    return specs


def _bn_key(conv_key):
    """the BatchNorm that follows a conv: convN -> bnN, <seq>.K -> <seq>.K+1"""
    stem, leaf = conv_key.rsplit(".", 1)
    return stem + ".bn" + leaf[4:] if leaf.startswith("conv") else stem + "." + str(int(leaf) + 1)


def init_state_dict(nc: int = 21, seed: int = 0):
    """The reference's construction + re-initialisation sequence on the global RNG (see the product's
    ``DeepLabV3PlusR101._init_like_reference`` for the order; resnet.py:150-178, deeplabv3plus.py:99-110)."""
    torch.manual_seed(seed)
    specs = {s[0]: s for s in _conv_specs(nc)}
    w, bias = {}, {}

    def construct(key):
        _, cout, cin, k, has_bias = specs[key]
        t = torch.empty(cout, cin, k, k)
        torch.nn.init.kaiming_uniform_(t, a=math.sqrt(5))
        w[key] = t
        if has_bias:
            bb = torch.empty(cout)
            bound = 1.0 / math.sqrt(cin * k * k)
            torch.nn.init.uniform_(bb, -bound, bound)
            bias[key] = bb

    order_b = ["backbone.conv1"]
    for b in blocks():
        p = b["prefix"]
        order_b += ([p + ".downsample.0"] if b["down"] else []) + [p + ".conv1", p + ".conv2", p + ".conv3"]
    for key in order_b:
        construct(key)
    for key in [s[0] for s in _conv_specs(nc) if s[0].startswith("backbone")]:
        torch.nn.init.kaiming_normal_(w[key], mode="fan_out", nonlinearity="relu")
    head = [s[0] for s in _conv_specs(nc) if s[0].startswith("classifier")]
    for key in head:                                             # construction order == module order in the head
        construct(key)
    for key in head:
        torch.nn.init.kaiming_normal_(w[key])
    sd = OrderedDict()
    for key, cout, cin, k, has_bias in _conv_specs(nc):
        sd[key + ".weight"] = w[key]
        if has_bias:
            sd[key + ".bias"] = bias[key]
            continue
        bk = _bn_key(key)
        sd[bk + ".weight"], sd[bk + ".bias"] = torch.ones(cout), torch.zeros(cout)
        sd[bk + ".running_mean"], sd[bk + ".running_var"] = torch.zeros(cout), torch.ones(cout)
        sd[bk + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    return sd


# fp16-STORAGE emulation (as in oracle/yolov8_ref.py, oracle/centernet_ref.py): with FP16_STORAGE[0] = True every convolution
# sees its input and its weights rounded to fp16 (the engine's MFMA operands) and every activation tensor is stored in fp16.
FP16_STORAGE = [False]


def _q(t):
    if not FP16_STORAGE[0]:
        return t
    if t.requires_grad:                       # straight-through: the value is rounded, the gradient passes unrounded
        return t + (t.detach().half().float() - t.detach())
    return t.half().float()


TRAIN = [False]                               # set by forward(train=True): batch statistics + running-statistics update


def _conv_bn(sd, key, x, stride=1, dil=1, act=True, res=None):
    w = sd[key + ".weight"]
    k = w.shape[-1]
    y = F.conv2d(_q(x), _q(w), None, stride, dil * (k // 2), dil)
    bk = _bn_key(key)
    y = F.batch_norm(y, sd[bk + ".running_mean"], sd[bk + ".running_var"], sd[bk + ".weight"], sd[bk + ".bias"], TRAIN[0], 0.1, BN_EPS)
    if res is not None:
        y = y + res
    return _q(F.relu(y) if act else y)


def forward(sd, x, nc: int = 21, return_rows: bool = False, train: bool = False, keep_mask=None, dropout_p: float = 0.1):
    """(B,3,H,W) fp32 -> (B,nc,H,W) fp32 logits (deeplabv3plus.py:142-148).  ``return_rows``: also the logits at the decoder's
    resolution, NHWC (B, h, w, nc) -- what the engine's last convolution writes.  ``train``: batch-statistics BatchNorm (the
    running statistics in ``sd`` are updated in place) and dropout after the ASPP projection with ``keep_mask`` (B, 256, h, w)
    of 0/1 -- None = no dropout (p = 0)."""
    TRAIN[0] = bool(train)
    try:
        return _forward(sd, x, nc, return_rows, keep_mask if train else None, dropout_p)
    finally:
        TRAIN[0] = False


def _forward(sd, x, nc, return_rows, keep_mask, dropout_p):
    H, W = x.shape[-2:]
    y = _conv_bn(sd, "backbone.conv1", x, stride=2)
    y = F.max_pool2d(y, 3, 2, 1)
    low = None
    for b in blocks():
        p = b["prefix"]
        t = _conv_bn(sd, p + ".conv1", y)
        t = _conv_bn(sd, p + ".conv2", t, stride=b["stride"], dil=b["dil"])
        ident = _conv_bn(sd, p + ".downsample.0", y, stride=b["stride"], act=False) if b["down"] else y
        y = _conv_bn(sd, p + ".conv3", t, res=ident)
        if b["layer"] == 1:
            low = y
    c = "classifier."
    br = [_conv_bn(sd, c + "aspp.convs.0.0", y)]
    br += [_conv_bn(sd, c + f"aspp.convs.{i + 1}.0", y, dil=r) for i, r in enumerate(ASPP_RATES)]
    pooled = _q(F.adaptive_avg_pool2d(y, 1))
    pooled = _conv_bn(sd, c + "aspp.convs.4.1", pooled)
    br.append(_q(F.interpolate(pooled, size=y.shape[-2:], mode="bilinear", align_corners=False)))
    a = _conv_bn(sd, c + "aspp.project.0", torch.cat(br, 1))
    if keep_mask is not None:                                    # nn.Dropout(0.1) in training mode (deeplabv3plus.py:67)
        a = _q(a * keep_mask / (1.0 - dropout_p))
    lowp = _conv_bn(sd, c + "project.0", low)
    a = _q(F.interpolate(a, size=lowp.shape[-2:], mode="bilinear", align_corners=False))
    h = _conv_bn(sd, c + "classifier.0", torch.cat([lowp, a], 1))
    logits = F.conv2d(_q(h), _q(sd[c + "classifier.3.weight"]), sd[c + "classifier.3.bias"])
    out = F.interpolate(logits, size=(H, W), mode="bilinear", align_corners=False)
    return (out, logits.permute(0, 2, 3, 1).contiguous()) if return_rows else out


def focal_loss(logits, target, alpha: float = 0.25, gamma: float = 2.0, ignore_index: int = -100):
    """FocalLoss.forward (core/loss/focal_loss.py:14-22), size_average=True."""
    ce = F.cross_entropy(logits, target, ignore_index=ignore_index, reduction="none")
    pt = torch.exp(-ce)
    return (alpha * (1 - pt) ** gamma * ce).mean()


def loss_and_grads(sd, x, target, nc: int = 21, loss_type: str = "focal", keep_mask=None, dropout_p: float = 0.1):
    """One training step's loss and parameter gradients (segmentation_trainer.py:121-130 without the optimiser): returns
    (loss, {key: grad}, rows (B, h, w, nc)); running statistics in ``sd`` are updated in place."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()
              if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var"))}
    work = dict(sd)
    work.update(params)
    out, rows = forward(work, x, nc, return_rows=True, train=True, keep_mask=keep_mask, dropout_p=dropout_p)
    loss = focal_loss(out, target) if loss_type == "focal" else F.cross_entropy(out, target, reduction="mean")
    loss.backward()
    return loss.detach(), {k: p.grad for k, p in params.items() if p.grad is not None}, rows.detach()
