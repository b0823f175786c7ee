"""Pin the oracle against the REAL reference and write tests/golden/*.npz  (run in the build container only).

    python oracle/make_golden.py            # needs /root/reference; never runs on the GPU box

What it does:
1. installs inert ``sys.modules`` stubs for the third-party imports the image lacks (thop, cv2,
   torchvision, pycocotools, tensorboard) -- none of them takes part in model / loss arithmetic;
2. imports the reference through its own plugin API (``builder.export_from_registry``);
3. asserts ``oracle.yolov8_ref`` == reference: init weights bit-exact, train/eval forward, v8 loss,
   every parameter gradient, and two Adam steps, to fp32 round-off;
4. writes small fixtures (inputs + expected outputs, data only) that ``tests/test_oracle_golden.py``
   re-checks against the oracle wherever the reference is absent.

The NMS fixture is produced by ``oracle.nms_ref`` itself (upstream parity unpinned -- see its header).
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("CVX_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")


def _install_stubs():
    class _Inert:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return _Inert()

        def __getattr__(self, n):
            if n.startswith("__"):
                raise AttributeError(n)
            return _Inert()

    def _mod(name):
        m = types.ModuleType(name)

        def _ga(n):
            if n.startswith("__"):
                raise AttributeError(n)
            return _Inert()

        m.__getattr__ = _ga
        m.__path__ = []
        sys.modules[name] = m

    for n in ("thop", "cv2", "torchvision", "torchvision.ops", "torchvision.transforms",
              "torchvision.transforms.functional", "torchvision.models", "pycocotools", "pycocotools.coco",
              "pycocotools.cocoeval", "tensorboard", "torch.utils.tensorboard"):
        if n not in sys.modules:
            _mod(n)


def _import_reference():
    """Returns the reference's (cfg, algorithm instance) for yolo8_det with the repo's own same-named
    modules kept out of the way."""
    sys.dont_write_bytecode = True
    _install_stubs()
    for k in [k for k in sys.modules if k.split(".")[0] in ("builder", "registry", "check", "configs", "core")]:
        del sys.modules[k]
    sys.path = [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
    sys.path.insert(0, REF)
    import builder  # noqa: the reference's
    assert os.path.abspath(builder.__file__).startswith(REF), builder.__file__
    return builder


def deeplab_train_section(builder, report):
    """10b. DeepLabv3+ training step (BASELINE configs[5]): the REAL reference model in train mode, FocalLoss(), loss.backward()
    (core/trainer/segmentation_trainer.py:121-130 without the optimiser) -> loss, every parameter gradient, updated running
    statistics.  Asserts the oracle's restatement (deeplab_ref.loss_and_grads) against all of them, then writes the fixture:
    inputs, loss, logits rows, per-tensor gradient norms / sums, a handful of gradient tensors in full."""
    from oracle import deeplab_ref as D
    dcfg, dalgo_cls, _ = builder.export_from_registry("deeplabv3plus")
    torch.manual_seed(0)
    algo = dalgo_cls(dcfg, torch.device("cpu"))
    dmodel, _ = algo.build_model()
    crit = algo.build_loss()
    assert type(crit).__name__ == "FocalLoss"
    GAMMA3 = 0.1                                                 # conditioning of section 10 (unit bn3 weights make a random-init R101 chaotic)
    with torch.no_grad():
        for k_, v_ in dmodel.named_parameters():
            if k_.endswith(".bn3.weight"):
                v_.fill_(GAMMA3)
    sd0 = {k: v.clone() for k, v in dmodel.state_dict().items()}
    B, H, W = 2, 97, 129
    g = torch.Generator().manual_seed(77)
    x = torch.rand(B, 3, H, W, generator=g)
    t = torch.randint(0, dcfg.dataset.num_classes, (B, H, W), generator=g)
    t[torch.rand(B, H, W, generator=g) < 0.1] = -100             # FocalLoss's ignore_index
    dmodel.train()
    drop = [m_ for m_ in dmodel.modules() if isinstance(m_, torch.nn.Dropout)]
    assert len(drop) == 1 and drop[0].p == 0.1
    drop[0].p = 0.0                                              # the mask is a draw of the caller's RNG: the fixture runs without it
    rows_ref = []
    hook = dmodel.classifier.classifier.register_forward_hook(lambda m_, i_, o_: rows_ref.append(o_.detach()))
    loss = crit(dmodel(x.clone()), t)
    hook.remove()
    loss.backward()
    ref_grads = {k: p.grad.clone() for k, p in dmodel.named_parameters()}
    ref_sd = {k: v.clone() for k, v in dmodel.state_dict().items()}
    my_loss, my_grads, my_rows = D.loss_and_grads({k: v.clone() for k, v in sd0.items()}, x.clone(), t, dcfg.dataset.num_classes)
    assert abs(float(my_loss) - float(loss)) < 1e-6 * abs(float(loss)), (float(my_loss), float(loss))
    assert torch.allclose(my_rows, rows_ref[0].permute(0, 2, 3, 1), rtol=1e-4, atol=1e-5)
    worst = 0.0
    for k, gr in ref_grads.items():
        e = float((my_grads[k] - gr).norm() / gr.norm().clamp_min(1e-30))
        worst = max(worst, e)
        assert e < 1e-4, (k, e)
    # a second pass with dropout ON and the reference's own mask captured: pins the oracle's keep_mask semantics
    drop[0].p = 0.1
    masks = []
    hk = drop[0].register_forward_hook(lambda m_, i_, o_: masks.append((o_ != 0) | (i_[0] == 0)))
    dmodel.load_state_dict(sd0)
    dmodel.zero_grad()
    torch.manual_seed(123)
    loss_d = crit(dmodel(x.clone()), t)
    hk.remove()
    loss_d.backward()
    keep = masks[0].float()
    my_loss_d, my_grads_d, _ = D.loss_and_grads({k: v.clone() for k, v in sd0.items()}, x.clone(), t, dcfg.dataset.num_classes, keep_mask=keep)
    assert abs(float(my_loss_d) - float(loss_d)) < 1e-6 * abs(float(loss_d))
    kd = "classifier.aspp.project.0.weight"
    assert float((my_grads_d[kd] - dmodel.state_dict(keep_vars=True)[kd].grad).norm() / my_grads_d[kd].norm()) < 1e-4
    full = ["classifier.classifier.3.weight", "classifier.classifier.3.bias", "classifier.classifier.1.weight", "classifier.aspp.convs.4.2.weight",
            "classifier.project.1.bias", "backbone.layer4.2.bn3.weight", "backbone.layer3.11.bn2.bias", "backbone.layer2.0.downsample.1.weight",
            "backbone.layer1.0.bn1.bias", "backbone.bn1.weight", "backbone.bn1.bias", "backbone.conv1.weight"]
    keys = list(ref_grads.keys())
    stat_keys = ["backbone.bn1.running_mean", "backbone.bn1.running_var", "backbone.layer4.2.bn3.running_var", "classifier.aspp.convs.4.2.running_mean",
                 "classifier.classifier.1.running_var"]
    np.savez_compressed(os.path.join(GOLD, "deeplab_train_97x129.npz"), x=x.numpy(), target=t.numpy().astype(np.int16), loss=np.array(float(loss)),
                        loss_dropout=np.array(float(loss_d)), keep_mask=np.packbits(keep.numpy().astype(np.uint8)), keep_shape=np.array(keep.shape),
                        rows=rows_ref[0].permute(0, 2, 3, 1).numpy().copy(), bn3_gamma=np.array(GAMMA3), grad_keys=np.array(keys),
                        grad_norm=np.array([float(ref_grads[k].double().norm()) for k in keys]),
                        grad_sum=np.array([float(ref_grads[k].double().sum()) for k in keys]),
                        stat_keys=np.array(stat_keys), **{"g:" + k: ref_grads[k].numpy() for k in full},
                        **{"s:" + k: ref_sd[k].numpy().copy() for k in stat_keys})
    report["deeplab_train"] = dict(loss=float(loss), loss_dropout=float(loss_d), worst_grad_rel_oracle_vs_reference=worst, tensors=len(keys),
                                   note="bn3.weight = 0.1; dropout p = 0 (second pass: p = 0.1 with the reference's captured mask)")


def yolov7_train_section(builder, report):
    """11b. YOLOv7-l network forward + backward in training mode: the REAL reference model (model.train()), a fixed linear functional
    of its three outputs (yolov7_ref.projection_loss; the reference's Yolo7Loss is torch code on these tensors and stays what it is),
    loss.backward() -> every parameter gradient and the updated running statistics.  Pins the oracle's train-mode restatement."""
    from oracle import yolov7_ref as Y7
    ycfg, yalgo_cls, _ = builder.export_from_registry("yolo7")
    ycfg.train.pretrained = False
    torch.manual_seed(0)
    ymodel, _ = yalgo_cls(ycfg, torch.device("cpu")).build_model()
    nc = ycfg.dataset.num_classes
    sd0 = {k: v.clone() for k, v in ymodel.state_dict().items()}
    my0 = Y7.init_state_dict(nc, seed=0)
    assert all(torch.equal(sd0[k], my0[k]) for k in sd0)
    B, H, W = 2, 160, 224
    g = torch.Generator().manual_seed(31)
    x = torch.rand(B, 3, H, W, generator=g)
    ymodel.train()
    outs = ymodel(x.clone())
    weights = Y7.projection_weights([o.shape for o in outs], seed=9)
    loss = Y7.projection_loss(outs, weights)
    loss.backward()
    ref_grads = {k: p.grad.clone() for k, p in ymodel.named_parameters()}
    ref_sd = {k: v.clone() for k, v in ymodel.state_dict().items()}
    work = {k: v.clone() for k, v in sd0.items()}
    my_loss, my_grads, my_outs = Y7.loss_and_grads(work, x.clone(), weights)
    assert abs(float(my_loss) - float(loss)) <= 1e-6 * max(abs(float(loss)), 1e-3), (float(my_loss), float(loss))
    for a_, b_ in zip(my_outs, outs):
        assert torch.allclose(a_, b_.detach(), rtol=1e-4, atol=1e-5)
    worst = 0.0
    for k, gr in ref_grads.items():
        e = float((my_grads[k] - gr).norm() / gr.norm().clamp_min(1e-30))
        worst = max(worst, e)
        assert e < 1e-4, (k, e)
    for k in ref_sd:
        if k.endswith(("running_mean", "running_var")):
            assert torch.allclose(work[k], ref_sd[k], rtol=1e-5, atol=1e-6), k
    keys = list(ref_grads.keys())
    full = ["yolo_head_P3.weight", "yolo_head_P5.bias", "rep_conv_1.rbr_dense.1.weight", "rep_conv_1.rbr_1x1.0.weight", "sppcspc.cv1.bn.weight",
            "down_sample1.cv1.conv.weight", "backbone.dark2.0.conv.weight", "backbone.stem.0.conv.weight", "backbone.stem.0.bn.bias"]
    full = [k for k in full if k in ref_grads]
    stat_keys = ["backbone.stem.0.bn.running_mean", "backbone.stem.0.bn.running_var", "rep_conv_2.rbr_1x1.1.running_var", "sppcspc.cv7.bn.running_mean"]
    stat_keys = [k for k in stat_keys if k in ref_sd]
    np.savez_compressed(os.path.join(GOLD, "yolov7_train_160x224.npz"), x=x.numpy(), proj_seed=np.array(9), loss=np.array(float(loss)),
                        out_sub=np.concatenate([o.detach().flatten()[::7].numpy() for o in outs]), grad_keys=np.array(keys),
                        grad_norm=np.array([float(ref_grads[k].double().norm()) for k in keys]),
                        grad_sum=np.array([float(ref_grads[k].double().sum()) for k in keys]), stat_keys=np.array(stat_keys),
                        **{"g:" + k: ref_grads[k].numpy() for k in full}, **{"s:" + k: ref_sd[k].numpy().copy() for k in stat_keys})
    report["yolov7_train"] = dict(loss=float(loss), worst_grad_rel_oracle_vs_reference=worst, tensors=len(keys), full=full, stats=stat_keys)


def centernet_train_section(builder, report):
    """9b. CenterNet DLA-34 network forward + backward in training mode: the REAL reference model (model.train()), a fixed linear
    functional of its output tensor (centernet_ref.projection_loss), loss.backward() -> every parameter gradient (incl. the depthwise
    transposed convolutions' weights) and the updated running statistics.  Pins the oracle's train-mode restatement."""
    from oracle import centernet_ref as C
    ccfg, calgo_cls, _ = builder.export_from_registry("centernet")
    torch.manual_seed(0)
    cmodel, _ = calgo_cls(ccfg, torch.device("cpu")).build_model()
    nc = ccfg.dataset.num_classes
    sd0 = {k: v.clone() for k, v in cmodel.state_dict().items()}
    B, H, W = 2, 128, 160
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(41))
    cmodel.train()
    out = cmodel(x.clone())
    weights = C.projection_weights(out.shape, seed=9)
    loss = C.projection_loss(out, weights)
    loss.backward()
    ref_grads = {k: p.grad.clone() for k, p in cmodel.named_parameters() if p.grad is not None}
    ref_sd = {k: v.clone() for k, v in cmodel.state_dict().items()}
    work = {k: v.clone() for k, v in sd0.items()}
    my_loss, my_grads, my_out = C.loss_and_grads(work, x.clone(), nc, weights)
    assert abs(float(my_loss) - float(loss)) <= 1e-5 * max(abs(float(loss)), 1e-3), (float(my_loss), float(loss))
    assert torch.allclose(my_out, out.detach(), rtol=1e-4, atol=1e-5)
    worst = 0.0
    for k, gr in ref_grads.items():
        e = float((my_grads[k] - gr).norm() / gr.norm().clamp_min(1e-30))
        worst = max(worst, e)
        assert e < 1e-4, (k, e)
    for k in ref_sd:
        if k.endswith(("running_mean", "running_var")):
            assert torch.allclose(work[k], ref_sd[k], rtol=1e-5, atol=1e-6), k
    keys = list(ref_grads.keys())
    full = [k for k in keys if (".up_" in k and k.endswith("weight"))][:3] + [k for k in ("backbone.heatmap_head.2.weight", "backbone.wh_head.0.bias",
            "backbone.reg_head.2.bias", "backbone.base.base_layer.0.weight", "backbone.base.base_layer.1.bias", "backbone.base.level_2.tree1.bn2.weight",
            "backbone.base.level_5.root.conv.weight") if k in ref_grads and ref_grads[k].numel() < 300000]
    stat_keys = [k for k in ("backbone.base.base_layer.1.running_mean", "backbone.base.base_layer.1.running_var", "backbone.base.level_5.root.bn.running_var")
                 if k in ref_sd]
    np.savez_compressed(os.path.join(GOLD, "centernet_train_128x160.npz"), x=x.numpy(), proj_seed=np.array(9), loss=np.array(float(loss)), nc=np.array(nc),
                        out_sub=out.detach().flatten()[::7].numpy().copy(), grad_keys=np.array(keys),
                        grad_norm=np.array([float(ref_grads[k].double().norm()) for k in keys]),
                        grad_sum=np.array([float(ref_grads[k].double().sum()) for k in keys]), stat_keys=np.array(stat_keys),
                        **{"g:" + k: ref_grads[k].numpy() for k in full}, **{"s:" + k: ref_sd[k].numpy().copy() for k in stat_keys})
    report["centernet_train"] = dict(loss=float(loss), worst_grad_rel_oracle_vs_reference=worst, tensors=len(keys), full=full, stats=stat_keys)


def ssd_train_section(builder, report):
    """12b. SSD300 VGG16-BN network forward + backward in training mode: the REAL reference model (model.train()), a fixed linear
    functional of (loc, conf) (ssd_ref.projection_loss), loss.backward() -> every parameter gradient (biases behind a BatchNorm: zero up
    to round-off; L2Normalize's weight) and the updated running statistics.  Pins the oracle's train-mode restatement."""
    from oracle import ssd_ref as S
    scfg, salgo_cls, _ = builder.export_from_registry("ssd")
    torch.manual_seed(0)
    smodel, _ = salgo_cls(scfg, torch.device("cpu")).build_model()
    nc = scfg.dataset.num_classes
    sd0 = {k: v.clone() for k, v in smodel.state_dict().items()}
    x = (torch.rand(2, 3, 300, 300, generator=torch.Generator().manual_seed(61)) * 255).round() / 255      # stored as bytes in the fixture
    smodel.train()
    outs = smodel(x.clone())
    weights = S.projection_weights([o.shape for o in outs], seed=9)
    loss = S.projection_loss(outs, weights)
    loss.backward()
    ref_grads = {k: p.grad.clone() for k, p in smodel.named_parameters() if p.grad is not None}
    ref_sd = {k: v.clone() for k, v in smodel.state_dict().items()}
    work = {k: v.clone() for k, v in sd0.items()}
    my_loss, my_grads, my_outs = S.loss_and_grads(work, x.clone(), nc, weights)
    assert abs(float(my_loss) - float(loss)) <= 1e-5 * max(abs(float(loss)), 1e-3), (float(my_loss), float(loss))
    for a_, b_ in zip(my_outs, outs):
        assert torch.allclose(a_, b_.detach(), rtol=1e-4, atol=1e-5)
    gmax = max(float(v.norm()) for v in ref_grads.values())
    worst = 0.0
    for k, gr in ref_grads.items():
        if float(gr.norm()) < 1e-6 * gmax:                        # conv biases in front of a BatchNorm: zero in exact arithmetic
            continue
        e = float((my_grads[k] - gr).norm() / gr.norm())
        worst = max(worst, e)
        assert e < 1e-3, (k, e)
    for k in ref_sd:
        if k.endswith(("running_mean", "running_var")):
            assert torch.allclose(work[k], ref_sd[k], rtol=1e-5, atol=1e-6), k
    keys = list(ref_grads.keys())
    full = [k for k in ("l2_norm.weight", "locs.0.bias", "confs.5.bias", "locs.5.weight", "extras.conv8.weight", "extras.conv1.bias", "backbone.layers.0.weight",
                        "backbone.layers.1.weight", "backbone.layers.1.bias", "backbone.layers.31.weight") if k in ref_grads]
    stat_keys = [k for k in ("backbone.layers.1.running_mean", "backbone.layers.1.running_var", "backbone.layers.31.running_mean") if k in ref_sd]
    np.savez_compressed(os.path.join(GOLD, "ssd_train_300.npz"), x=(x.numpy() * 255).round().astype(np.uint8), proj_seed=np.array(9),
                        loss=np.array(float(loss)), nc=np.array(nc), out_sub=np.concatenate([o.detach().flatten()[::7].numpy() for o in outs]),
                        grad_keys=np.array(keys), grad_norm=np.array([float(ref_grads[k].double().norm()) for k in keys]), stat_keys=np.array(stat_keys),
                        **{"g:" + k: ref_grads[k].numpy() for k in full}, **{"s:" + k: ref_sd[k].numpy().copy() for k in stat_keys})
    report["ssd_train"] = dict(loss=float(loss), worst_grad_rel_oracle_vs_reference=worst, tensors=len(keys), full=full, stats=stat_keys)


def centernet_loss_section(builder, report):
    """9c. CombinedLoss (core/loss/centernet_loss.py): the REAL reference loss object on random head outputs and seeded targets (two
    cases: objects present incl. two sharing a centre; no object at all) -> loss and its gradient w.r.t. the predictions by torch
    autograd.  Pins oracle/centernet_ref.combined_loss."""
    from oracle import centernet_ref as C
    ccfg, calgo_cls, _ = builder.export_from_registry("centernet")
    calgo = calgo_cls(ccfg, torch.device("cpu"))
    crit = calgo.build_loss()
    assert type(crit).__name__ == "CombinedLoss"
    nc = ccfg.dataset.num_classes
    out = {}
    for tag, B, h, w, K, seed, empty in (("a", 2, 24, 32, 30, 3, False), ("b", 1, 8, 8, 4, 5, True)):
        g = torch.Generator().manual_seed(100 + seed)
        pred = (torch.randn(B, h, w, nc + 4, generator=g) * 2).requires_grad_(True)
        targets = C.synth_targets(B, h, w, nc, K, seed)
        if empty:
            targets = [torch.zeros_like(targets[0]), targets[1], targets[2], torch.zeros_like(targets[3]), targets[4]]
        else:
            targets[4][0, 1] = targets[4][0, 0]                  # two objects on one centre
            targets[3][0, 1] = 1.0
        loss = crit(pred, targets)
        loss.backward()
        p2 = pred.detach().clone().requires_grad_(True)
        mine = C.combined_loss(p2, targets, nc, ccfg.loss.hm_weight, ccfg.loss.wh_weight, ccfg.loss.off_weight)
        mine[0].backward()
        assert abs(float(mine[0]) - float(loss)) <= 1e-6 * abs(float(loss)) and torch.allclose(p2.grad, pred.grad, rtol=1e-5, atol=1e-9)
        out.update({f"{tag}_pred": pred.detach().numpy(), f"{tag}_heat": targets[0].numpy(), f"{tag}_reg": targets[1].numpy(), f"{tag}_wh": targets[2].numpy(),
                    f"{tag}_mask": targets[3].numpy(), f"{tag}_idx": targets[4].numpy(), f"{tag}_loss": np.array(float(loss)),
                    f"{tag}_grad": pred.grad.numpy()})
    np.savez_compressed(os.path.join(GOLD, "centernet_loss.npz"), nc=np.array(nc), weights=np.array([ccfg.loss.hm_weight, ccfg.loss.wh_weight, ccfg.loss.off_weight]), **out)
    report["centernet_loss"] = dict(loss_a=float(out["a_loss"]), loss_b=float(out["b_loss"]), nc=int(nc))


def ssd_loss_section(builder, report):
    """12c. MultiBoxLossV2 (core/loss/multi_box_loss.py): the REAL reference loss object on random (loc, conf) and seeded encoded targets
    (case a: positives in every image; case b: no positive anywhere -> the 100-negatives branch) -> the three loss values and the
    gradient w.r.t. (loc, conf) by torch autograd.  Pins oracle/ssd_ref.multibox_loss."""
    from oracle import ssd_ref as S
    scfg, salgo_cls, _ = builder.export_from_registry("ssd")
    crit = salgo_cls(scfg, torch.device("cpu")).build_loss()
    assert type(crit).__name__ == "MultiBoxLossV2"
    nc = scfg.dataset.num_classes
    out = {}
    for tag, B, A, seed, empty in (("a", 3, 1200, 5, False), ("b", 2, 400, 6, True)):
        g = torch.Generator().manual_seed(200 + seed)
        loc = torch.randn(B, A, 4, generator=g).requires_grad_(True)
        conf = (torch.randn(B, A, nc + 1, generator=g) * 2).requires_grad_(True)
        y = S.synth_y_true(B, A, nc, 12, seed)
        if empty:
            y[:, :, :4], y[:, :, 4], y[:, :, 5:-1], y[:, :, -1] = 0.0, 1.0, 0.0, 0.0
        total, l_loss, c_loss = crit(y_true=y, y_pred=(loc, conf))
        total.backward()
        l2, c2 = loc.detach().clone().requires_grad_(True), conf.detach().clone().requires_grad_(True)
        mine = S.multibox_loss(y, l2, c2, scfg.loss.neg_pos, 0.5)
        mine[0].backward()
        assert abs(float(mine[0]) - float(total)) <= 1e-6 * abs(float(total)), (float(mine[0]), float(total))
        assert torch.allclose(l2.grad, loc.grad, rtol=1e-5, atol=1e-9) and torch.allclose(c2.grad, conf.grad, rtol=1e-5, atol=1e-9)
        out.update({f"{tag}_loc": loc.detach().numpy(), f"{tag}_conf": conf.detach().numpy(), f"{tag}_y": y.numpy(),
                    f"{tag}_items": np.array([float(total), float(l_loss), float(c_loss)]), f"{tag}_dloc": loc.grad.numpy(), f"{tag}_dconf": conf.grad.numpy()})
    np.savez_compressed(os.path.join(GOLD, "ssd_loss.npz"), nc=np.array(nc), neg_pos=np.array(float(scfg.loss.neg_pos)), **out)
    report["ssd_loss"] = dict(items_a=[float(v) for v in out["a_items"]], items_b=[float(v) for v in out["b_items"]], nc=int(nc))


def ssd_targets_section(builder, report):
    """12d. Ssd.generate_targets (core/algorithms/ssd.py:327-480): the REAL reference method on seeded label sets -- ordinary boxes, a box no
    prior overlaps above the threshold (forced best prior), two boxes competing for the same priors, no box at all.  Pins
    oracle/ssd_ref.generate_targets (exact) and is the fixture of the device kernel."""
    from oracle import ssd_ref as S
    scfg, salgo_cls, _ = builder.export_from_registry("ssd")
    algo = salgo_cls(scfg, torch.device("cpu"))
    nc = scfg.dataset.num_classes
    g = torch.Generator().manual_seed(77)
    cases = []
    for n in (5, 1, 12, 0):
        lab = np.zeros((n, 6), dtype=np.float32)
        if n:
            lab[:, 1] = torch.randint(0, nc, (n,), generator=g).numpy()
            lab[:, 2:4] = (torch.rand(n, 2, generator=g) * 0.8 + 0.1).numpy()
            lab[:, 4:6] = (torch.rand(n, 2, generator=g) * 0.5 + 0.02).numpy()
        cases.append(lab)
    cases[1][0, 4:6] = (0.004, 0.9)                              # a sliver: nothing above the threshold -> the best prior is forced
    cases[2][1] = cases[2][0]                                    # two identical boxes of different class: the first wins the ties
    cases[2][1, 1] = (cases[2][0, 1] + 1) % nc
    outs = []
    for lab in cases:
        ref = algo.generate_targets(lab.copy()).numpy()
        mine = S.generate_targets(lab.copy(), algo.anchors, nc, scfg.loss.overlap_threshold, algo.variance)
        assert ref.dtype == np.float32 and np.array_equal(ref, mine), float(np.abs(ref - mine).max())
        outs.append(ref)
    nmax = max(len(c) for c in cases)
    labels = np.zeros((len(cases), nmax, 5), dtype=np.float32)
    for i, c in enumerate(cases):
        labels[i, :len(c)] = c[:, 1:]
    np.savez_compressed(os.path.join(GOLD, "ssd_targets.npz"), labels=labels, counts=np.array([len(c) for c in cases], dtype=np.int32),
                        y_true=np.stack(outs), anchors=algo.anchors.astype(np.float32), nc=np.array(nc), thr=np.array(scfg.loss.overlap_threshold),
                        variance=np.asarray(algo.variance, dtype=np.float32))
    report["ssd_targets"] = dict(positives=[int(o[:, -1].sum()) for o in outs])


def centernet_targets_section(builder, report):
    """9d. CenterNet.generate_targets (core/algorithms/centernet.py:66-112): the REAL reference method on seeded label sets -- ordinary
    boxes, overlapping boxes of one class (maximum merge), a box at the map's border (clipped Gaussian), a tiny box (radius 0), no box.
    Pins oracle/centernet_ref.generate_targets (exact) and is the fixture of the device kernel."""
    from oracle import centernet_ref as C
    ccfg, calgo_cls, _ = builder.export_from_registry("centernet")
    algo = calgo_cls(ccfg, torch.device("cpu"))
    nc, K = ccfg.dataset.num_classes, ccfg.train.max_num_boxes
    fh, fw = algo.feature_size
    g = torch.Generator().manual_seed(91)
    cases = []
    for n in (6, 3, 2, 0):
        lab = np.zeros((n, 6), dtype=np.float32)
        if n:
            lab[:, 1] = torch.randint(0, nc, (n,), generator=g).numpy()
            lab[:, 2:4] = (torch.rand(n, 2, generator=g) * 0.8 + 0.1).numpy()
            lab[:, 4:6] = (torch.rand(n, 2, generator=g) * 0.4 + 0.03).numpy()
        cases.append(lab)
    cases[1][1, 1] = cases[1][0, 1]                               # same class, overlapping
    cases[1][1, 2:4] = cases[1][0, 2:4] + 0.03
    cases[2][0, 2:6] = (0.99, 0.02, 0.3, 0.25)                    # at the corner: the Gaussian is clipped
    cases[2][1, 4:6] = (0.004, 0.006)                             # sub-pixel box: h = w = 0, radius 0
    outs = []
    for lab in cases:
        ref = [t.numpy() for t in algo.generate_targets(lab.copy())]
        mine = C.generate_targets(lab.copy(), (fh, fw), nc, K)
        for a_, b_ in zip(ref, mine):
            assert a_.dtype == np.float32 and np.array_equal(a_, b_), float(np.abs(a_ - b_).max())
        outs.append(ref)
    labels = np.zeros((len(cases), K, 5), dtype=np.float32)
    for i, c in enumerate(cases):
        labels[i, :len(c)] = c[:, 1:]
    np.savez_compressed(os.path.join(GOLD, "centernet_targets.npz"), labels=labels, counts=np.array([len(c) for c in cases], dtype=np.int32), nc=np.array(nc),
                        fh=np.array(fh), fw=np.array(fw), heat=np.stack([o[0] for o in outs]), reg=np.stack([o[1] for o in outs]),
                        wh=np.stack([o[2] for o in outs]), mask=np.stack([o[3] for o in outs]), ind=np.stack([o[4] for o in outs]))
    report["centernet_targets"] = dict(feature=[int(fh), int(fw)], objects=[int(o[3].sum()) for o in outs])


def yolov7_loss_section(builder, report):
    """11c. Yolo7Loss (core/loss/yolo7_loss.py): the REAL reference loss object on random head outputs (scaled so that assignments are
    non-trivial) and seeded targets -- an image with several objects (some sharing cells), an image with one object, an image without --
    -> the four loss values and the gradient w.r.t. the three outputs by torch autograd.  Pins oracle/yolov7_ref.yolo7_loss."""
    from oracle import yolov7_ref as Y7
    ycfg, yalgo_cls, _ = builder.export_from_registry("yolo7")
    ycfg.train.pretrained = False
    algo = yalgo_cls(ycfg, torch.device("cpu"))
    crit = algo.build_loss()
    assert type(crit).__name__ == "Yolo7Loss"
    nc = ycfg.dataset.num_classes
    H = W = 256
    B = 3
    g = torch.Generator().manual_seed(17)
    outs = [(torch.randn(B, 3 * (5 + nc), H // s, W // s, generator=g) * 1.5).requires_grad_(True) for s in (32, 16, 8)]
    rows = []
    for b, n in ((0, 6), (1, 1)):
        for _ in range(n):
            wh = torch.rand(2, generator=g) * 0.5 + 0.05
            rows.append([b, int(torch.randint(0, nc, (1,), generator=g)), float(torch.rand(1, generator=g) * 0.8 + 0.1),
                         float(torch.rand(1, generator=g) * 0.8 + 0.1), float(wh[0]), float(wh[1])])
    rows[1][2:] = rows[0][2:]                                    # two objects on top of each other (different class): shared candidate cells
    targets = torch.tensor(rows, dtype=torch.float32)
    imgs = torch.zeros(B, 3, H, W)
    crit.input_shape = (H, W)
    crit.obj_ratio = 1 * (H * W) / (640 ** 2)
    loss, box_l, obj_l, cls_l = crit(outs, targets.clone(), imgs)
    loss.backward()
    mine_in = [o.detach().clone().requires_grad_(True) for o in outs]
    mine = Y7.yolo7_loss(mine_in, targets.clone(), float(H), nc, (H, W))
    mine[0].backward()
    for a_, b_ in zip(mine, (loss, box_l, obj_l, cls_l)):
        assert abs(float(a_) - float(b_)) <= 1e-5 * max(abs(float(b_)), 1e-6), (float(a_), float(b_))
    for a_, b_ in zip(mine_in, outs):
        assert torch.allclose(a_.grad, b_.grad, rtol=1e-4, atol=1e-8), float((a_.grad - b_.grad).abs().max())
    np.savez_compressed(os.path.join(GOLD, "yolov7_loss.npz"), nc=np.array(nc), hw=np.array([H, W]), targets=targets.numpy(),
                        items=np.array([float(loss), float(box_l), float(obj_l), float(cls_l)]),
                        **{f"out{i}": o.detach().numpy() for i, o in enumerate(outs)}, **{f"grad{i}": o.grad.numpy() for i, o in enumerate(outs)})
    report["yolov7_loss"] = dict(items=[float(loss), float(box_l), float(obj_l), float(cls_l)], targets=len(rows))


def yolov8_traj_section(builder, report):
    """50 Adam steps of the IMPORTED REFERENCE (core/trainer/yolo8_train.py:93-111 without AMP: model, Loss, torch.optim.Adam via
    lr_scheduler.get_optimizer) on one repeated 160x160 batch of 8 at lr 1e-4 -- a regime in which the loss curve is smooth (at lr 1e-3 on two
    images it is chaotic: fp16 storage alone moves it by 20-30 % per window) -- next to the oracle's fp32 restatement (asserted against
    the reference step by step) and the oracle with the engine's fp16 rounding points emulated -> tests/golden/yolov8n_traj_160.npz."""
    sys.path.insert(0, ROOT)
    from oracle import yolov8_ref as O
    from oracle import synth
    from core.trainer.lr_scheduler import get_optimizer
    STEPS, LR = 50, 1e-4
    torch.manual_seed(0)
    cfg, algo_cls, _ = builder.export_from_registry("yolo8_det")
    algo = algo_cls(cfg, torch.device("cpu"))
    model, _ = algo.build_model()
    crit = algo.build_loss(model)
    opt = get_optimizer("Adam", model, LR)
    x, batch = synth.images(8, 160, 160, seed=11), synth.targets(8, seed=12)
    model.train()
    ref = []
    for _ in range(STEPS):
        opt.zero_grad()
        loss, _items = crit(model(x.clone()), {k: v.clone() for k, v in batch.items()})
        loss.backward()
        opt.step()
        ref.append(float(loss))
    curves = {}
    for name, emulate in (("fp32", False), ("fp16_emulation", True)):
        sd, state = O.init_state_dict("n", 80, seed=0), {}
        O.FP16_STORAGE[0] = emulate
        try:
            curves[name] = np.array([float(O.train_step(sd, x.clone(), batch, state, "n", 80, LR)[0]) for _ in range(STEPS)])
        finally:
            O.FP16_STORAGE[0] = False
    ref = np.array(ref)
    dev32 = float(np.max(np.abs(curves["fp32"] - ref) / ref))
    dev16 = float(np.max(np.abs(curves["fp16_emulation"] - ref) / ref))
    # The restatement follows the reference to fp32 round-off until a discrete event (a TaskAlignedAssigner top-k flip) amplifies that
    # round-off: steps 0-19 agree to < 1e-3 per step, from step ~24 on the two fp32 curves differ by up to 13 %.  The pinned window is the
    # first 20 steps; the rest of the curves is recorded for reference.
    PIN = 20
    dev32_pin = float(np.max(np.abs(curves["fp32"][:PIN] - ref[:PIN]) / ref[:PIN]))
    dev16_pin = float(np.max(np.abs(curves["fp16_emulation"][:PIN] - ref[:PIN]) / ref[:PIN]))
    assert dev32_pin < 1e-3, ("oracle fp32 vs the reference over the first 20 steps", dev32_pin)
    report["yolov8_traj"] = ("50 reference Adam steps (160x160, batch 8, lr 1e-4): oracle fp32 within %.1e per step over steps 0-19 (%.1e over all 50: "
                             "an assignment flip near step 24 amplifies round-off), fp16-storage emulation within %.1e over steps 0-19" % (dev32_pin, dev32, dev16_pin))
    np.savez(os.path.join(GOLD, "yolov8n_traj_160.npz"), reference=ref, fp32=curves["fp32"], fp16_emulation=curves["fp16_emulation"], steps=STEPS, lr=LR, pinned_steps=PIN,
             note="oracle/make_golden.py yolov8_traj: images seed 11, targets seed 12, batch 8, 160x160, Adam lr 1e-4, seed-0 init")
    print(report["yolov8_traj"], flush=True)


def main():
    sys.path.insert(0, ROOT)
    from oracle import yolov8_ref as O
    from oracle import nms_ref, synth
    os.makedirs(GOLD, exist_ok=True)
    builder = _import_reference()
    torch.set_num_threads(8)
    report = {}

    # ---- 1. init weights ----------------------------------------------------------------------
    torch.manual_seed(0)
    cfg, algo_cls, _ = builder.export_from_registry("yolo8_det")
    algo = algo_cls(cfg, torch.device("cpu"))
    model, name = algo.build_model()
    ref_sd = model.state_dict()
    my_sd = O.init_state_dict("n", 80, seed=0)
    assert list(ref_sd.keys()) == list(my_sd.keys()), "state_dict key order differs"
    for k in ref_sd:
        assert ref_sd[k].shape == my_sd[k].shape, k
        assert torch.equal(ref_sd[k], my_sd[k]), f"init mismatch {k}"
    report["init"] = "bit-exact, %d tensors" % len(ref_sd)
    sums = {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in ref_sd.items()}
    with open(os.path.join(GOLD, "yolov8n_seed0_init_sums.json"), "w") as f:
        json.dump(sums, f)

    # ---- 2. forward (train + eval), small input -----------------------------------------------
    g = torch.Generator().manual_seed(1)
    x = torch.rand(2, 3, 128, 128, generator=g)
    model.train()
    ref_train = [t.detach().clone() for t in model(x.clone())]
    sd_a = {k: v.clone() for k, v in my_sd.items()}
    my_train = O.forward(sd_a, x.clone(), "n", 80, training=True)
    for r, m in zip(ref_train, my_train):
        assert torch.allclose(r, m, rtol=1e-5, atol=1e-6), float((r - m).abs().max())
    for k, v in model.state_dict().items():                  # BN running stats after one train fwd
        assert torch.allclose(v.float(), sd_a[k].float(), rtol=1e-5, atol=1e-7), k
    model.eval()
    with torch.no_grad():
        ref_y, ref_feats = model(x.clone())
        my_y, my_feats = O.forward(sd_a, x.clone(), "n", 80, training=False)
    assert torch.allclose(ref_y, my_y, rtol=1e-5, atol=1e-5), float((ref_y - my_y).abs().max())
    report["forward"] = "train+eval allclose 1e-5"
    bn_probe = {k: sd_a[k].numpy() for k in ("model.0.bn.running_mean", "model.0.bn.running_var",
                                             "model.22.cv3.2.1.bn.running_mean", "model.22.cv3.2.1.bn.running_var")}
    np.savez_compressed(os.path.join(GOLD, "yolov8n_fwd_128.npz"), x=x.numpy(),
                        train0=ref_train[0].numpy(), train1=ref_train[1].numpy(), train2=ref_train[2].numpy(),
                        eval_y=ref_y.numpy(), **{"bn:" + k: v for k, v in bn_probe.items()})

    # ---- 3. loss + grads + two Adam steps -----------------------------------------------------
    torch.manual_seed(0)
    cfg, algo_cls, _ = builder.export_from_registry("yolo8_det")
    algo = algo_cls(cfg, torch.device("cpu"))
    model, _ = algo.build_model()
    crit = algo.build_loss(model)
    from core.trainer.lr_scheduler import get_optimizer
    opt = get_optimizer("Adam", model, 1e-3)
    my_sd = O.init_state_dict("n", 80, seed=0)
    state = {}
    g = torch.Generator().manual_seed(1)
    x = torch.rand(4, 3, 160, 160, generator=g)
    batch = synth.targets(4, seed=2)
    keys = O.trainable_keys(my_sd)
    named = dict(model.named_parameters())
    step_out = {}
    model.train()
    for step in range(2):
        opt.zero_grad()
        preds = model(x.clone())
        loss, items = crit(preds, {k: v.clone() for k, v in batch.items()})
        loss.backward()
        ref_grads = {k: named[k].grad.detach().clone() for k in keys if named[k].grad is not None}
        opt.step()
        my_loss, my_items, my_grads, _ = O.train_step(my_sd, x.clone(), batch, state, "n", 80, 1e-3)
        assert torch.allclose(loss.detach(), my_loss, rtol=1e-5), (float(loss), float(my_loss))
        assert torch.allclose(items, my_items, rtol=1e-5, atol=1e-6), (items, my_items)
        worst = 0.0
        for k in keys:
            if k not in ref_grads:
                continue
            gr, gm = ref_grads[k], my_grads[k]
            err = float((gr - gm).norm() / (gr.norm() + 1e-12))
            worst = max(worst, err)
            assert err < 2e-4, (k, err)
        for k, v in model.state_dict().items():
            if v.dtype.is_floating_point:
                assert torch.allclose(v, my_sd[k], rtol=1e-4, atol=1e-6), (k, float((v - my_sd[k]).abs().max()))
        step_out[step] = dict(loss=float(loss), items=items.numpy().copy(), worst_grad_rel=worst,
                              grad_norms=np.array([float(ref_grads[k].norm()) if k in ref_grads else 0.0 for k in keys]),
                              g_stem=ref_grads["model.0.conv.weight"].numpy().copy(),
                              g_c2f=ref_grads["model.2.m.0.cv1.conv.weight"].numpy().copy(),
                              g_headb=ref_grads["model.22.cv3.0.2.bias"].numpy().copy(),
                              g_bn=ref_grads["model.9.cv2.bn.weight"].numpy().copy())
    report["train"] = {s: dict(loss=v["loss"], worst_grad_rel=v["worst_grad_rel"]) for s, v in step_out.items()}
    post = model.state_dict()
    np.savez_compressed(
        os.path.join(GOLD, "yolov8n_train_160.npz"), x=x.numpy(), batch_idx=batch["batch_idx"].numpy(),
        cls=batch["cls"].numpy(), bboxes=batch["bboxes"].numpy(), keys=np.array(keys),
        loss=np.array([step_out[0]["loss"], step_out[1]["loss"]]),
        items=np.stack([step_out[0]["items"], step_out[1]["items"]]),
        grad_norms=np.stack([step_out[0]["grad_norms"], step_out[1]["grad_norms"]]),
        g_stem=step_out[0]["g_stem"], g_c2f=step_out[0]["g_c2f"], g_headb=step_out[0]["g_headb"], g_bn=step_out[0]["g_bn"],
        w_stem_after2=post["model.0.conv.weight"].numpy(), w_head_after2=post["model.22.cv2.1.2.weight"].numpy(),
        bn_after2=post["model.4.m.1.cv2.bn.weight"].numpy(), rv_after2=post["model.4.m.1.cv2.bn.running_var"].numpy())

    # ---- 4. assigner-only fixture (pins TAL edge handling on harder targets) -------------------
    from core.utils.bboxes import TaskAlignedAssigner
    g = torch.Generator().manual_seed(5)
    B, A, nc, G = 2, 336, 80, 6
    anchors, stride_t = O.make_anchors([(16, 16), (8, 8), (4, 4)], (8.0, 16.0, 32.0))
    pd_scores = torch.rand(B, A, nc, generator=g) * 0.3
    ctr = anchors * stride_t
    half = torch.rand(B, A, 2, generator=g) * 40 + 4
    pd_bboxes = torch.cat((ctr - half, ctr + half * (0.5 + torch.rand(B, A, 2, generator=g))), -1)
    gxy = torch.rand(B, G, 2, generator=g) * 80 + 24
    gwh = torch.rand(B, G, 2, generator=g) * 60 + 10
    gt_bboxes = torch.cat((gxy - gwh / 2, gxy + gwh / 2), -1)
    gt_labels = torch.randint(0, nc, (B, G, 1), generator=g).float()
    gt_bboxes[1, 4:] = 0                                   # padded rows in image 1
    gt_labels[1, 4:] = 0
    gt_bboxes[0, 1] = gt_bboxes[0, 0] + 3.0                # heavy overlap -> multi-assignment path
    mask_gt = (gt_bboxes.sum(2, keepdim=True) > 0).float()
    ref_asg = TaskAlignedAssigner(topk=10, num_classes=nc, alpha=0.5, beta=6.0)
    _, rb, rs, rf, ri = ref_asg(pd_scores, pd_bboxes, ctr, gt_labels, gt_bboxes, mask_gt)
    mb, ms, mf, mi = O.task_aligned_assign(pd_scores, pd_bboxes, ctr, gt_labels, gt_bboxes, mask_gt)
    assert torch.equal(rf, mf) and torch.equal(ri, mi), "assigner fg/gt_idx mismatch"
    assert torch.allclose(rb, mb) and torch.allclose(rs, ms, rtol=1e-5, atol=1e-7)
    report["assigner"] = "fg=%d exact" % int(rf.sum())
    np.savez_compressed(os.path.join(GOLD, "tal_assign.npz"), pd_scores=pd_scores.numpy(), pd_bboxes=pd_bboxes.numpy(),
                        anc=ctr.numpy(), gt_labels=gt_labels.numpy(), gt_bboxes=gt_bboxes.numpy(), mask_gt=mask_gt.numpy(),
                        t_boxes=rb.numpy(), t_scores_sum=rs.sum(-1).numpy(), t_cls=rs.argmax(-1).numpy(),
                        fg=rf.numpy(), gt_idx=ri.numpy())

    # ---- 5. full-size forward, sub-sampled ----------------------------------------------------
    torch.manual_seed(0)
    cfg, algo_cls, _ = builder.export_from_registry("yolo8_det")
    model, _ = algo_cls(cfg, torch.device("cpu")).build_model()
    g = torch.Generator().manual_seed(1)
    x = torch.rand(1, 3, 640, 640, generator=g)
    model.train()
    with torch.no_grad():
        full = model(x)
    sub = {f"lvl{i}": t.flatten()[::97].numpy() for i, t in enumerate(full)}
    norms = np.array([float(t.norm()) for t in full])
    my = O.forward(O.init_state_dict("n", 80, seed=0), x, "n", 80, training=True)
    for r, m in zip(full, my):
        assert torch.allclose(r, m.detach(), rtol=1e-5, atol=1e-5)
    np.savez_compressed(os.path.join(GOLD, "yolov8n_fwd_640_sub.npz"), norms=norms, **sub)
    report["forward640"] = "allclose; norms " + str(norms.tolist())

    # ---- 7. data-parallel simulation: the REFERENCE run shard by shard, gradients averaged ----------
    # N ranks = N sequential shards of one global batch with identical weights; every shard normalises its loss by its
    # own target_scores_sum and batch size (core/algorithms/yolo_v8.py:109,124) and uses its own BN batch statistics; the
    # exchange must deliver the MEAN over shards of the per-shard gradients (not the single-process full-batch gradient).
    g = torch.Generator().manual_seed(21)
    xg = torch.rand(8, 3, 96, 96, generator=g)
    bg = synth.targets(8, seed=22)
    dp = {"x": xg.numpy(), "batch_idx": bg["batch_idx"].numpy(), "cls": bg["cls"].numpy(), "bboxes": bg["bboxes"].numpy()}
    keys = O.trainable_keys(O.init_state_dict("n", 80, seed=0))
    probe = ("model.0.conv.weight", "model.9.cv2.bn.weight", "model.15.cv2.conv.weight", "model.22.cv2.0.2.weight", "model.22.cv3.0.2.bias")
    for world in (1, 2, 4, 8):
        per = 8 // world
        acc = None
        for r in range(world):
            torch.manual_seed(0)
            cfg, algo_cls, _ = builder.export_from_registry("yolo8_det")
            algo = algo_cls(cfg, torch.device("cpu"))
            model, _ = algo.build_model()
            crit = algo.build_loss(model)
            model.train()
            sel = (bg["batch_idx"] >= r * per) & (bg["batch_idx"] < (r + 1) * per)
            sb = {"batch_idx": bg["batch_idx"][sel] - r * per, "cls": bg["cls"][sel], "bboxes": bg["bboxes"][sel]}
            loss, _ = crit(model(xg[r * per:(r + 1) * per].clone()), sb)
            loss.backward()
            named = dict(model.named_parameters())
            gr = {k: (named[k].grad.detach().clone() if named[k].grad is not None else torch.zeros_like(named[k])) for k in keys}
            # the oracle must give the same shard gradients (it is what the CPU gloo test of the exchange runs per rank)
            mine = O.train_step(O.init_state_dict("n", 80, seed=0), xg[r * per:(r + 1) * per].clone(), sb, {})[2]
            for k in keys:
                assert float((gr[k] - mine[k]).norm() / (gr[k].norm() + 1e-12)) < 2e-4, (world, r, k)
            acc = gr if acc is None else {k: acc[k] + gr[k] for k in keys}
        mean = {k: acc[k] / world for k in keys}
        flat = torch.cat([mean[k].flatten() for k in keys])
        dp[f"w{world}_norm"] = np.array(float(flat.norm()))
        dp[f"w{world}_sub"] = flat[::211].numpy().copy()
        for k in probe:
            dp[f"w{world}:{k}"] = mean[k].numpy().copy()
    dp["keys"] = np.array(keys)
    np.savez_compressed(os.path.join(GOLD, "dp_sim_96.npz"), **dp)
    report["dp_sim"] = {f"world{w}": float(dp[f"w{w}_norm"]) for w in (1, 2, 4, 8)}

    # ---- 8. YOLOv8-s (BASELINE config 3's per-rank model): init sums, train forward, loss, gradient norms -------------
    torch.manual_seed(0)
    cfg, algo_cls, _ = builder.export_from_registry("yolo8_det")
    cfg.arch.model_type = "s"
    algo = algo_cls(cfg, torch.device("cpu"))
    model, _ = algo.build_model()
    crit = algo.build_loss(model)
    s_sd = O.init_state_dict("s", 80, seed=0)
    ref_sd = model.state_dict()
    assert list(ref_sd.keys()) == list(s_sd.keys())
    for k in ref_sd:
        assert torch.equal(ref_sd[k], s_sd[k]), f"YOLOv8-s init mismatch {k}"
    g = torch.Generator().manual_seed(31)
    xs = torch.rand(2, 3, 160, 160, generator=g)
    bs_ = synth.targets(2, seed=32)
    model.train()
    preds = model(xs.clone())
    loss, items = crit(preds, {k: v.clone() for k, v in bs_.items()})
    loss.backward()
    named = dict(model.named_parameters())
    skeys = O.trainable_keys(s_sd)
    my_loss, my_items, my_grads, my_feats = O.train_step(s_sd, xs.clone(), bs_, {}, "s", 80, 1e-3)
    assert torch.allclose(loss.detach(), my_loss, rtol=1e-5)
    for r_, m_ in zip(preds, my_feats):
        assert torch.allclose(r_, m_.detach(), rtol=1e-5, atol=1e-5)
    for k in skeys:
        if named[k].grad is not None:
            assert float((named[k].grad - my_grads[k]).norm() / (named[k].grad.norm() + 1e-12)) < 2e-4, k
    np.savez_compressed(
        os.path.join(GOLD, "yolov8s_train_160.npz"), x=xs.numpy(), batch_idx=bs_["batch_idx"].numpy(), cls=bs_["cls"].numpy(),
        bboxes=bs_["bboxes"].numpy(), train0=preds[0].detach().numpy(), train1=preds[1].detach().numpy(), train2=preds[2].detach().numpy(),
        loss=np.array(float(loss)), items=items.numpy().copy(), keys=np.array(skeys),
        grad_norms=np.array([float(named[k].grad.norm()) if named[k].grad is not None else 0.0 for k in skeys]),
        g_headb=named["model.22.cv3.0.2.bias"].grad.numpy().copy(), g_stem=named["model.0.conv.weight"].grad.numpy().copy(),
        n_params=np.array(sum(p_.numel() for p_ in model.parameters())))
    report["yolov8s"] = dict(loss=float(loss), params=int(sum(p_.numel() for p_ in model.parameters())))

    # ---- 9. CenterNet DLA-34 (BASELINE config 4): init, forward, decode ------------------------------------
    from oracle import centernet_ref as C
    torch.manual_seed(0)
    ccfg, calgo_cls, _ = builder.export_from_registry("centernet")
    ccfg.dataset.num_classes = 80
    ccfg.arch.input_size = (3, 128, 128)
    calgo = calgo_cls(ccfg, torch.device("cpu"))
    cmodel, cname = calgo.build_model()
    ref_sd = cmodel.state_dict()
    my_sd = C.init_state_dict(80, seed=0)
    assert list(ref_sd.keys()) == list(my_sd.keys()), [(a_, b_) for a_, b_ in zip(ref_sd, my_sd) if a_ != b_][:5]
    for k in ref_sd:
        assert ref_sd[k].shape == my_sd[k].shape and torch.equal(ref_sd[k], my_sd[k]), f"CenterNet init mismatch {k}"
    csums = {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in ref_sd.items()}
    with open(os.path.join(GOLD, "centernet_seed0_init_sums.json"), "w") as f:
        json.dump(csums, f)
    g = torch.Generator().manual_seed(41)
    xc = torch.rand(2, 3, 128, 128, generator=g)
    cmodel.train()
    ref_tr = cmodel(xc.clone()).detach()
    sd_t = {k: v.clone() for k, v in my_sd.items()}
    my_tr = C.forward(sd_t, xc.clone(), 80, training=True)
    assert torch.allclose(ref_tr, my_tr, rtol=1e-4, atol=1e-5), float((ref_tr - my_tr).abs().max())
    for k, v in cmodel.state_dict().items():                    # BN running statistics after one train-mode forward
        assert torch.allclose(v.float(), sd_t[k].float(), rtol=1e-4, atol=1e-6), k
    cmodel.eval()
    with torch.no_grad():
        ref_ev = cmodel(xc.clone())
        my_ev = C.forward(sd_t, xc.clone(), 80, training=False)
    assert torch.allclose(ref_ev, my_ev, rtol=1e-4, atol=1e-5), float((ref_ev - my_ev).abs().max())
    # decode: the network output of image 0, and a synthetic head output with real peaks
    gs = torch.Generator().manual_seed(42)
    synth_pred = torch.cat((torch.randn(1, 32, 32, 80, generator=gs) * 2.5 - 3.0, torch.rand(1, 32, 32, 2, generator=gs),
                            torch.rand(1, 32, 32, 2, generator=gs) * 12 + 1), -1)
    dec = {}
    for tag, pr, hw in (("net", ref_ev[0:1].clone(), (96, 128)), ("synth", synth_pred, (375, 500))):
        rb, rs, rc = calgo.decode_boxes(pr.clone(), hw[0], hw[1])
        mb, ms, mc, mpos = C.decode(pr.clone(), 80, (128, 128), hw, k=ccfg.decode.max_boxes_per_img, conf=ccfg.decode.score_threshold,
                                    nms_thr=ccfg.decode.nms_threshold, use_nms=ccfg.decode.use_nms)
        assert rb.shape[0] == mb.shape[0] and rb.shape[0] > 3, (tag, rb.shape, mb.shape)
        assert np.array_equal(rc, mc.numpy()) and np.allclose(rs, ms.numpy(), rtol=0, atol=0), tag
        assert np.allclose(rb, mb.numpy(), rtol=1e-6, atol=1e-5), tag
        dec[tag + "_pred"] = pr.numpy().copy()
        dec[tag + "_hw"] = np.array(hw)
        dec[tag + "_boxes"], dec[tag + "_scores"], dec[tag + "_classes"], dec[tag + "_pos"] = rb, rs, rc, mpos.numpy()
    np.savez_compressed(os.path.join(GOLD, "centernet_fwd_128.npz"), x=xc.numpy(), eval_sub=ref_ev.flatten()[::7].numpy().copy(),
                        train_sub=ref_tr.flatten()[::7].numpy().copy(), eval_norm=np.array(float(ref_ev.norm())),
                        bn_rm=cmodel.state_dict()["backbone.dla_up.ida_2.node_3.1.running_mean"].numpy().copy(),
                        bn_rv=cmodel.state_dict()["backbone.base.level_5.root.bn.running_var"].numpy().copy(),
                        k=np.array(ccfg.decode.max_boxes_per_img), conf=np.array(ccfg.decode.score_threshold),
                        nms_thr=np.array(ccfg.decode.nms_threshold), **dec)
    report["centernet"] = dict(init="bit-exact, %d tensors" % len(ref_sd), params=int(sum(p_.numel() for p_ in cmodel.parameters())),
                               decode={t: int(dec[t + "_boxes"].shape[0]) for t in ("net", "synth")})

    centernet_train_section(builder, report)
    centernet_loss_section(builder, report)
    centernet_targets_section(builder, report)

    # ---- 10. DeepLabv3+ ResNet-101: init, eval forward on a calibrated network ---------------------------------
    from oracle import deeplab_ref as D
    dcfg, dalgo_cls, _ = builder.export_from_registry("deeplabv3plus")
    torch.manual_seed(0)
    dmodel, _ = dalgo_cls(dcfg, torch.device("cpu")).build_model()
    ref_sd = dmodel.state_dict()
    my_sd = D.init_state_dict(dcfg.dataset.num_classes, seed=0)
    assert list(ref_sd.keys()) == list(my_sd.keys()), [(a_, b_) for a_, b_ in zip(ref_sd, my_sd) if a_ != b_][:5]
    for k in ref_sd:
        assert ref_sd[k].shape == my_sd[k].shape and torch.equal(ref_sd[k], my_sd[k]), f"DeepLab init mismatch {k}"
    dsums = {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in ref_sd.items() if not k.endswith("num_batches_tracked")}
    with open(os.path.join(GOLD, "deeplab_seed0_init_sums.json"), "w") as f:
        json.dump(dsums, f)
    # Forward fixture on a well-conditioned network.  (i) With unit BN weights everywhere a random-init ResNet-101 is chaotic:
    # every residual branch is as large as the stream it joins, and a perturbation of 5e-4 (one fp16 rounding) grows to 40 %
    # over the 33 blocks in the reference's own arithmetic with fp16 operands -- no implementation can be compared on it.
    # The last BatchNorm of every block gets weight 0.1 (torchvision's zero_init_residual idea, resnet.py:180-185, not
    # quite zero so that the branches still matter): 4e-3 instead.  (ii) In eval mode the untouched running statistics (0 / 1)
    # leave the network un-normalised: they are calibrated with ONE train-mode pass at momentum 1 (running := batch
    # statistics); the eval-mode forward of another batch is the fixture.
    GAMMA3 = 0.1
    with torch.no_grad():
        for k_, v_ in dmodel.named_parameters():
            if k_.endswith(".bn3.weight"):
                v_.fill_(GAMMA3)
    g = torch.Generator().manual_seed(51)
    xcal, xd = torch.rand(2, 3, 193, 225, generator=g), torch.rand(2, 3, 193, 225, generator=g)
    for m_ in dmodel.modules():
        if isinstance(m_, torch.nn.BatchNorm2d):
            m_.momentum = 1.0
    dmodel.train()
    with torch.no_grad():
        dmodel(xcal)
    dmodel.eval()
    with torch.no_grad():
        ref_out = dmodel(xd.clone())
    cal_sd = {k: v.clone() for k, v in dmodel.state_dict().items()}
    with torch.no_grad():
        my_out, my_rows = D.forward(cal_sd, xd.clone(), dcfg.dataset.num_classes, return_rows=True)
    assert torch.allclose(ref_out, my_out, rtol=1e-4, atol=1e-4 * float(ref_out.abs().max())), float((ref_out - my_out).abs().max())
    stats = {k: v.numpy().copy() for k, v in cal_sd.items() if k.endswith("running_mean") or k.endswith("running_var")}
    np.savez_compressed(os.path.join(GOLD, "deeplab_fwd_193x225.npz"), x=xd.numpy(), rows=my_rows.numpy().copy(),
                        out_sub=ref_out.flatten()[::11].numpy().copy(), out_norm=np.array(float(ref_out.norm())), bn3_gamma=np.array(GAMMA3),
                        stat_keys=np.array(list(stats.keys())), stat_vals=np.concatenate([v.ravel() for v in stats.values()]))
    report["deeplab"] = dict(init="bit-exact, %d tensors" % len(ref_sd), params=int(sum(p_.numel() for p_ in dmodel.parameters())),
                             out_absmax=float(ref_out.abs().max()), rows_shape=list(my_rows.shape))

    deeplab_train_section(builder, report)

    # ---- 11. YOLOv7-l: init, eval forward on a calibrated network, decode, NMS bookkeeping ------------
    from oracle import yolov7_ref as Y7
    ycfg, yalgo_cls, _ = builder.export_from_registry("yolo7")
    ycfg.train.pretrained = False                                # (the default loads saves/yolov7_weights.pth)
    torch.manual_seed(0)
    yalgo = yalgo_cls(ycfg, torch.device("cpu"))
    ymodel, _ = yalgo.build_model()
    ref_sd = ymodel.state_dict()
    nc7 = ycfg.dataset.num_classes
    my_sd = Y7.init_state_dict(nc7, seed=0)
    assert list(ref_sd.keys()) == list(my_sd.keys())
    for k in ref_sd:
        assert ref_sd[k].shape == my_sd[k].shape and torch.equal(ref_sd[k], my_sd[k]), f"YOLOv7 init mismatch {k}"
    ysums = {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in ref_sd.items() if not k.endswith("num_batches_tracked")}
    with open(os.path.join(GOLD, "yolov7_seed0_init_sums.json"), "w") as f:
        json.dump(ysums, f)
    # N(0, 0.02) weights under untouched running statistics let the signal die out layer by layer: calibrate the BatchNorms
    # with one train-mode pass at momentum 1, then the eval forward of another batch is the fixture; the head biases get a
    # spread so that objectness / class scores are not all 0.5
    g = torch.Generator().manual_seed(61)
    HW7 = (160, 224)
    xcal, x7 = torch.rand(2, 3, *HW7, generator=g), torch.rand(2, 3, *HW7, generator=g)
    with torch.no_grad():
        for hk in ("yolo_head_P3", "yolo_head_P4", "yolo_head_P5"):
            getattr(ymodel, hk).bias.copy_(torch.randn(3 * (5 + nc7), generator=g) * 1.5 - 1.0)
    for m_ in ymodel.modules():
        if isinstance(m_, torch.nn.BatchNorm2d):
            m_.momentum = 1.0
    ymodel.train()
    with torch.no_grad():
        ymodel(xcal)
    ymodel.eval()
    with torch.no_grad():
        ref7 = ymodel(x7.clone())
    cal7 = {k: v.clone() for k, v in ymodel.state_dict().items()}
    with torch.no_grad():
        my7 = Y7.forward(cal7, x7.clone())
    for a_, b_ in zip(ref7, my7):
        assert torch.allclose(a_, b_, rtol=1e-4, atol=1e-5 * float(a_.abs().max())), float((a_ - b_).abs().max())
    # decode: the tensor decode_box hands to _nms (captured), then _nms itself with torchvision's nms replaced by the
    # oracle's greedy restatement -- everything around the suppression primitive is the reference's own code
    yalgo.input_image_size = list(HW7)
    captured = {}
    yalgo._nms = lambda pred, *a_: captured.setdefault("dec", pred)
    yalgo.decode_box(ref7, HW7[0], HW7[1])
    del yalgo._nms
    my_dec = Y7.decode(ref7, nc7, HW7)
    assert torch.equal(captured["dec"], my_dec), float((captured["dec"] - my_dec).abs().max())
    import core.algorithms.yolo_v7 as ref_y7mod
    from oracle import nms_ref as _nr

    def _tv_nms(boxes, scores, iou_threshold):
        o_ = np.argsort(-scores.numpy(), kind="stable")
        return torch.from_numpy(o_[_nr._greedy(boxes.numpy()[o_], None, iou_threshold)].copy())

    ref_y7mod.nms = _tv_nms
    yalgo.letterbox_image = False
    conf7, thr7 = ycfg.decode.conf_threshold, ycfg.decode.nms_threshold
    ref_nms = yalgo._nms(captured["dec"].clone(), HW7, [1, 1], conf7)
    my_nms = Y7.nms(my_dec, nc7, conf7, thr7)
    n_det = []
    for r_, (m_rows, m_idx) in zip(ref_nms, my_nms):
        assert (r_ is None) == (m_rows is None)
        if r_ is not None:
            assert r_.shape == m_rows.shape and np.allclose(r_, m_rows, rtol=1e-5, atol=1e-6), float(np.abs(r_ - m_rows).max())
            n_det.append(int(r_.shape[0]))
    assert sum(n_det) > 10, n_det
    stats7 = {k: v.numpy().copy() for k, v in cal7.items() if k.endswith("running_mean") or k.endswith("running_var")}
    np.savez_compressed(os.path.join(GOLD, "yolov7_fwd_160x224.npz"), x=x7.numpy(), out0=ref7[0].numpy().copy(), out1=ref7[1].numpy().copy(),
                        out2_sub=ref7[2].flatten()[::5].numpy().copy(), dec_sub=my_dec.flatten()[::7].numpy().copy(), conf=np.array(conf7),
                        nms_thr=np.array(thr7), keep0=my_nms[0][1], keep1=my_nms[1][1], rows0=my_nms[0][0], rows1=my_nms[1][0],
                        head_bias=np.stack([cal7[h_ + ".bias"].numpy() for h_ in ("yolo_head_P3", "yolo_head_P4", "yolo_head_P5")]),
                        stat_keys=np.array(list(stats7.keys())), stat_vals=np.concatenate([v.ravel() for v in stats7.values()]))
    report["yolov7"] = dict(init="bit-exact, %d tensors" % len(ref_sd), params=int(sum(p_.numel() for p_ in ymodel.parameters())),
                            detections=n_det, out_absmax=[float(o_.abs().max()) for o_ in ref7])

    yolov7_train_section(builder, report)
    yolov7_loss_section(builder, report)

    # ---- 12. SSD300 VGG16-BN: init, priors, eval forward on a calibrated network, decode ----------------------
    from oracle import ssd_ref as SS
    scfg, salgo_cls, _ = builder.export_from_registry("ssd")
    torch.manual_seed(0)
    salgo = salgo_cls(scfg, torch.device("cpu"))
    smodel, sname = salgo.build_model()
    ref_sd = smodel.state_dict()
    ncs = scfg.dataset.num_classes
    my_sd = SS.init_state_dict(ncs, seed=0)
    assert list(ref_sd.keys()) == list(my_sd.keys())
    for k in ref_sd:
        assert ref_sd[k].shape == my_sd[k].shape and torch.equal(ref_sd[k], my_sd[k]), f"SSD init mismatch {k}"
    ssums = {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in ref_sd.items() if not k.endswith("num_batches_tracked")}
    with open(os.path.join(GOLD, "ssd_seed0_init_sums.json"), "w") as f:
        json.dump(ssums, f)
    assert np.array_equal(SS.priors((300, 300)), salgo.anchors)
    g = torch.Generator().manual_seed(71)
    xcal = torch.rand(2, 3, 300, 300, generator=g)
    xs_u8 = (torch.rand(2, 3, 300, 300, generator=g) * 255).round().to(torch.uint8)   # stored as bytes: x = u8 / 255
    xs_ = xs_u8.float() / 255.0
    for m_ in smodel.modules():
        if isinstance(m_, torch.nn.BatchNorm2d):
            m_.momentum = 1.0
    smodel.train()
    with torch.no_grad():
        smodel(xcal)
    smodel.eval()
    with torch.no_grad():
        rloc, rconf = smodel(xs_.clone())
    cals = {k: v.clone() for k, v in smodel.state_dict().items()}
    with torch.no_grad():
        mloc, mconf = SS.forward(cals, xs_.clone(), ncs)
    assert torch.allclose(rloc, mloc, rtol=1e-4, atol=1e-5) and torch.allclose(rconf, mconf, rtol=1e-4, atol=1e-5)
    # decode on synthetic head outputs with real peaks (a random-init network's class scores never pass the 0.7 threshold);
    # the reference's decode_boxes runs with torchvision's nms replaced by the oracle's greedy restatement, no letterbox
    gs = torch.Generator().manual_seed(72)
    sloc = torch.randn(2, 8732, 4, generator=gs)
    sconf = torch.randn(2, 8732, ncs + 1, generator=gs)
    hot = torch.randint(0, 8732, (2, 400), generator=gs)
    hotc = torch.randint(1, ncs + 1, (2, 400), generator=gs)
    for b_ in range(2):
        sconf[b_, hot[b_], hotc[b_]] += 7.0 + torch.rand(400, generator=gs) * 3
    sloc, sconf = sloc.half().float(), sconf.half().float()      # stored as fp16: the synthetic inputs ARE the rounded values
    import core.algorithms.ssd as ref_ssdmod

    def _tv_nms_s(boxes, scores, iou_threshold):
        o_ = np.argsort(-scores.numpy(), kind="stable")
        return torch.from_numpy(o_[_nr._greedy(boxes.numpy()[o_], None, iou_threshold)].copy())

    ref_ssdmod.nms = _tv_nms_s
    salgo.letterbox_image = False
    ref_dec = salgo.decode_boxes((sloc.clone(), sconf.clone()), 1, 1)
    my_dec = SS.decode(sloc, sconf, SS.priors((300, 300)), ncs, scfg.decode.confidence_threshold, scfg.decode.nms_threshold)
    boxes0 = SS.parse_loc(sloc[0], SS.priors((300, 300)))
    assert torch.equal(boxes0, salgo._parse_mbox_loc(sloc[0]))
    nds = []
    for r_, (m_rows, _p) in zip(ref_dec, my_dec):
        r_ = np.asarray(r_, dtype=np.float32).reshape(-1, 6)
        assert r_.shape == m_rows.shape and np.allclose(r_, m_rows, rtol=1e-5, atol=1e-6), (r_.shape, m_rows.shape)
        nds.append(int(r_.shape[0]))
    assert min(nds) > 50, nds
    stats_s = {k: v.numpy().copy() for k, v in cals.items() if k.endswith("running_mean") or k.endswith("running_var")}
    np.savez_compressed(os.path.join(GOLD, "ssd_fwd_300.npz"), x_u8=xs_u8.numpy(), loc=rloc.numpy().copy(), conf_sub=rconf.flatten()[::5].numpy().copy(),
                        conf_norm=np.array(float(rconf.norm())), sloc=sloc.half().numpy(), sconf=sconf.half().numpy(), conf_thr=np.array(scfg.decode.confidence_threshold),
                        nms_thr=np.array(scfg.decode.nms_threshold), rows0=my_dec[0][0], rows1=my_dec[1][0], pairs0=my_dec[0][1], pairs1=my_dec[1][1],
                        stat_keys=np.array(list(stats_s.keys())), stat_vals=np.concatenate([v.ravel() for v in stats_s.values()]))
    report["ssd"] = dict(init="bit-exact, %d tensors" % len(ref_sd), params=int(sum(p_.numel() for p_ in smodel.parameters())), name=sname,
                         detections=nds, out_absmax=[float(rloc.abs().max()), float(rconf.abs().max())])

    # ---- 6. NMS tail fixture (oracle-generated; upstream parity unpinned) ----------------------
    pred = synth.nms_pred(7)
    res = nms_ref.non_max_suppression(pred, 0.25, 0.7, 300)
    np.savez_compressed(os.path.join(GOLD, "nms_tail.npz"), seed=np.array(7), pred_sum=np.array(pred.astype(np.float64).sum()),
                        keep0=res[0][1], keep1=res[1][1], rows0=res[0][0], rows1=res[1][0])
    report["nms"] = "kept %d / %d" % (len(res[0][1]), len(res[1][1]))

    ssd_train_section(builder, report)
    ssd_loss_section(builder, report)
    ssd_targets_section(builder, report)
    with open(os.path.join(GOLD, "PIN_REPORT.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


def only(section):
    """python oracle/make_golden.py deeplab_train: regenerate one fixture, merge its entry into PIN_REPORT.json."""
    sys.path.insert(0, ROOT)
    import oracle  # noqa: the package stays importable after the reference takes over sys.path
    os.makedirs(GOLD, exist_ok=True)
    builder = _import_reference()
    torch.set_num_threads(8)
    report = {}
    {"yolov8_traj": yolov8_traj_section, "deeplab_train": deeplab_train_section, "yolov7_train": yolov7_train_section, "centernet_train": centernet_train_section, "ssd_train": ssd_train_section, "centernet_loss": centernet_loss_section, "ssd_loss": ssd_loss_section, "ssd_targets": ssd_targets_section, "centernet_targets": centernet_targets_section, "yolov7_loss": yolov7_loss_section}[section](builder, report)
    path = os.path.join(GOLD, "PIN_REPORT.json")
    full = json.load(open(path)) if os.path.exists(path) else {}
    full.update(report)
    with open(path, "w") as f:
        json.dump(full, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        only(sys.argv[1])
        sys.exit(0)
    main()
