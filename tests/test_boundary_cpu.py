"""CPU: the drop-in boundary (registry / builder / configs / model object contract) and the C-ABI library
loading -- no compute calls (there is no GPU here)."""
import ctypes
import os
import re

import pytest
import numpy as np
import torch

import builder
import check
import registry
from computervision.pytorch_amd import CvxError, LIB_PATH
from computervision.pytorch_amd import _lib as L
from oracle import yolov8_ref as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- registry.py / builder.py / check.py (reference registry.py:1-61, builder.py:8-31, check.py:1-14) ----
def test_register_decorator_forms_and_key_prefix(capsys):
    reg = registry.Register("thing")
    assert reg.name == "thing"

    @reg("a")
    def fa():
        return 1

    @reg
    def fb():
        return 2

    assert set(reg.keys()) == {"thing_a", "thing_fb"}
    assert reg["thing_a"] is fa and "thing_fb" in reg and fb in reg.values()
    assert dict(reg.items())["thing_fb"]() == 2
    reg("a")(fb)                                   # duplicate: warns and overwrites
    assert "thing_a" in capsys.readouterr().out and reg["thing_a"] is fb
    with pytest.raises(Exception):
        reg["x"] = 3                               # non-callable
    assert str(reg).startswith("{")


def test_export_from_registry_contract():
    cfg, algo, trainer = builder.export_from_registry("YOLO8_DET")          # lower-cased
    assert type(cfg).__name__ == "Yolo8DetConfig" and isinstance(algo, type) and isinstance(trainer, type)
    assert algo.__name__ == "YOLOv8" and trainer.__name__ == "Yolo8Trainer"
    with pytest.raises(ValueError):
        builder.export_from_registry("resnet")                               # not in check.MODELS
    assert check.MODELS == ["yolo7", "yolo8_det", "ssd", "centernet", "deeplabv3plus"]
    for name in check.MODELS:                                               # every whitelisted model resolves
        c, a, t = builder.export_from_registry(name)
        assert isinstance(a, type) and isinstance(t, type)
    check.MODELS.append("whitelisted_but_unregistered")                     # builder.py:19-24: known name, no registry entry -> KeyError
    try:
        with pytest.raises(KeyError):
            builder.export_from_registry("whitelisted_but_unregistered")
    finally:
        check.MODELS.pop()


def test_config_fields_match_reference_defaults():
    cfg, _, _ = builder.export_from_registry("yolo8_det")
    assert (cfg.arch.model_type, cfg.arch.input_size) == ("n", (3, 640, 640))
    assert (cfg.dataset.num_classes, cfg.dataset.dataset_name) == (80, "coco")
    t = cfg.train
    assert (t.batch_size, t.initial_lr, t.epoch, t.warmup_iters, t.milestones, t.gamma) == (8, 1e-3, 100, 0, [], 0.1)
    assert t.mixed_precision is True and t.num_workers == 0 and t.resume_training == "" and t.last_epoch == -1
    assert (cfg.loss.box, cfg.loss.cls, cfg.loss.dfl) == (7.5, 0.5, 1.5)
    assert cfg.optimizer.name == "Adam" and cfg.log.print_interval == 50
    d = cfg.decode
    assert (d.conf_threshold, d.nms_threshold, d.max_det, d.letterbox_image) == (0.25, 0.7, 300, True)


# ---- model object contract -------------------------------------------------------------------------
@pytest.fixture(scope="module")
def model():
    cfg, algo, _ = builder.export_from_registry("yolo8_det")
    torch.manual_seed(0)
    m, name = algo(cfg, torch.device("cpu")).build_model()
    assert name == "YOLOv8n"
    return m


def test_state_dict_is_the_references(model):
    sd = model.state_dict()
    ref = O.init_state_dict("n", 80, seed=0)       # pinned bit-exact against the reference in make_golden.py
    assert list(sd.keys()) == list(ref.keys()) and len(sd) == 355
    for k in ref:
        assert sd[k].shape == ref[k].shape and sd[k].dtype == ref[k].dtype, k
        assert torch.equal(sd[k], ref[k]), k
    det = model.model[-1]
    assert (det.nc, det.no, det.reg_max) == (80, 144, 16) and det.stride.tolist() == [8.0, 16.0, 32.0]
    assert sum(p.numel() for p in model.parameters()) == 3157200
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 3157184


def test_parameters_are_views_of_the_flat_arenas(model):
    flat = model.flat_params
    w = dict(model.named_parameters())["model.22.cv3.1.0.conv.weight"]
    assert w.shape == (80, 128, 3, 3) and w.stride() == (9 * 128, 1, 3 * 128, 128)    # [cout][kh][kw][cin] storage
    before = float(flat.sum())
    with torch.no_grad():
        w.add_(1.0)
    assert abs(float(flat.sum()) - before - w.numel()) < 1.0
    with torch.no_grad():
        w.sub_(1.0)
    # fused head conv: box and class branch weights are adjacent in the arena
    lay = model.layout
    a, b = lay.slots["model.22.cv2.1.0.conv.weight"], lay.slots["model.22.cv3.1.0.conv.weight"]
    assert b.offset == a.offset + 64 * 9 * 128


def test_load_state_dict_roundtrip(model):
    ref = O.init_state_dict("n", 80, seed=123)
    missing = model.load_state_dict(ref, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    for k, v in model.state_dict().items():
        assert torch.equal(v, ref[k]), k
    model.load_state_dict(O.init_state_dict("n", 80, seed=0))


def test_no_cpu_fallback(model):
    with pytest.raises(CvxError):
        model(torch.zeros(1, 3, 64, 64))
    with pytest.raises(CvxError):
        model.model[0](torch.zeros(1, 3, 64, 64))


def test_other_scales_have_reference_parameter_counts():
    from computervision.pytorch_amd.graph import ParamLayout, build_yolov8_graph
    # yolo_v8.py:116: YOLOv8s 11166560 parameters (incl. the 16 frozen DFL weights)
    lay = ParamLayout("s", 80)
    n = sum(int(torch.tensor(s.shape).prod()) for s in lay.slots.values() if s.trainable)
    assert n + 16 == 11166560
    g = build_yolov8_graph(lay, 640, 640)
    assert g.anchors == 8400 and g.level_hw == [(80, 80), (40, 40), (20, 20)]
    with pytest.raises(ValueError):
        build_yolov8_graph(lay, 100, 100)
    with pytest.raises(ValueError):
        ParamLayout("q", 80)


# ---- the C-ABI shared library ------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB_PATH), "build the HIP library first (python __graft_entry__.py)"
    header = open(os.path.join(ROOT, "include", "cvx_engine.h")).read()
    declared = set(re.findall(r"\b(cvx_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    lib = ctypes.CDLL(LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/cvx_engine.h but not exported"
    assert declared == set(L.PROTOTYPES), declared ^ set(L.PROTOTYPES)
    assert L.load().cvx_abi_version() == L.ABI_VERSION


def test_release_library_does_not_carry_the_experimental_entry_points():
    """include/cvx_engine_experimental.h (the chain kernel's unit entries) is the tuning build's: the release library must not export them,
    and the bindings list them apart from the release prototypes"""
    if os.path.realpath(LIB_PATH) != os.path.realpath(os.path.join(ROOT, "computervision.pytorch_amd", "lib", "libcvx_engine.so")):
        pytest.skip("CVX_LIB points at another build")
    header = open(os.path.join(ROOT, "include", "cvx_engine_experimental.h")).read()
    declared = set(re.findall(r"\b(cvx_[a-z0-9_]+)\s*\(", header)) - set(L.PROTOTYPES)   # (its comment mentions release entry points)
    assert declared == set(L.EXPERIMENTAL_PROTOTYPES) and declared
    lib = ctypes.CDLL(LIB_PATH)
    for name in declared:
        assert not hasattr(lib, name), f"{name} is experimental but exported by the release library"
        assert name not in L.PROTOTYPES


def test_engine_refuses_cpu_device():
    from computervision.pytorch_amd.engine import Engine
    from computervision.pytorch_amd.graph import ParamLayout, build_yolov8_graph
    with pytest.raises(CvxError):
        Engine(build_yolov8_graph(ParamLayout("n", 80), 64, 64), torch.device("cpu"))


def test_deeplab_plugin_surface_cpu():
    """deeplabv3plus resolves through the reference's registry contract; the model holds the reference's 674 state_dict entries
    (58.75 M parameters) and refuses to run without the MI355X engine; the VOC palette is the reference's."""
    from computervision.pytorch_amd import _lib as L
    from core.algorithms.segmentation_2d import voc_colormap
    cfg, algo_cls, trainer_cls = builder.export_from_registry("deeplabv3plus")
    assert cfg.dataset.num_classes == 21 and cfg.arch.input_size == (3, 513, 513) and cfg.arch.output_stride == 16
    algo = algo_cls(cfg, torch.device("cpu"))
    model, name = algo.build_model()
    assert name == "deeplabv3plus" and len(model.state_dict()) == 674
    assert sum(p.numel() for p in model.parameters()) == 58753973
    cm = voc_colormap()
    assert cm[0] == (0, 0, 0) and cm[1] == (128, 0, 0) and cm[15] == (192, 128, 128) and cm[20] == (0, 64, 128) and len(cm) == 21
    with pytest.raises(L.CvxError):
        model.eval()(torch.zeros(1, 3, 65, 65))
    # training surface: the reference's loss choices (segmentation_2d.py:59-64), no CPU path for either the model or the loss
    crit = algo.build_loss()
    assert type(crit).__name__ == "SegLoss" and crit.mode == 0 and (crit.alpha, crit.gamma, crit.ignore_index) == (0.25, 2.0, -100)
    cfg.loss.loss_type = "ce"
    assert algo_cls(cfg, torch.device("cpu")).build_loss().mode == 1
    with pytest.raises(L.CvxError):
        model.train()(torch.zeros(2, 3, 65, 65))
    with pytest.raises(L.CvxError):
        crit.op(torch.zeros(1, 4, 24), torch.zeros(1, 8, 8, dtype=torch.long), (2, 2), 1.0)
    assert all(p.requires_grad for p in model.parameters()) and model.loss_scale == 65536.0 and model.dropout_p == 0.1
    g = model.flat_grads
    model.attach_grads()
    assert dict(model.named_parameters())["classifier.classifier.3.bias"].grad.data_ptr() == g[model.layout.convs["classifier.classifier.3"]["bias_off"]:].data_ptr()


def test_segmentation_metrics_and_synthetic_loader():
    """SegmentationMetrics (core/metrics/seg_metrics.py:4-44) on a hand-made confusion matrix, labels outside [0, nc) dropped;
    the synthetic loader's batch contract (images in [0, 1), int64 targets with ignored pixels)."""
    from core.trainer.segmentation_trainer import SegmentationMetrics, SyntheticSegmentationLoader
    m = SegmentationMetrics(3)
    gt = torch.tensor([[0, 0, 1, 1, 2, 2, -100, 255]])
    pr = torch.tensor([[0, 1, 1, 1, 2, 0, 0, 1]])
    m.add_batch(pr, gt)
    r = m.get_results()
    hist = np.array([[1, 1, 0], [0, 2, 0], [1, 0, 1]], dtype=float)
    assert np.array_equal(m.confusion_matrix.numpy(), hist)
    iu = np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist))
    assert abs(r["Overall Acc"] - 4 / 6) < 1e-12 and abs(r["Mean IoU"] - iu.mean()) < 1e-12
    assert abs(r["Mean Acc"] - np.mean(np.diag(hist) / hist.sum(1))) < 1e-12
    assert abs(r["FreqW Acc"] - float((hist.sum(1) / hist.sum() * iu).sum())) < 1e-12
    m.reset()
    assert float(m.confusion_matrix.sum()) == 0
    x, t = next(iter(SyntheticSegmentationLoader(2, (65, 97), 21, length=1, seed=4)))
    assert tuple(x.shape) == (2, 3, 65, 97) and tuple(t.shape) == (2, 65, 97) and t.dtype == torch.long
    assert 0 <= float(x.min()) and float(x.max()) < 1 and int(t.max()) <= 20 and int((t == -100).sum()) > 0 and int(t[t >= 0].min()) >= 0


def test_ssd_plugin_surface_cpu(gold_dir=os.path.join(os.path.dirname(__file__), "golden")):
    """ssd resolves through the registry; 136 state_dict entries / 26.29 M parameters; the prior boxes are the
    reference's (8732 x 4, checked against the oracle, itself asserted equal to the reference's in make_golden.py); no CPU path."""
    from oracle import ssd_ref as SS
    cfg, algo_cls, _ = builder.export_from_registry("ssd")
    algo = algo_cls(cfg, torch.device("cpu"))
    assert algo.anchors.shape == (8732, 4) and algo.anchors.dtype.name == "float32"
    assert (algo.anchors == SS.priors((300, 300))).all()
    model, name = algo.build_model()
    assert name == "SSD300_vgg" and len(model.state_dict()) == 136 and sum(p.numel() for p in model.parameters()) == 26293934
    with pytest.raises(L.CvxError):
        model.eval()(torch.zeros(1, 3, 300, 300))


def test_letterbox_geometry_and_oracle():
    """cvx_letterbox_geometry (host arithmetic of the library, no GPU needed) against the reference's letter_box lines
    (core/utils/image_process.py:56-62) restated in oracle/letterbox_ref.py, over a grid of odd sizes; the oracle's nearest resize
    and padding on a hand-checkable image; no CPU path for the pixel kernel."""
    from computervision.pytorch_amd.engine import letterbox_geometry, letterbox_u8
    from oracle import letterbox_ref as LB
    for (h, w) in [(480, 640), (375, 500), (1080, 1920), (1, 1), (333, 77), (640, 640), (641, 639), (2000, 3), (7, 1999)]:
        for (H, W) in [(640, 640), (300, 300), (513, 513), (512, 384)]:
            nh, nw, top, left, sc = LB.geometry(h, w, H, W)
            if nh <= 0 or nw <= 0:
                with pytest.raises(L.CvxError):
                    letterbox_geometry(h, w, H, W)
                continue
            assert letterbox_geometry(h, w, H, W) == (nh, nw, top, left, sc), (h, w, H, W)
    img = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)                 # 2 x 3 image -> 4 x 6 inside a 6 x 6 canvas
    out, scale, pads = LB.letter_box(img, (6, 6))
    assert scale == 2.0 and pads == [1, 1, 0, 0] and out.shape == (6, 6, 3)
    assert (out[0] == 128).all() and (out[5] == 128).all()
    assert np.array_equal(out[1:5, :, 0], np.repeat(np.repeat(img[:, :, 0], 2, 0), 2, 1))
    t = LB.to_tensor(out)
    assert t.dtype == np.float32 and t.shape == (3, 6, 6) and t[0, 0, 0] == np.float32(128) / np.float32(255)
    with pytest.raises(CvxError):
        letterbox_u8(torch.zeros(4, 4, 3, dtype=torch.uint8), torch.zeros(3, 8, 8))


def test_generic_grad_buckets_on_every_graph():
    """graph.generic_grad_buckets (the op ranges of the overlapped gradient exchange of the DeepLab / CenterNet / SSD / YOLOv7 steps): on
    each model's graph the ranges tile the op list from the last op down to 0, their parameter slices are disjoint and ordered, every
    op's parameters lie inside the slice of its own range, and for YOLOv8 they cover the arena exactly like its hand-made buckets."""
    from computervision.pytorch_amd import deeplab, dla, graph, ssd, yolov7
    from computervision.pytorch_amd.model import Yolo8
    graphs = {
        "deeplab": deeplab.build_deeplab_graph(deeplab.DeepLabLayout(21), 513, 513),
        "dla": dla.build_dla_graph(dla.DlaLayout(20), 384, 384),
        "yolov7": yolov7.build_yolov7_graph(yolov7.Yolo7Layout(20), 640, 640),
        "ssd": ssd.build_ssd_graph(ssd.SsdLayout(20), 300, 300),
    }
    m8 = Yolo8("n", 80)
    graphs["yolov8"] = graph.build_yolov8_graph(m8.layout, 640, 640)
    for name, g in graphs.items():
        for nb in (1, 4, 7):
            b = graph.generic_grad_buckets(g, nb)
            assert 1 <= len(b) <= nb and b[0][0] == len(g.ops) - 1 and b[-1][1] == 0, name
            for (hi, lo, p0, p1), nxt in zip(b, list(b[1:]) + [None]):
                assert hi >= lo and p1 >= p0
                if nxt is not None:
                    assert nxt[0] == lo - 1 and nxt[3] <= p0, (name, nb)
                for o in g.ops[lo:hi + 1]:
                    iv = graph.op_param_interval(o)
                    assert iv is None or (p0 <= iv[0] and iv[1] <= p1), (name, o["name"])
    own = graph.grad_buckets(graphs["yolov8"], m8.layout, 5)
    gen = graph.generic_grad_buckets(graphs["yolov8"], 5)
    assert gen[-1][2] == 0 and abs(gen[0][3] - own[0][3]) <= 4                # same arena extent (up to the 16-byte alignment of the last slot)


@pytest.mark.parametrize("H,W", [(640, 640), (320, 320), (128, 128), (96, 160)])
def test_raw_fp16_layer_rule_is_the_same_in_the_graph_and_in_the_oracle(H, W):
    """Which convs keep their raw output in fp16 during training (CVX_OPF_RAW_F16) is decided in graph.py; the oracle's fp16 emulation
    (yolov8_ref._raw_f16) has to round exactly those layers, or the engine-vs-emulation tests compare two different networks."""
    from computervision.pytorch_amd.graph import ParamLayout, build_yolov8_graph
    g = build_yolov8_graph(ParamLayout("n", 80), H, W)
    flagged, pixels = set(), {}
    for op in g.ops:
        if op["type"] != L.OP_CONV:
            continue
        pixels[op["name"]] = op["oh"] * op["ow"]
        if op.get("flags", 0) & L.OPF_RAW_F16:
            flagged.add(op["name"])

    def oracle_prefix(name):       # engine op name -> the state_dict prefix of the oracle's _unit
        part = name.split(".")
        if part[0] == "22":        # Detect: "22.<lvl>.0" = cv2[lvl][0] + cv3[lvl][0] in one launch, "1b" / "1c" = cv2[lvl][1] / cv3[lvl][1]
            return None
        if len(part) == 3 and part[1].startswith("m"):
            return f"model.{part[0]}.m.{part[1][1:]}.{part[2]}"
        return "model." + name

    assert "0" not in flagged                                  # the fp32 stem never
    for name, px in pixels.items():
        p = oracle_prefix(name)
        if p is None:
            if not name.endswith(("2b", "2c")):                # the head's BN convs: always (module 22); its bias convs have no BatchNorm
                assert name in flagged, name
            continue
        res = ".m" in name and name.endswith("cv2") and int(name.split(".")[0]) < 10    # backbone Bottlenecks add their input
        assert (name in flagged) == (O._raw_f16(p, px) and p != "model.0" and not res), (name, px)
    if H * W <= 160 * 160:                                       # the fixtures' shapes: no backbone layer (outputs below 80 x 80)
        assert all(int(n.split(".")[0]) >= 12 for n in flagged)
    else:
        assert {"1", "2.cv1", "2.cv2"} <= flagged and "2.m0.cv1" not in flagged
