"""Per-op timing of a train step (YOLOv8-n 640x640 batch 32, or `deeplab`: DeepLabv3+ R101 513x513 batch 16): joins
cvx_engine_profile_dump records with the op list and prints, per (class, op), the mean time, algorithmic TF/s and GB/s, and the
time the tighter of the two rooflines would allow.

    python tools/op_profile.py [steps] [yolov8|yolov8_eval|deeplab|yolo7|ssd|centernet] > gpurun_out/op_profile.txt
"""
import collections
import csv
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CLASSES = ("conv_fwd", "conv_dgrad", "conv_wgrad", "bn_fwd", "bn_bwd", "misc", "slab_reduce")
PEAK_TF, PEAK_GB = 2516.6, 8000.0


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    if len(sys.argv) > 2 and sys.argv[2] == "deeplab":
        return report(*deeplab_step(), steps)
    if len(sys.argv) > 2 and sys.argv[2] in ("yolo7", "ssd", "centernet"):
        return report(*trainer_step(sys.argv[2]), steps)
    if len(sys.argv) > 2 and sys.argv[2] == "yolov8_eval":
        return report(*yolov8_eval(), steps)
    from computervision.pytorch_amd.model import Yolo8
    from computervision.pytorch_amd.train import FlatAdam, FusedTrainStep, V8DetectionLoss
    from configs import Yolo8DetConfig
    from oracle import synth
    dev = torch.device("cuda", 0)
    cfg = Yolo8DetConfig()
    torch.manual_seed(0)
    model = Yolo8(os.environ.get("OP_PROFILE_SCALE", "n"), 80, loss_scale=cfg.engine.loss_scale).to(dev).train()  # OP_PROFILE_SCALE=s: YOLOv8-s
    step = FusedTrainStep(model, V8DetectionLoss(cfg, model), FlatAdam(model, lr=1e-3))
    B = 32
    x = synth.images(B, 640, 640, seed=1).to(dev)
    batch = {k: v.to(dev) for k, v in synth.targets(B, seed=2).items()}
    for _ in range(3):
        step(x, batch)
    torch.cuda.synchronize()
    report(model._last_engine, lambda: step(x, batch), steps)


def yolov8_eval():
    """eval-mode forward of YOLOv8-n, batch 32, 640x640 (the forward north_star quotes its MFMA fraction on)"""
    from computervision.pytorch_amd.model import Yolo8
    from computervision.pytorch_amd import synth
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = Yolo8("n", 80).to(dev).eval()
    x = synth.images(32, 640, 640, seed=1).to(dev)
    with torch.no_grad():
        for _ in range(3):
            model._run_forward(x, False)
    torch.cuda.synchronize()

    def run():
        with torch.no_grad():
            model._run_forward(x, False)
    return model._last_engine, run


def deeplab_step():
    import builder
    from core.trainer.segmentation_trainer import SyntheticSegmentationLoader
    dev = torch.device("cuda", 0)
    cfg, _, trainer_cls = builder.export_from_registry("deeplabv3plus")
    torch.manual_seed(0)
    tr = trainer_cls(cfg, dev, dataloader=SyntheticSegmentationLoader(16, (513, 513), 21, length=1, seed=1))
    tr.model.train()
    images, targets = next(iter(tr.train_dataloader))
    batch = (images.to(dev), targets.to(dev))
    for _ in range(3):
        tr.train_loop(batch, None)
    torch.cuda.synchronize()
    return tr.model._last_engine, lambda: tr.train_loop(batch, None)


def trainer_step(name):
    """the fused train step of a registered trainer on one synthetic batch of its default configuration"""
    import builder
    dev = torch.device("cuda", 0)
    cfg, _, trainer_cls = builder.export_from_registry(name)
    cfg.train.pretrained = False
    if name in ("yolo7", "ssd"):
        cfg.train.batch_size = 32
    torch.manual_seed(0)
    tr = trainer_cls(cfg, dev)
    tr.model.train()
    batch = next(iter(tr.train_dataloader))
    for _ in range(3):
        tr.train_loop(batch, None)
    torch.cuda.synchronize()
    return tr.model._last_engine, lambda: tr.train_loop(batch, None)


def report(eng, run, steps):
    eng.profile(True)
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    path = os.path.join(ROOT, "gpurun_out", "op_profile.csv")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    eng.profile_dump(path)
    eng.profile(False)

    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = (int(r["class"]), int(r["op"]))
        a = agg.setdefault(k, [0.0, 0.0, 0.0, 0])
        a[0] += float(r["ms"]); a[1] = float(r["flops"]); a[2] = float(r["bytes"]); a[3] += 1
    ops = eng.graph.ops
    tot = collections.defaultdict(float)
    ideal = collections.defaultdict(float)
    print(f"{'class':10s} {'op':>3s} {'shape':58s} {'us':>8s} {'TF/s':>7s} {'GB/s':>7s} {'roof us':>8s} {'frac':>6s}")
    for (cls, op), (ms, fl, by, n) in agg.items():
        per = n // steps if n >= steps else 1          # dgrad of a stride-2 conv: 4 launches per step under one key
        us = ms * 1e3 / steps
        fl, by = fl * per, by * per
        roof = max(fl / (PEAK_TF * 1e12), by / (PEAK_GB * 1e9)) * 1e6
        shape = ""
        if op >= 0:
            o = ops[op]
            if o["type"] == 1:
                shape = f"{o['name'][-28:]:12s} {o['ih']}x{o['iw']} {o['w_cin']}->{o['out'][2]} k{o['k']}s{o['stride']}" + (f"d{o['dil']}" if o.get("dil", 1) > 1 else "") + (" +res" if "res" in o else "")
            else:
                shape = f"{o['name'][-28:]:12s} {o['ih']}x{o['iw']} c{o['out'][2]}"
        tot[cls] += us
        ideal[cls] += roof
        print(f"{CLASSES[cls]:10s} {op:3d} {shape:58s} {us:8.1f} {fl / us / 1e6:7.1f} {by / us / 1e3:7.0f} {roof:8.1f} {roof / us:6.2f}")
    print()
    for c in sorted(tot):
        print(f"{CLASSES[c]:12s} {tot[c]:8.1f} us   roofline {ideal[c]:8.1f} us   frac {ideal[c] / tot[c]:.3f}")
    print(f"{'all':12s} {sum(tot.values()):8.1f} us   roofline {sum(ideal.values()):8.1f} us")


if __name__ == "__main__":
    main()
