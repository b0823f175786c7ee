"""Probe: eval-mode forward of each model family, eager launches vs one hipGraph replay (torch.cuda.CUDAGraph capture of the engine's
launch sequence).  python tools/eval_graph_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import builder
from computervision.pytorch_amd import synth

dev = torch.device("cuda:0")


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, B, hw in (("yolo8_det", 32, (640, 640)), ("centernet", 64, (512, 512)), ("yolo7", 32, (640, 640)), ("ssd", 32, (300, 300)), ("deeplabv3plus", 16, (513, 513))):
    cfg, algo_cls, _ = builder.export_from_registry(name)
    if name == "yolo7":
        cfg.train.pretrained = False
    torch.manual_seed(0)
    model, _ = algo_cls(cfg, dev).build_model()
    model = model.to(dev).eval()
    x = synth.images(B, *hw, seed=1).to(dev)
    run = (lambda: model._run_forward(x, False)) if hasattr(model, "_run_forward") else (lambda: model(x))
    with torch.no_grad():
        eager = timeit(run)
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = run()
        graphed = timeit(g.replay)
    print(f"{name:14s} B={B:3d} eager {eager:7.3f} ms  graph {graphed:7.3f} ms  ({B / eager * 1e3:8.1f} -> {B / graphed * 1e3:8.1f} img/s)", flush=True)
