"""Probe: does a hipGraph of ONE stream's dependent kernel chain shorten the launch gaps?

Runs the YOLOv8-n batch-32 eval forward (63 conv launches + pools / upsamples, no BatchNorm passes) and -- with `--train` -- the train
step, eager vs torch.cuda.CUDAGraph replay.  Meant for the TUNING library with the auxiliary streams switched off, so that the capture is a
single-stream chain:

    CVX_LIB=build/libcvx_tuning.so CVX_LANES=0 CVX_PACK_LANE=0 python tools/graph_main_chain_probe.py
    CVX_LIB=build/libcvx_tuning.so CVX_LANES=0 CVX_PACK_LANE=0 CVX_TUNE_SKIP_WGRAD=1 python tools/graph_main_chain_probe.py --train
"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from computervision.pytorch_amd import synth
from computervision.pytorch_amd.model import Yolo8
from computervision.pytorch_amd.train import FlatAdam, FusedTrainStep, V8DetectionLoss
from configs import Yolo8DetConfig

dev = torch.device("cuda:0")


def timeit(fn, n=40, rounds=3):
    best = 1e9
    for _ in range(rounds):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


cfg = Yolo8DetConfig()
torch.manual_seed(0)
B = 32
x = synth.images(B, 640, 640, seed=1).to(dev)
env = {k: os.environ.get(k) for k in ("CVX_LIB", "CVX_LANES", "CVX_PACK_LANE", "CVX_TUNE_SKIP_WGRAD", "GPU_MAX_HW_QUEUES")}
print("env", env, flush=True)
if "--train" not in sys.argv:
    model = Yolo8("n", 80).to(dev).eval()
    with torch.no_grad():
        run = lambda: model._run_forward(x, False)
        eager = timeit(run)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run()
        graphed = timeit(g.replay)
    print(f"eval forward B={B}: eager {eager:.4f} ms  graph {graphed:.4f} ms  ({(graphed - eager) * 1e3:+.1f} us)", flush=True)
else:
    model = Yolo8("n", 80, loss_scale=cfg.engine.loss_scale).to(dev).train()
    crit = V8DetectionLoss(cfg, model)
    batch = {k: v.to(dev) for k, v in synth.targets(B, seed=2).items()}
    step = FusedTrainStep(model, crit, FlatAdam(model, lr=1e-3))
    eager = timeit(lambda: step(x, batch), n=20)
    print(f"train step B={B}: eager {eager:.4f} ms", flush=True)
    os.environ.pop("GPU_MAX_HW_QUEUES", None)   # (the guard in FusedTrainStep reads the variable; the runtime has long read it)
    step2 = FusedTrainStep(model, crit, FlatAdam(model, lr=1e-3), use_graph=True)
    graphed = timeit(lambda: step2(x, batch), n=20)
    print(f"train step B={B}: graph {graphed:.4f} ms  ({(graphed - eager) * 1e3:+.1f} us)", flush=True)
