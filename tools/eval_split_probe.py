"""Probe: YOLOv8-n eval forward of one batch as TWO half-batch chains on two streams (two engines over the same parameter arenas)
against the single chain.  Eval-mode BatchNorm is folded, so the images of a batch are independent.  python tools/eval_split_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from computervision.pytorch_amd.model import Yolo8, BN_EPS, BN_MOMENTUM
from computervision.pytorch_amd.engine import Engine
from computervision.pytorch_amd.graph import build_yolov8_graph
from computervision.pytorch_amd import synth

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Yolo8("n", 80).to(dev).eval()
GF = 8.7
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = synth.images(B, 640, 640, seed=1).to(dev)


def timed(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def report(tag, ms):
    print(f"{tag:44s} {ms:7.3f} ms  {B / ms * 1e3:9.1f} img/s  {B * GF / ms / 2516.6 * 100:5.2f} % of MFMA peak", flush=True)


with torch.no_grad():
    ref = m._run_forward(x, False).clone()
    report("one chain", timed(lambda: m._run_forward(x, False)))
    for half in (B // 2,):
        report(f"one chain at batch {half}", timed(lambda: m._run_forward(x[:half], False)) * 1.0)
    m._run_forward(x, False)

    def make_engine():
        e = Engine(build_yolov8_graph(m.layout, 640, 640), dev)
        e.set_bn(BN_EPS, BN_MOMENTUM)
        e.bind(m.flat_params, m.flat_grads, m.flat_stats)
        return e

    for prio, nsplit in ((-1, 2), (0, 2), (-1, 4)):
        engs = [make_engine() for _ in range(nsplit)]
        streams = [None] + [torch.cuda.Stream(device=dev, priority=prio) for _ in range(nsplit - 1)]
        pred = torch.empty_like(ref) if ref.is_contiguous() else torch.empty(ref.shape[0], ref.shape[1], ref.stride(1), device=dev)
        full = torch.empty(B, ref.shape[1], ref.stride(1), device=dev, dtype=torch.float32)
        h = B // nsplit
        ev_fork = torch.cuda.Event()
        ev_join = [torch.cuda.Event() for _ in range(nsplit - 1)]

        def split():
            cur = torch.cuda.current_stream(dev)
            ev_fork.record(cur)
            for k in range(1, nsplit):
                s = streams[k]
                s.wait_event(ev_fork)
                with torch.cuda.stream(s):
                    engs[k].forward(x[k * h:(k + 1) * h], False, full[k * h:(k + 1) * h])
                    ev_join[k - 1].record(s)
            engs[0].forward(x[:h], False, full[:h])
            for k in range(1, nsplit):
                cur.wait_event(ev_join[k - 1])

        split()
        torch.cuda.synchronize()
        got = full[..., :ref.shape[2]]
        print(f"  max |split - one chain| = {float((got - ref).abs().max()):.3e}", flush=True)
        report(f"{nsplit} chains, side streams priority {prio}", timed(split))
