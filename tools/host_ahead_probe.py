"""Is the train step's wall time the GPU's, or does the host's enqueue rate show?  (diagnostic)

A long device-side sleep lets the host queue K whole steps before the GPU starts the first; events around those K steps then time the GPU
alone.  Compared with the free-running loop: equal = the host is ahead in steady state and every gap in a kernel trace that is not a
dependency is the profiler's.
"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from computervision.pytorch_amd.model import Yolo8
from computervision.pytorch_amd.train import FlatAdam, FusedTrainStep, V8DetectionLoss
from configs import Yolo8DetConfig
from oracle import synth

dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
torch.manual_seed(0)
m = Yolo8("n", 80).to(dev).train()
step = FusedTrainStep(m, V8DetectionLoss(Yolo8DetConfig(), m), FlatAdam(m), n_buckets=1)
x = synth.images(32, 640, 640, seed=1).to(dev); batch = {k: v.to(dev) for k, v in synth.targets(32, seed=2).items()}
for _ in range(5): step(x, batch)
torch.cuda.synchronize()
K = 4
for rep in range(3):
    host = []
    t0 = time.perf_counter()
    for _ in range(20):
        a = time.perf_counter(); step(x, batch); host.append(time.perf_counter() - a)
    torch.cuda.synchronize(); loop = (time.perf_counter() - t0) / 20 * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(6e7))     # ~25-30 ms of GPU time: the host queues K steps meanwhile
    e0.record()
    a = time.perf_counter()
    for _ in range(K): step(x, batch)
    hq = (time.perf_counter() - a) / K * 1e3
    e1.record(); torch.cuda.synchronize()
    print(f"loop {loop:.4f} ms/step (host call {1e3 * sum(host) / 20:.3f} ms, first calls {[round(1e3 * h, 2) for h in host[:3]]}); "
          f"behind a sleep: {e0.elapsed_time(e1) / K:.4f} ms/step (host enqueue {hq:.3f} ms/step)", flush=True)
