"""Probe: YOLOv8-n train step and eval forward with the engine's launch stream = the legacy default stream, a torch pool stream, or a raw
HIP stream of the probe's own (non-blocking / blocking).  One configuration per process (argv[1]: default | pool | raw | rawblocking)."""
import ctypes as C
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from computervision.pytorch_amd.model import Yolo8
from computervision.pytorch_amd.train import FlatAdam, FusedTrainStep, V8DetectionLoss
from configs import Yolo8DetConfig
from computervision.pytorch_amd import synth

mode = sys.argv[1] if len(sys.argv) > 1 else "default"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if mode == "pool":
    torch.cuda.set_stream(torch.cuda.Stream())
elif mode in ("raw", "rawblocking"):
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    s = C.c_void_p()
    rc = hip.hipStreamCreateWithFlags(C.byref(s), 1 if mode == "raw" else 0)   # hipStreamNonBlocking = 1
    assert rc == 0 and s.value
    torch.cuda.set_stream(torch.cuda.ExternalStream(s.value, device=dev))
torch.manual_seed(0)
if len(sys.argv) > 2 and sys.argv[2] == "second_engine":      # another engine (another input size) has run in this process before
    m0 = Yolo8("n", 80).to(dev).train()
    step0 = FusedTrainStep(m0, V8DetectionLoss(Yolo8DetConfig(), m0), FlatAdam(m0))
    x0 = synth.images(2, 128, 128, seed=1).to(dev)
    b0 = {k: v.to(dev) for k, v in synth.targets(2, seed=2).items()}
    for _ in range(3):
        step0(x0, b0)
    torch.cuda.synchronize()
    mode += "+2nd engine"
m = Yolo8("n", 80).to(dev).train()
step = FusedTrainStep(m, V8DetectionLoss(Yolo8DetConfig(), m), FlatAdam(m))
x = synth.images(32, 640, 640, seed=1).to(dev)
batch = {k: v.to(dev) for k, v in synth.targets(32, seed=2).items()}
for _ in range(5):
    step(x, batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step(x, batch)
torch.cuda.synchronize()
train_ms = (time.perf_counter() - t0) / 20 * 1e3
m.eval()
with torch.no_grad():
    for _ in range(5):
        m._run_forward(x, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        m._run_forward(x, False)
    torch.cuda.synchronize()
eval_ms = (time.perf_counter() - t0) / 30 * 1e3
print(f"launch stream {mode:22s}: train step {train_ms:7.3f} ms   eval forward {eval_ms:6.3f} ms", flush=True)
