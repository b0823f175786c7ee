"""Host-side cost of one eager train step with / without an initialised RCCL process group (diagnostic)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from computervision.pytorch_amd.model import Yolo8
from computervision.pytorch_amd.train import FlatAdam, FusedTrainStep, V8DetectionLoss
from configs import Yolo8DetConfig
from oracle import synth
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
use_dist = len(sys.argv) > 1 and sys.argv[1] == "dist"
if use_dist:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.manual_seed(0)
m = Yolo8("n", 80).to(dev).train()
step = FusedTrainStep(m, V8DetectionLoss(Yolo8DetConfig(), m), FlatAdam(m), n_buckets=1)
x = synth.images(32, 640, 640, seed=1).to(dev); batch = {k: v.to(dev) for k, v in synth.targets(32, seed=2).items()}
for _ in range(3): step(x, batch)
torch.cuda.synchronize()
host = []; t0 = time.perf_counter()
for _ in range(10):
    a = time.perf_counter(); step(x, batch); host.append(time.perf_counter() - a)
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print(f"dist={use_dist} distributed_step={step.distributed}: host {1e3*sum(host)/10:.2f} ms/step, wall {1e3*tot/10:.2f} ms/step")
if use_dist: dist.destroy_process_group()
