"""Saves the flat gradient of one train step (for A/B comparisons between builds / knobs).
   python tools/grad_dump.py out.pt [fuse 0|1]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
L = importlib.import_module("computervision.pytorch_amd._lib")
synth = importlib.import_module("computervision.pytorch_amd.synth")
from computervision.pytorch_amd.model import Yolo8
from computervision.pytorch_amd.train import V8DetectionLoss
from configs import Yolo8DetConfig
dev = torch.device("cuda:0")
B, H, W = 4, 160, 160
x, batch = synth.images(B, H, W, seed=5).to(dev), {k: v.to(dev) for k, v in synth.targets(B, seed=6).items()}
torch.manual_seed(0)
m = Yolo8("n", 80).to(dev).train()
crit = V8DetectionLoss(Yolo8DetConfig(), m)
m.engine_for(H, W).set_option(L.OPT_FUSE_BN_BWD_STATS, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
m.flat_grads.zero_()
loss, _ = crit(m(x), batch)
loss.backward()
torch.cuda.synchronize()
torch.save(m.flat_grads.cpu(), sys.argv[1])
