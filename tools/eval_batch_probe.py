"""Probe: YOLOv8-n eval forward (engine only, fp32 pred rows) throughput vs batch size.  python tools/eval_batch_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from computervision.pytorch_amd.model import Yolo8
from computervision.pytorch_amd import synth
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Yolo8("n", 80).to(dev).eval()
GF = 8.7
for B in (32, 64, 128, 256, 512):
    x = synth.images(B, 640, 640, seed=1).to(dev)
    with torch.no_grad():
        for _ in range(3):
            m._run_forward(x, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = max(3, 2048 // B)
        for _ in range(n):
            m._run_forward(x, False)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
    print(f"B={B:4d} {ms:8.3f} ms  {B / ms * 1e3:9.1f} img/s  {B * GF / ms:7.1f} TFLOP/s  {B * GF / ms / 2516.6 * 100:5.2f} % of MFMA peak", flush=True)
    del x
