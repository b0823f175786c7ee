"""Probe: YOLOv7-l train-mode forward + backward on the engine against the CPU oracle (fp32 and fp16-storage emulation)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import yolov7_ref as Y7
from computervision.pytorch_amd.yolov7 import Yolo7L
dev = torch.device("cuda:0")
g = np.load("tests/golden/yolov7_train_160x224.npz")
torch.manual_seed(0)
m = Yolo7L(20)
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
m = m.to(dev).train()
x = torch.from_numpy(g["x"])
outs = m(x.to(dev))
weights = Y7.projection_weights([o.shape for o in outs], int(g["proj_seed"]))
loss = Y7.projection_loss(outs, [w.to(dev) for w in weights])
loss.backward()
torch.cuda.synchronize()
print("engine loss", float(loss.detach()), "ref", float(g["loss"]))
res = {}
for fp16 in (False, True):
    Y7.FP16_STORAGE[0] = fp16
    l, grads, o = Y7.loss_and_grads({k: v.clone() for k, v in sd0.items()}, x, weights)
    Y7.FP16_STORAGE[0] = False
    res[fp16] = (l, grads, o)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
for i in range(3):
    print("out", i, "eng-ref %.3e eng-emu %.3e emu-ref %.3e" % (rel(outs[i].detach().cpu(), res[False][2][i]), rel(outs[i].detach().cpu(), res[True][2][i]), rel(res[True][2][i], res[False][2][i])))
eg = {k: p.grad.cpu() for k, p in m.named_parameters()}
def tot(a, b):
    return (sum(float((a[k].double() - b[k].double()).pow(2).sum()) for k in b) / sum(float(b[k].double().pow(2).sum()) for k in b)) ** 0.5
print("grads total: eng-ref %.3e eng-emu %.3e emu-ref %.3e" % (tot(eg, res[False][1]), tot(eg, res[True][1]), tot(res[True][1], res[False][1])))
ks = list(eg.keys())
for k in ks[::28] + ["yolo_head_P3.weight", "rep_conv_1.rbr_dense.1.weight"]:
    print("%-45s eng-ref %.3e emu-ref %.3e |g| %.3e" % (k, rel(eg[k], res[False][1][k]), rel(res[True][1][k], res[False][1][k]), float(res[False][1][k].norm())))
