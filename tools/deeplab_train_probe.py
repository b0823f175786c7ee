"""Probe: one DeepLabv3+ training step on the engine against the CPU oracle (fp32 and fp16-storage emulation)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import deeplab_ref as D
from computervision.pytorch_amd.deeplab import DeepLabV3PlusR101, SegLoss

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = DeepLabV3PlusR101(21, dropout_p=0.0)
with torch.no_grad():
    for k, v in m.state_dict().items():
        if k.endswith(".bn3.weight"):
            v.fill_(0.1)
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
m = m.to(dev).train()
B, H, W = 2, 97, 129
g = torch.Generator().manual_seed(77)
x = torch.rand(B, 3, H, W, generator=g)
t = torch.randint(0, 21, (B, H, W), generator=g)
t[torch.rand(B, H, W, generator=g) < 0.1] = -100
crit = SegLoss("focal")
out = m(x.to(dev))
loss = crit(out, t.to(dev))
loss.backward()
torch.cuda.synchronize()
print("engine loss", float(loss))
res = {}
for fp16 in (False, True):
    sd = {k: v.clone() for k, v in sd0.items()}
    D.FP16_STORAGE[0] = fp16
    l, grads, rows = D.loss_and_grads(sd, x, t)
    D.FP16_STORAGE[0] = False
    res[fp16] = (l, grads, rows, sd)
    print("oracle fp16=%s loss %.6f" % (fp16, float(l)))
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
rows_e = m.last_rows[..., :21].reshape(B, *m._last_engine.graph.level_hw[0], 21).cpu()
print("rows: eng-vs-ref %.3e eng-vs-emu %.3e emu-vs-ref %.3e" % (rel(rows_e, res[False][2]), rel(rows_e, res[True][2]), rel(res[True][2], res[False][2])))
eg = {k: p.grad.cpu() for k, p in m.named_parameters()}
def tot(ga, gb):
    num = sum(float((ga[k].double() - gb[k].double()).pow(2).sum()) for k in gb)
    den = sum(float(gb[k].double().pow(2).sum()) for k in gb)
    return (num / den) ** 0.5
print("grads total: eng-vs-ref %.3e eng-vs-emu %.3e emu-vs-ref %.3e" % (tot(eg, res[False][1]), tot(eg, res[True][1]), tot(res[True][1], res[False][1])))
for k in ["classifier.classifier.3.bias", "classifier.classifier.3.weight", "classifier.classifier.1.weight", "classifier.classifier.0.weight",
          "classifier.aspp.project.0.weight", "classifier.aspp.convs.4.1.weight", "classifier.aspp.convs.1.0.weight", "classifier.project.0.weight",
          "backbone.layer4.2.conv3.weight", "backbone.layer4.0.downsample.0.weight", "backbone.layer3.22.conv2.weight", "backbone.layer3.0.downsample.0.weight",
          "backbone.layer2.0.conv2.weight", "backbone.layer1.0.conv1.weight", "backbone.bn1.weight", "backbone.conv1.weight"]:
    print("%-45s eng-ref %.3e eng-emu %.3e emu-ref %.3e  |g| %.3e" % (k, rel(eg[k], res[False][1][k]), rel(eg[k], res[True][1][k]), rel(res[True][1][k], res[False][1][k]), float(res[False][1][k].norm())))
# running statistics
sdm = {k: v.cpu() for k, v in m.state_dict().items()}
worst = max((rel(sdm[k], res[False][3][k]), k) for k in sdm if k.endswith("running_var") or k.endswith("running_mean"))
print("running stats worst", worst)
