import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.getcwd())
from computervision.pytorch_amd import _lib as L
lib = L.load(); dev = torch.device("cuda", 0)
def run(B,H,W,Ci,Co,seed=0):
    g = torch.Generator().manual_seed(seed)
    x16 = torch.randn(B, Ci, H, W, generator=g).half()
    w16 = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).half()
    conv = F.conv2d(x16.float(), w16.float(), None, 1, 1)
    xd, wd = x16.permute(0, 2, 3, 1).contiguous().to(dev), w16.permute(0, 2, 3, 1).contiguous().to(dev)
    out = torch.empty(B, H, W, Co, dtype=torch.float16, device=dev)
    res=[]
    for rep in range(3):
        out.zero_()
        L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, 3, 1, 1, 1, 0x2000, None, None, L.ptr(out), L.stream_ptr(dev)), "plain")
        got = out.float().permute(0, 3, 1, 2).cpu()
        bad = (got - conv).abs() > 2e-2 * conv.abs().max()
        res.append(int(bad.sum()))
    return res
for (B,H,Ci,Co) in [(1,2,32,16),(1,4,64,16),(2,20,128,128)]:
    for W in [8,12,15,16,17,18,19,20,21,22,23,24,28,32,33,36,40,44,48]:
        print((B,H,W,Ci,Co), run(B,H,W,Ci,Co), flush=True)
