"""One guarded launch of the GEMM-shaped conv kernel with the debug range checks on (cvx_debug_clock_buffer): prints what was out of range."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from computervision.pytorch_amd import _lib as L
lib = L.load()
dev = torch.device("cuda", 0)
B, H, W, Ci, Co, k, s = 8, 19, 19, 512, 512, 3, 1
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, Ci, generator=g).half().to(dev)
w = (torch.randn(Co, k, k, Ci, generator=g) * (k * k * Ci) ** -0.5).half().to(dev)
out = torch.zeros(B, H, W, Co, dtype=torch.float16, device=dev)
dbg = torch.zeros(64, dtype=torch.int64, device=dev)
lib.cvx_debug_clock_buffer(L.ptr(dbg))
L.check(lib.cvx_conv2d_nhwc(L.ptr(x), B, H, W, Ci, L.ptr(w), Co, k, s, k // 2, 1, 0, None, None, L.ptr(out), L.stream_ptr(dev)), "conv")
torch.cuda.synchronize()
lib.cvx_debug_clock_buffer(None)
print("debug words:", dbg.cpu().tolist()[:20])
import torch.nn.functional as F
ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
print("rel err", float((out.float().cpu() - ref).abs().max() / ref.abs().max()))
