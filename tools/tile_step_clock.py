"""Tuning build (CVX_LIB=build/libcvx_tuning.so CVX_TILE_DBG=256): shader-clock stamps of workgroup 0's wave 0 at the top of every K-step of the
row-band kernel.   python tools/tile_step_clock.py B H W Cin Cout"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from computervision.pytorch_amd import _lib as L  # noqa: E402
B, H, W, Ci, Co = [int(v) for v in sys.argv[1:6]]
lib = L.load(); dev = torch.device("cuda", 0)
x = torch.randn(B, H, W, Ci).half().to(dev); w = (torch.randn(Co, 3, 3, Ci) * 0.05).half().to(dev)
out = torch.empty(B, H, W, Co, dtype=torch.float16, device=dev)
dump = torch.zeros(1 << 16, dtype=torch.int64, device=dev)
for rep in range(3):
    dump.zero_()
    lib.cvx_debug_clock_buffer(L.ptr(dump))
    L.check(lib.cvx_conv2d_nhwc(L.ptr(x), B, H, W, Ci, L.ptr(w), Co, 3, 1, 1, 1, 0x2000, None, None, L.ptr(out), L.stream_ptr(dev)), "plain")
    lib.cvx_debug_clock_buffer(None); torch.cuda.synchronize()
    d = dump.cpu().numpy()[64:64 + 9 * ((Ci + 31) // 32) + 2]
    d = d[d != 0]
    print("rep", rep, "steps", len(d) - 1, "cycles per step:", np.diff(d).tolist())
