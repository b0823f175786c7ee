"""Tuning build (CVX_LIB=build/libcvx_tuning.so CVX_TILE_DBG=16): the LDS address every lane of workgroup 0 requests for the pixel fragment of
K-steps 1.. against the addressing rules.   python tools/tile_addr_dump.py H W Cin Cout"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from computervision.pytorch_amd import _lib as L  # noqa: E402

H, W, Ci, Co = [int(v) for v in sys.argv[1:5]]
B = 1
lib = L.load(); dev = torch.device("cuda", 0)
plan = (ctypes.c_int32 * 8)()
assert lib.cvx_debug_conv_tile_plan(B, H, W, Ci, Co, plan) == 0
TR, MT = plan[0], plan[1]
print("plan TR %d MT %d NTW %d NB %d" % tuple(plan[:4]))
x = torch.randn(B, H, W, Ci).half().to(dev); w = torch.randn(Co, 3, 3, Ci).half().to(dev)
out = torch.empty(B, H, W, Co, dtype=torch.float16, device=dev)
dump = torch.zeros(1 << 20, dtype=torch.int32, device=dev)
lib.cvx_debug_clock_buffer(L.ptr(dump))
L.check(lib.cvx_conv2d_nhwc(L.ptr(x), B, H, W, Ci, L.ptr(w), Co, 3, 1, 1, 1, 0x2000, None, None, L.ptr(out), L.stream_ptr(dev)), "plain")
lib.cvx_debug_clock_buffer(None); torch.cuda.synchronize()
d = dump.cpu().numpy().astype(np.int64) & 0xffffffff
P = Ci // 8; pow2 = (P & (P - 1)) == 0
sh = 2 if P == 4 else (1 if P == 8 else 0); swm = 15 if P >= 16 else P - 1
Wp = W + 2; SPT = (Ci + 31) // 32; rowpitch = Wp * P * 16
npix = min(H, TR) * W
bad = 0
for n in range(1, 9 * SPT):
    t, s_ = divmod(n, SPT); dh = t // 3 - 1; dwi = t % 3
    for i in range(MT):
        for tid in range(256):
            wave, lane = divmod(tid, 64); fr, fq = lane & 15, lane >> 4
            m = (wave * MT + i) * 16 + fr
            mm = m if m < npix else 0
            pr, pc = divmod(mm, W); col = pc + dwi
            sw = ((col >> sh) & swm) if pow2 else 0
            ph = ((fq ^ sw) & (P - 1)) if pow2 else fq
            S = (pr + 1) * rowpitch + ((col * P + ph) << 4)
            xk = (s_ << 6) if pow2 else 0
            ak = dh * rowpitch
            if not pow2: ak += (min(4 * s_ + fq, P - 1) - fq) << 4
            want = ((S ^ xk) + ak) & 0xffffffff
            got = int(d[4096 + (n * MT + i) * 256 + tid])
            if got != want:
                bad += 1
                if bad <= 16: print("step", n, "group", i, "tid", tid, "got", got, "want", want)
print("wrong addresses:", bad)
