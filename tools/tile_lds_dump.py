"""Tuning build only (CVX_LIB=build/libcvx_tuning.so CVX_TILE_DBG=8): dumps the LDS image of workgroup 0 of the row-band kernel after its
first barrier and compares the halo patch with the image the addressing rules predict.   python tools/tile_lds_dump.py H W Cin Cout"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from computervision.pytorch_amd import _lib as L  # noqa: E402


def main():
    H, W, Ci, Co = [int(v) for v in sys.argv[1:5]]
    B = 1
    lib = L.load()
    dev = torch.device("cuda", 0)
    plan = (ctypes.c_int32 * 8)()
    assert lib.cvx_debug_conv_tile_plan(B, H, W, Ci, Co, plan) == 0
    TR = plan[0]
    print("plan TR %d MT %d NTW %d NB %d" % tuple(plan[:4]))
    g = torch.Generator().manual_seed(1)
    x16 = torch.randn(B, Ci, H, W, generator=g).half()
    w16 = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).half()
    xd, wd = x16.permute(0, 2, 3, 1).contiguous().to(dev), w16.permute(0, 2, 3, 1).contiguous().to(dev)
    out = torch.empty(B, H, W, Co, dtype=torch.float16, device=dev)
    dump = torch.full((160 * 1024 // 2,), 777.0, dtype=torch.float16, device=dev)
    for rep in range(3):
        dump.fill_(777.0)
        lib.cvx_debug_clock_buffer(L.ptr(dump))
        L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, 3, 1, 1, 1, 0x2000, None, None, L.ptr(out), L.stream_ptr(dev)), "plain")
        lib.cvx_debug_clock_buffer(None)
        torch.cuda.synchronize()
        d = dump.cpu().numpy()
        P = Ci // 8
        pow2 = (P & (P - 1)) == 0
        sh = 2 if P == 4 else (1 if P == 8 else 0)
        swm = 15 if P >= 16 else P - 1
        Wp = W + 2
        xn = x16.permute(0, 2, 3, 1).numpy()[0]
        nbad = 0
        first = []
        for q in range((TR + 2) * Wp):
            pr, pc = divmod(q, Wp)
            y, x = pr - 1, pc - 1
            for phys in range(P):
                c = phys ^ ((pc >> sh) & swm) if pow2 else phys
                want = xn[y, x, c * 8:c * 8 + 8] if (0 <= y < H and 0 <= x < W) else np.zeros(8, np.float16)
                got = d[(q * P + phys) * 8:(q * P + phys) * 8 + 8]
                if not np.array_equal(got, want):
                    nbad += 1
                    if len(first) < 12:
                        first.append((pr, pc, phys, got[:3].tolist(), want[:3].tolist()))
        print(f"rep {rep}: patch units wrong {nbad} of {(TR + 2) * Wp * P}")
        for f in first:
            print("   (row, col, unit)", f[:3], "got", f[3], "want", f[4])
        # ring: chunks 0 and 1 (slots 0 and 1) right behind the patch pieces; channel block 0
        NTW, KSC = plan[2], plan[5]
        BN = 16 * NTW
        SPT = (Ci + 31) // 32
        PP = ((TR + 2) * Wp * P + 63) // 64
        ring = d[PP * 512:]
        wn = w16.permute(0, 2, 3, 1).reshape(Co, 9, Ci).numpy()
        for slot in (0, 1):
            wrong = []
            for ks in range(KSC):
                n = slot * KSC + ks
                tap, s_ = divmod(n, SPT)
                for r in range(BN):
                    for g_ in range(4):
                        k = s_ * 32 + g_ * 8
                        want = wn[r, tap, k:k + 8] if (r < Co and k < Ci) else np.zeros(8, np.float16)
                        off = slot * KSC * BN * 32 + ks * BN * 32 + r * 32 + ((g_ ^ ((r >> 1) & 3)) << 3)
                        got = ring[off:off + 8]
                        if not np.array_equal(got, want):
                            wrong.append((ks, r, g_, got[:2].tolist(), want[:2].tolist()))
            print(f"        ring slot {slot}: units wrong {len(wrong)} of {KSC * BN * 4}", wrong[:4])


if __name__ == "__main__":
    main()
