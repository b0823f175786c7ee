"""Where does the row-band kernel (conv_tile.hip, forced with mode | 0x2000) differ from torch's CPU convolution?
   python tools/tile_debug.py B H W Cin Cout        error pattern on random data + tap signature
Tap signature: x = 1 on channel `ci` only, w[co][tap][ci] = 2^tap: the output is the bit set of the taps that were summed for a pixel."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from computervision.pytorch_amd import _lib as L  # noqa: E402


def run(lib, dev, x16, w16, mode=0x2000):
    B, Ci, H, W = x16.shape
    Co = w16.shape[0]
    xd, wd = x16.permute(0, 2, 3, 1).contiguous().to(dev), w16.permute(0, 2, 3, 1).contiguous().to(dev)
    out = torch.empty(B, H, W, Co, dtype=torch.float16, device=dev)
    L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, 3, 1, 1, 1, mode, None, None, L.ptr(out), L.stream_ptr(dev)), "plain")
    return out.float().permute(0, 3, 1, 2).cpu()


def main():
    B, H, W, Ci, Co = [int(v) for v in sys.argv[1:6]]
    lib = L.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(B + H * 3 + W + Ci + Co)
    x16 = torch.randn(B, Ci, H, W, generator=g).half()
    w16 = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).half()
    conv = F.conv2d(x16.float(), w16.float(), None, 1, 1)
    for rep in range(2):
        got = run(lib, dev, x16, w16)
        err = (got - conv).abs()
        bad = err > 2e-2 * conv.abs().max()
        print(f"rep {rep}: bad elements", int(bad.sum()), "of", bad.numel(), "max err", float(err.max()))
    if bad.any():
        idx = bad.nonzero().numpy()
        for name, col in (("image", 0), ("row", 2), ("col", 3)):
            vals, cnt = np.unique(idx[:, col], return_counts=True)
            print(name, dict(zip(vals.tolist(), cnt.tolist())))
    # tap signature, one input channel at a time (a few of them)
    for ci in sorted({0, 7, 8, Ci // 2 + 3, Ci - 1}):
        xs = torch.zeros(B, Ci, H, W, dtype=torch.float16)
        xs[:, ci] = 1
        ws = torch.zeros(Co, Ci, 3, 3, dtype=torch.float16)
        for t in range(9):
            ws[:, ci, t // 3, t % 3] = float(2 ** t)
        want = F.conv2d(xs.float(), ws.float(), None, 1, 1)
        ctl = run(lib, dev, xs, ws, mode=0x4000)
        got = run(lib, dev, xs, ws)
        print(f"   control (other kernels) bad {int((ctl != want).sum())}, nan in result {int(torch.isnan(got).sum())}")
        got = torch.nan_to_num(got, nan=-1.0)
        bad = got != want
        print(f"signature ci={ci}: bad {int(bad.sum())}")
        if bad.any():
            idx = bad.nonzero().numpy()
            seen = set()
            for b, c, y, x in idx:
                key = (int(y), int(x), int(got[b, c, y, x]), int(want[b, c, y, x]))
                if key in seen:
                    continue
                seen.add(key)
                if len(seen) <= 24:
                    print("   (y, x) =", key[:2], "got taps", format(key[2] & 0x1ff, "09b"), "(value", key[2], ") want", format(key[3], "09b"))


if __name__ == "__main__":
    main()
