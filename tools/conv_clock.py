"""In-kernel phase timing of the convolution kernels (cvx_debug_clock_buffer): where does a workgroup spend its time,
and how are the workgroups of one launch spread over the launch's lifetime?

    python tools/conv_clock.py            # the YOLOv8-n layer shapes that dominate the step (batch 32)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from computervision.pytorch_amd import _lib as L  # noqa: E402

SHAPES = [  # (B, H, W, Cin, Cout, k, stride)
    (1, 40, 40, 64, 64, 3, 1),
    (4, 40, 40, 64, 64, 3, 1),
    (8, 40, 40, 64, 64, 3, 1),
    (16, 40, 40, 64, 64, 3, 1),
    (1, 20, 20, 512, 256, 1, 1),
    (32, 40, 40, 64, 64, 3, 1),
    (32, 20, 20, 128, 128, 3, 1),
    (32, 80, 80, 32, 32, 3, 1),
    (32, 160, 160, 16, 16, 3, 1),
    (32, 80, 80, 64, 144, 3, 1),
    (32, 20, 20, 256, 144, 3, 1),
    (32, 40, 40, 256, 128, 1, 1),
    (32, 20, 20, 512, 256, 1, 1),
    (32, 160, 160, 32, 64, 3, 2),
    (32, 80, 80, 64, 64, 1, 1),
]


# the <= 40x40 launches of YOLOv8-n (ops 15-25, 32-33, 42-49, 55-57, 60-64 of the per-op table), batch 32
R04 = [
    (32, 40, 40, 64, 64, 3, 1),     # ops 15-18, 32-33, 42-43, 56
    (32, 40, 40, 256, 128, 1, 1),   # 19
    (32, 80, 80, 128, 256, 3, 2),   # 20 (input 80x80 -> 40x40: listed as 40x40 in the table's output convention? no: op 20 reads 40x40)
    (32, 40, 40, 128, 256, 3, 2),   # 20
    (32, 20, 20, 256, 256, 1, 1),   # 21
    (32, 20, 20, 128, 128, 3, 1),   # 22-23, 47-48
    (32, 20, 20, 384, 256, 1, 1),   # 24, 46, 49
    (32, 20, 20, 256, 128, 1, 1),   # 25
    (32, 40, 40, 192, 128, 1, 1),   # 34, 41, 44
    (32, 40, 40, 128, 128, 3, 2),   # 45
    (32, 40, 40, 128, 144, 3, 1),   # 55
    (32, 40, 40, 80, 80, 3, 1),     # 57
    (32, 20, 20, 256, 144, 3, 1),   # 60
    (32, 20, 20, 64, 64, 3, 1),     # 61
    (32, 20, 20, 80, 80, 3, 1),     # 62
    (32, 20, 20, 64, 64, 1, 1),     # 63
    (32, 20, 20, 80, 80, 1, 1),     # 64
    (32, 80, 80, 32, 32, 3, 1),     # 8-11, 37-38 (for comparison)
    (32, 80, 80, 64, 64, 3, 1),     # 51
    (32, 80, 80, 64, 144, 3, 1),    # 50
    (32, 80, 80, 80, 80, 3, 1),     # 52
]


def main():
    lib = L.load()
    dev = torch.device("cuda", 0)
    clk = torch.zeros(8 * 200000, dtype=torch.int64, device=dev)
    print(f"{'shape':34s} {'blocks':>6s} {'evt us':>7s} {'span':>6s} | start p50/p90/max | block mean: {'issue':>5s} {'land':>5s} {'kloop':>6s} {'epi':>5s} {'total':>6s} max")
    sel = os.environ.get("CONV_CLOCK_SHAPES")
    shapes = [SHAPES[int(i)] for i in sel.split(",")] if sel else SHAPES
    if os.environ.get("CONV_CLOCK_SET") == "r04":
        shapes = [s for i, s in enumerate(R04) if i != 2]
    for (B, H, W, Ci, Co, k, s) in shapes:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, H, W, Ci, generator=g).half().to(dev)
        w = (torch.randn(Co, k, k, Ci, generator=g) * 0.05).half().to(dev)
        Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
        out = torch.empty(B, Ho, Wo, Co, dtype=torch.float16, device=dev)

        mode = int(os.environ.get("CONV_CLOCK_MODE", "0"))           # 3: training epilogue (raw fp32 + statistics)
        force = int(os.environ.get("CONV_CLOCK_FORCE", "0"), 0)      # 0x2000: row-band kernel (3x3 s1 only), 0x4000: never the row-band kernel
        if force == 0x2000 and not (k == 3 and s == 1 and Ci >= 32):
            continue
        out32 = torch.empty(B, Ho, Wo, Co, dtype=torch.float32, device=dev) if mode == 3 else None
        slab = torch.zeros(16 * Co * 4, dtype=torch.int64, device=dev)

        sc = torch.ones(Co, dtype=torch.float32, device=dev)
        sh = torch.zeros(Co, dtype=torch.float32, device=dev)

        def run():
            L.check(lib.cvx_conv2d_nhwc(L.ptr(x), B, H, W, Ci, L.ptr(w), Co, k, s, k // 2, 1, mode | force, L.ptr(sc) if mode == 1 else None,
                                        L.ptr(slab) if mode == 3 else (L.ptr(sh) if mode == 1 else None),
                                        L.ptr(out32 if mode == 3 else out), L.stream_ptr(dev)), "conv")
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        evt = 0.0  # (cvx_conv2d_nhwc synchronises the stream: a host-side figure would time that, not the kernel)
        clk.zero_()
        lib.cvx_debug_clock_buffer(L.ptr(clk))
        run()
        torch.cuda.synchronize()
        lib.cvx_debug_clock_buffer(None)
        c = clk.cpu().numpy().reshape(-1, 8)
        c = c[c[:, 0] != 0]
        mhz = float(np.median((c[:, 6] - c[:, 5]) / np.maximum(c[:, 4] - c[:, 0], 1)) * 100.0)
        pre = float(np.mean((c[:, 7] - c[:, 2]) / 100.0)) if c[:, 7].any() else -1.0   # (row-band kernel: first barrier -> K loop entry)
        c = c[:, :5].astype(np.float64)
        if c.shape[0] == 0:
            print(f"{B}x{H}x{W} {Ci}->{Co} k{k}s{s}: no stamps (kernel without clock marks)")
            continue
        t0 = c[:, 0].min()
        c = (c - t0) / 100.0                      # microseconds
        start = c[:, 0]
        ph = np.diff(c, axis=1)
        tot = c[:, 4] - c[:, 0]
        print(f"{B}x{H}x{W} {Ci}->{Co} k{k}s{s}".ljust(34) + f" {len(c):6d} {evt:7.1f} {c[:, 4].max():6.1f} | "
              f"{np.percentile(start, 50):5.1f} {np.percentile(start, 90):5.1f} {start.max():5.1f} | "
              f"{ph[:, 0].mean():5.2f} {ph[:, 1].mean():5.2f} {ph[:, 2].mean():6.2f} {ph[:, 3].mean():5.2f} {tot.mean():6.2f} {tot.max():5.1f}  clk {mhz:5.0f} MHz  pre-loop {pre:.2f}",
              flush=True)


if __name__ == "__main__":
    main()
