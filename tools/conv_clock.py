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


def main():
    lib = L.load()
    dev = torch.device("cuda", 0)
    clk = torch.zeros(8 * 200000, dtype=torch.int64, device=dev)
    print(f"{'shape':34s} {'blocks':>6s} {'evt us':>7s} {'span':>6s} | start p50/p90/max | block mean: {'issue':>5s} {'land':>5s} {'kloop':>6s} {'epi':>5s} {'total':>6s} max")
    sel = os.environ.get("CONV_CLOCK_SHAPES")
    shapes = [SHAPES[int(i)] for i in sel.split(",")] if sel else SHAPES
    for (B, H, W, Ci, Co, k, s) in shapes:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, H, W, Ci, generator=g).half().to(dev)
        w = (torch.randn(Co, k, k, Ci, generator=g) * 0.05).half().to(dev)
        Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
        out = torch.empty(B, Ho, Wo, Co, dtype=torch.float16, device=dev)

        mode = int(os.environ.get("CONV_CLOCK_MODE", "0"))           # 3: training epilogue (raw fp32 + statistics)
        out32 = torch.empty(B, Ho, Wo, Co, dtype=torch.float32, device=dev) if mode == 3 else None
        slab = torch.zeros(16 * Co * 4, dtype=torch.int64, device=dev)

        def run():
            L.check(lib.cvx_conv2d_nhwc(L.ptr(x), B, H, W, Ci, L.ptr(w), Co, k, s, k // 2, 1, mode, None, L.ptr(slab) if mode == 3 else None,
                                        L.ptr(out32 if mode == 3 else out), L.stream_ptr(dev)), "conv")
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        evt = e0.elapsed_time(e1) / 20 * 1e3
        clk.zero_()
        lib.cvx_debug_clock_buffer(L.ptr(clk))
        run()
        torch.cuda.synchronize()
        lib.cvx_debug_clock_buffer(None)
        c = clk.cpu().numpy().reshape(-1, 8)
        c = c[c[:, 0] != 0]
        mhz = float(np.median((c[:, 6] - c[:, 5]) / np.maximum(c[:, 4] - c[:, 0], 1)) * 100.0)
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
              f"{ph[:, 0].mean():5.2f} {ph[:, 1].mean():5.2f} {ph[:, 2].mean():6.2f} {ph[:, 3].mean():5.2f} {tot.mean():6.2f} {tot.max():5.1f}  clk {mhz:5.0f} MHz",
              flush=True)


if __name__ == "__main__":
    main()
