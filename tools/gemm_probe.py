"""Device time of single convolutions through the dispatcher (cvx_conv2d_nhwc) on the big-channel shapes of SSD300 / DeepLabv3+ / YOLOv8-s:
CUDA events around 10 back-to-back calls.  A/B against the LDS-DMA ring kernel: CVX_LIB=build/libcvx_tuning.so CVX_NO_GEMM=1.
    python tools/gemm_probe.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from computervision.pytorch_amd import _lib as L  # noqa: E402

SHAPES = [  # (B, H, W, Cin, Cout, k, stride, what)
    (32, 38, 38, 512, 512, 3, 1, "SSD conv4_x"), (32, 75, 75, 256, 256, 3, 1, "SSD conv3_x"), (32, 19, 19, 512, 512, 3, 1, "SSD conv5_x"),
    (32, 38, 38, 256, 512, 3, 1, "SSD conv4_1"), (32, 19, 19, 1024, 1024, 1, 1, "SSD conv7"),
    (16, 33, 33, 1024, 256, 1, 1, "R101 layer3 1x1 a"), (16, 33, 33, 256, 256, 3, 1, "R101 layer3 3x3"), (16, 33, 33, 256, 1024, 1, 1, "R101 layer3 1x1 c"),
    (16, 33, 33, 2048, 256, 1, 1, "ASPP 1x1"), (16, 65, 65, 128, 128, 3, 1, "R101 layer2 3x3"), (16, 65, 65, 512, 128, 1, 1, "R101 layer2 1x1 a"),
    (32, 40, 40, 256, 256, 3, 1, "YOLOv8-s head 40x40"), (32, 80, 80, 128, 128, 3, 1, "YOLOv8-s 80x80 c128"), (32, 20, 20, 256, 256, 3, 1, "YOLOv8-s 20x20 c256"),
]


MODE = 0x101 if os.environ.get("GEMM_PROBE_FORCE") else 1  # GEMM_PROBE_FORCE=1: the GEMM-shaped kernel on every shape


def main():
    lib = L.load()
    dev = torch.device("cuda", 0)
    for (B, H, W, Ci, Co, k, s, what) in SHAPES:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, H, W, Ci, generator=g).half().to(dev)
        w = (torch.randn(Co, k, k, Ci, generator=g) * (k * k * Ci) ** -0.5).half().to(dev)
        Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
        out = torch.empty(B, Ho, Wo, Co, dtype=torch.float16, device=dev)
        sc, sh = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)

        def run():
            L.check(lib.cvx_conv2d_nhwc(L.ptr(x), B, H, W, Ci, L.ptr(w), Co, k, s, k // 2, 1, MODE, L.ptr(sc), L.ptr(sh), L.ptr(out), L.stream_ptr(dev)), "conv")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        gf = 2.0 * B * Ho * Wo * Co * Ci * k * k / 1e9
        print(f"{what:24s} B{B} {H}x{W} {Ci}->{Co} k{k}s{s}: {us:8.1f} us  {gf / us * 1e3:7.1f} TF/s ({gf / us * 1e3 / 2516.6 * 100:4.1f} % of the MFMA roof)", flush=True)


if __name__ == "__main__":
    main()
