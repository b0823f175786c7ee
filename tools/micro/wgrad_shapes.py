"""Stand-alone durations of the weight-gradient kernels on an otherwise idle GPU (the per-op table of tools/op_profile.py times them
on the lowest-priority stream while the main chain runs).  Run under the profiler and read the kernel trace:

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wg -- python tools/micro/wgrad_shapes.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from computervision.pytorch_amd import _lib as L  # noqa: E402

SHAPES = [(32, 38, 38, 512, 512, 3), (32, 19, 19, 512, 512, 3), (16, 33, 33, 256, 256, 3), (32, 75, 75, 256, 256, 3),
          (32, 150, 150, 128, 128, 3), (32, 300, 300, 64, 64, 3), (16, 33, 33, 1024, 256, 1), (16, 33, 33, 256, 1024, 1),
          (32, 80, 80, 128, 128, 3), (32, 40, 40, 256, 256, 3)]


def main():
    lib = L.load()
    dev = torch.device("cuda", 0)
    st = L.stream_ptr(dev)
    for B, H, W, Ci, Co, k in SHAPES:
        x = torch.randn(B, H, W, Ci, device=dev).half()
        dy = torch.randn(B, H, W, Co, device=dev).half()
        need = lib.cvx_conv2d_wgrad_workspace_bytes(B, H, W, Ci, Co, k)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        dw = torch.empty(Co, k, k, Ci, dtype=torch.float32, device=dev)
        for _ in range(3):
            L.check(lib.cvx_conv2d_wgrad_nhwc(L.ptr(x), L.ptr(dy), B, H, W, Ci, Co, k, 1, k // 2, 1, L.ptr(dw), L.ptr(ws), need, st), "wgrad")
        torch.cuda.synchronize()
        print(f"shape B{B} {H}x{W} {Ci}->{Co} k{k}: {2.0 * B * H * W * Ci * Co * k * k / 1e9:.1f} GFLOP", flush=True)


if __name__ == "__main__":
    main()
