"""Stand-alone device time of the weight-gradient kernels on YOLOv8-n's stride-1 layers (batch 32), alone on the GPU: the streaming
kernel (conv_wgrad_stream.hip, round 5) against the two older kernels (CVX_NO_WGRAD_STREAM=1 in the environment selects those).
TUNING library only (cvx_wgrad_time_unit):

    CVX_LIB=build/libcvx_tuning.so python tools/micro/wgrad_stream_probe.py            # streaming kernel, its planner's splits
    CVX_LIB=build/libcvx_tuning.so CVX_NO_WGRAD_STREAM=1 python tools/micro/wgrad_stream_probe.py   # the older kernels, the engine's round-4 splits
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from computervision.pytorch_amd import _lib as L  # noqa: E402

# (H, Cin, Cout, k, x_ld, count in YOLOv8-n)   -- stride-1 convs of the train step, batch 32
LAYERS = [(160, 16, 16, 3, 48, 2), (160, 32, 32, 1, 32, 1), (160, 48, 32, 1, 48, 1),
          (80, 32, 32, 3, 128, 6), (80, 64, 64, 1, 64, 1), (80, 128, 64, 1, 128, 1), (80, 192, 64, 1, 192, 1), (80, 96, 64, 1, 96, 1),
          (80, 64, 144, 3, 64, 1), (80, 64, 64, 3, 144, 1), (80, 80, 80, 3, 144, 1), (80, 64, 64, 1, 64, 1), (80, 80, 80, 1, 80, 1),
          (40, 64, 64, 3, 256, 9), (40, 128, 128, 1, 128, 1), (40, 256, 128, 1, 256, 1), (40, 384, 128, 1, 384, 1), (40, 192, 128, 1, 192, 3),
          (40, 128, 144, 3, 128, 1), (40, 80, 80, 3, 144, 1), (40, 64, 64, 1, 64, 1), (40, 80, 80, 1, 80, 1),
          (20, 128, 128, 3, 384, 4), (20, 256, 256, 1, 256, 1), (20, 384, 256, 1, 384, 3), (20, 256, 128, 1, 256, 1), (20, 512, 256, 1, 512, 1),
          (20, 256, 144, 3, 256, 1), (20, 64, 64, 3, 144, 1), (20, 80, 80, 3, 144, 1), (20, 64, 64, 1, 64, 1), (20, 80, 80, 1, 80, 1)]


def old_splits(B, H, cin, cout, k):
    """the engine's round-4 pixel splits of the older kernels (engine.hip: plan), for the comparison run"""
    M = B * H * H
    jtot = k * k * ((cin + 15) // 16 * 16)
    if k == 3 and cin >= 16 and cin * cout <= 16383:
        def pt(c):
            return 1 if c <= 16 else 2 if c <= 32 else 4 if c <= 64 else 5 if c <= 80 else (4 if c % 64 == 0 else 5)
        gx = -(-cout // (16 * pt(cout)))
        gy = -(-cin // (16 * pt(cin))) * (3 if pt(cout) * pt(cin) >= 8 else 1)
        tiles = -(-H // 16) * -(-H // 8) * B
        ns = max(1, 128 // (gx * gy))
        return min(ns, tiles, max(1, (16 << 20) // (cout * jtot * 4)))
    if cout <= 16:
        cb, jb = 16, 192
    elif cout <= 32:
        cb, jb = 32, 128
    elif cout % 64 != 0 and (cout % 48 == 0 or cout <= 96):
        cb, jb = 48, 128
    else:
        cb, jb = 64, 64
    tiles = -(-cout // cb) * -(-jtot // jb)
    ns = min(max(1, M // 256), max(1, 2048 // tiles))
    return min(ns, max(1, (8 << 20) // (cout * jtot * 4)), 512)


def main():
    lib = L.load()
    assert hasattr(lib, "cvx_wgrad_time_unit"), "needs the tuning library (CVX_LIB=build/libcvx_tuning.so)"
    dev = torch.device("cuda", 0)
    st = L.stream_ptr(dev)
    B = 32
    old = bool(os.environ.get("CVX_NO_WGRAD_STREAM"))
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    tot = 0.0
    print(f"{'layer':28s} {'splits':>6s} {'us':>8s} {'GB/s':>7s} {'HBM-floor us':>12s}   x count")
    for H, cin, cout, k, x_ld, cnt in LAYERS:
        x = torch.randn(B, H, H, x_ld, device=dev).half()
        dy = torch.randn(B, H, H, cout, device=dev).half()
        us, ns = C.c_float(0), C.c_int32(0)
        nsplit = old_splits(B, H, cin, cout, k) if old else 0
        L.check(lib.cvx_wgrad_time_unit(L.ptr(x), L.ptr(dy), B, H, H, cin, cout, k, x_ld, cout, nsplit, 20, L.ptr(ws), ws.numel(), C.byref(us), C.byref(ns), st), "time")
        by = 2.0 * B * H * H * (cin + cout)
        print(f"{H:3d}x{H:<3d} {cin:3d}->{cout:<3d} k{k} ld{x_ld:<4d}     {ns.value:6d} {us.value:8.1f} {by / us.value / 1e3:7.0f} {by / 4.5e6:12.1f}   x {cnt}", flush=True)
        tot += us.value * cnt
    print(f"sum over the step's stride-1 layers: {tot:.0f} us")


if __name__ == "__main__":
    main()
