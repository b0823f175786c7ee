"""In-kernel phase stamps of the fat 3x3 weight-gradient kernel (conv_wgrad_k3.hip) on YOLOv8-n's shapes, batch 32 (tuning library):
workgroup start | prologue DMA issued | first chunk landed | K loop done | end, 100 MHz wall clock of thread 0 of every workgroup.

    CVX_LIB=build/libcvx_tuning.so python tools/micro/wgrad_k3_clock.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from computervision.pytorch_amd import _lib as L  # noqa: E402

SHAPES = [(160, 16, 16, 48), (80, 32, 32, 128), (80, 64, 144, 64), (80, 64, 64, 144), (80, 80, 80, 144), (40, 64, 64, 256), (40, 128, 144, 128),
          (40, 80, 80, 144), (20, 128, 128, 384), (20, 256, 144, 256), (20, 64, 64, 144), (20, 80, 80, 144)]


def main():
    lib = L.load()
    dev = torch.device("cuda", 0)
    st = L.stream_ptr(dev)
    B = 32
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    clk = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    print(f"{'layer':24s} {'wgs':>4s} {'us':>7s} | block means (us): issue  land  kloop   epi | span first-start..last-end, mean block life")
    for H, cin, cout, x_ld in SHAPES:
        x = torch.randn(B, H, H, x_ld, device=dev).half()
        dy = torch.randn(B, H, H, cout, device=dev).half()
        us, ns = C.c_float(0), C.c_int32(0)
        L.check(lib.cvx_debug_clock_buffer(None), "clk off")
        L.check(lib.cvx_wgrad_time_unit(L.ptr(x), L.ptr(dy), B, H, H, cin, cout, 3, x_ld, cout, 0, 20, L.ptr(ws), ws.numel(), C.byref(us), C.byref(ns), st), "time")
        clk.zero_()
        L.check(lib.cvx_debug_clock_buffer(L.ptr(clk)), "clk on")
        u2 = C.c_float(0)
        L.check(lib.cvx_wgrad_time_unit(L.ptr(x), L.ptr(dy), B, H, H, cin, cout, 3, x_ld, cout, ns.value, 1, L.ptr(ws), ws.numel(), C.byref(u2), None, st), "time")
        torch.cuda.synchronize()
        L.check(lib.cvx_debug_clock_buffer(None), "clk off")
        c = clk.cpu().numpy().reshape(-1, 8)
        c = c[(c[:, 0] > 0) & (c[:, 4] > 0)].astype(np.float64) / 100.0     # us
        d = np.diff(c[:, :5], axis=1).mean(0)
        span = c[:, 4].max() - c[:, 0].min()
        print(f"{H:3d}x{H:<3d} {cin:3d}->{cout:<3d} ld{x_ld:<4d} {len(c):4d} {us.value:7.1f} |                  {d[0]:6.2f} {d[1]:5.2f} {d[2]:6.2f} {d[3]:5.2f} | {span:6.1f}  {(c[:, 4] - c[:, 0]).mean():6.1f}", flush=True)


if __name__ == "__main__":
    main()
