"""Standalone durations of the BatchNorm backward kernels (no concurrent weight-gradient stream), for comparison with the
in-step durations of a kernel trace.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bnclk -- python tools/bn_clock.py
    python tools/bn_clock.py report <kernel_trace.csv>"""
import os
import sys

SHAPES = [(32, 160, 32), (32, 160, 16), (32, 80, 64), (32, 80, 32), (32, 80, 144), (32, 40, 128), (32, 40, 64), (32, 20, 256), (32, 20, 128), (32, 20, 64)]
REP = 10

if len(sys.argv) > 2 and sys.argv[1] == "report":
    import pandas as pd
    df = pd.read_csv(sys.argv[2]).sort_values("Start_Timestamp")
    df["dur"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
    for key in ("bn_bwd_reduce", "bn_bwd_apply", "bn_silu_apply"):
        d = df[df["Kernel_Name"].str.contains(key)]["dur"].values
        shapes = SHAPES
        if len(d) != len(shapes) * REP:
            print(key, "unexpected launch count", len(d))
            continue
        for i, (B, S, C) in enumerate(shapes):
            x = d[i * REP + 2:(i + 1) * REP]
            el = B * S * S * C
            print(f"{key:14s} {S:4d}^2 x{C:4d}  {x.mean():7.1f} us (min {x.min():6.1f})  {el * (4 if 'reduce' in key else 6 if 'bwd' in key else 8) / x.mean() / 1e3:7.0f} GB/s")
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import torch
L = importlib.import_module("computervision.pytorch_amd._lib")
lib = L.load()
dev = torch.device("cuda:0")
st = L.stream_ptr(dev)
for (B, S, C) in SHAPES:
    y = torch.randn(B, S, S, C, device=dev)
    out = torch.empty(B, S, S, C, dtype=torch.float16, device=dev)
    xh, dy = torch.empty_like(out), torch.empty_like(out)
    gout = torch.randn(B, S, S, C, device=dev).half()
    ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mean, invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    for _ in range(REP):
        L.check(lib.cvx_bn_silu_train_nhwc(L.ptr(y), B, S * S, C, L.ptr(ga), L.ptr(be), 1e-3, 0.03, L.ptr(rm), L.ptr(rv), None, L.ptr(out), L.ptr(xh),
                                           L.ptr(mean), L.ptr(invstd), st), "fwd")
        L.check(lib.cvx_bn_silu_bwd_nhwc(L.ptr(xh), L.ptr(gout), B, S * S, C, L.ptr(ga), L.ptr(be), L.ptr(invstd), 1.0, L.ptr(dg), L.ptr(db), L.ptr(dy),
                                         None, 0, st), "bwd")
    torch.cuda.synchronize()
print("done")
