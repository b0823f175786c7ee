"""Phase timeline of the GEMM-shaped conv kernel (tuning build; cvx_debug_clock_buffer): per workgroup 100 MHz stamps at entry, after the
per-lane setup, when chunk 0 has landed, after the K loop, after the epilogue.   CVX_LIB=build/libcvx_tuning.so python tools/gemm_debug.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from computervision.pytorch_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda", 0)
SHAPES = [(32, 38, 38, 512, 512, 3), (32, 75, 75, 256, 256, 3), (32, 19, 19, 512, 512, 3), (32, 40, 40, 256, 256, 3), (16, 33, 33, 1024, 256, 1)]
if os.environ.get("GEMM_DEBUG_FIRST"):
    SHAPES = SHAPES[:1]
for (B, H, W, Ci, Co, k) in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, Ci, generator=g).half().to(dev)
    w = (torch.randn(Co, k, k, Ci, generator=g) * (k * k * Ci) ** -0.5).half().to(dev)
    out = torch.zeros(B, H, W, Co, dtype=torch.float16, device=dev)
    sc, sh = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)

    MODE = int(os.environ.get("GEMM_DEBUG_MODE", "1"))  # 1: folded BN + SiLU -> fp16; 3: the training epilogue (raw fp32 + statistics)
    out32 = torch.zeros(B, H, W, Co, dtype=torch.float32, device=dev) if MODE == 3 else None
    slab = torch.zeros(16 * Co * 4, dtype=torch.int64, device=dev)

    def run():
        if MODE == 3:
            L.check(lib.cvx_conv2d_nhwc(L.ptr(x), B, H, W, Ci, L.ptr(w), Co, k, 1, k // 2, 1, 0x103, None, L.ptr(slab), L.ptr(out32), L.stream_ptr(dev)), "conv")
        else:
            L.check(lib.cvx_conv2d_nhwc(L.ptr(x), B, H, W, Ci, L.ptr(w), Co, k, 1, k // 2, 1, 0x101, L.ptr(sc), L.ptr(sh), L.ptr(out), L.stream_ptr(dev)), "conv")
    for _ in range(3):
        run()
    dbg = torch.zeros(8 * 4096, dtype=torch.int64, device=dev)
    lib.cvx_debug_clock_buffer(L.ptr(dbg))
    run()
    torch.cuda.synchronize()
    lib.cvx_debug_clock_buffer(None)
    t = dbg.cpu().view(-1, 8)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    rel = (t[:, :5] - t0).double() / 100.0  # us
    d = rel[:, 1:] - rel[:, :-1]
    cyc = (t[:, 6] - t[:, 5]).double()
    wall = (t[:, 4] - t[:, 0]).double() / 100.0
    print(f"B{B} {H}x{W} {Ci}->{Co} k{k}: {len(t)} workgroups; start spread {rel[:, 0].max():.1f} us; phases (mean us) setup {d[:, 0].mean():.2f} first chunk {d[:, 1].mean():.2f} "
          f"K loop {d[:, 2].mean():.2f} (min {d[:, 2].min():.2f} max {d[:, 2].max():.2f}) epilogue {d[:, 3].mean():.2f} (max {d[:, 3].max():.2f}); last end {rel[:, 4].max():.1f} us; "
          f"clock {float((cyc / wall).mean()):.0f} MHz", flush=True)
