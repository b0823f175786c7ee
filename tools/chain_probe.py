"""Probe of the tile-resident chain kernel (csrc/conv_chain.hip): correctness against torch fp32 on the same fp16 operands and
time against the per-layer kernels (cvx_conv2d_nhwc) on the YOLOv8-n shapes.

    python tools/chain_probe.py [pair|detect|conv|all]
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from computervision.pytorch_amd import _lib as L  # noqa: E402

dev = torch.device("cuda", 0)
lib = L.load()


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def dev_time(call, reps=20):
    """mean device time (us) of the chain launch behind `call(reps, elapsed_ptr)` -- the plan is built once, HIP events around the launches"""
    import ctypes
    us = ctypes.c_float(0)
    call(reps, ctypes.byref(us))
    return us.value


def clock_report(call, label):
    """per-phase wall time of the workgroups of one launch (cvx_debug_clock_buffer stamps: 32 slots per workgroup, 100 MHz)"""
    import numpy as np
    clk = torch.zeros(32 * 4096, dtype=torch.int64, device=dev)
    lib.cvx_debug_clock_buffer(L.ptr(clk))
    call(1, None)
    torch.cuda.synchronize()
    lib.cvx_debug_clock_buffer(None)
    c = clk.cpu().numpy().reshape(-1, 32).astype(np.float64)
    c = c[c[:, 0] != 0]
    if not len(c):
        return
    t0 = c[:, 0].min()
    rel = (c - t0) / 100.0
    slots = [s for s in range(32) if (c[:, s] != 0).all()]
    mhz = float(np.median((c[:, 29] - c[:, 28]) / np.maximum(c[:, 31] - c[:, 0], 1)) * 100.0)
    slots = [s for s in slots if s < 28 or s > 29]
    print(f"   clocks {label}: {len(c)} workgroups, starts {np.percentile(rel[:, 0], 50):.1f}/{rel[:, 0].max():.1f} us, ends max {rel[:, 31].max():.1f} us; "
          f"shader clock {mhz:.0f} MHz; mean stamps since own start: " + " ".join(f"[{s}]{(rel[:, s] - rel[:, 0]).mean():.2f}" for s in slots), flush=True)


def mk(shape, g, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).half()


def conv_ref(x_nhwc, w_okkc, k, stride=1):
    """fp32 conv of fp16-valued operands; x (B,H,W,C), w (O,k,k,C) -> (B,H,W,O)"""
    x = x_nhwc.float().permute(0, 3, 1, 2)
    w = w_okkc.float().permute(0, 3, 1, 2)
    return F.conv2d(x, w, stride=stride, padding=k // 2).permute(0, 2, 3, 1)


def affine_silu(y, sc, sh):
    return F.silu(y * sc + sh)


def old_conv(x, w, cout, k, stride, sc, sh, out):
    B, H, W, Ci = x.shape
    L.check(lib.cvx_conv2d_nhwc(L.ptr(x), B, H, W, Ci, L.ptr(w), cout, k, stride, k // 2, 1, 1, L.ptr(sc), L.ptr(sh), L.ptr(out), L.stream_ptr(dev)), "conv")


def probe_pair(B, H, W, C, th, tw, shortcut=True):
    g = torch.Generator().manual_seed(0)
    x = mk((B, H, W, C), g)
    w1, w2 = mk((C, 3, 3, C), g, (9 * C) ** -0.5), mk((C, 3, 3, C), g, (9 * C) ** -0.5)
    sc1, sh1 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    sc2, sh2 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    nref = min(B, 2)
    m = affine_silu(conv_ref(x[:nref], w1, 3), sc1, sh1).half()
    ref = affine_silu(conv_ref(m, w2, 3), sc2, sh2) + (x[:nref].float() if shortcut else 0)
    xd, w1d, w2d = x.to(dev), w1.to(dev), w2.to(dev)
    sc1d, sh1d, sc2d, sh2d = sc1.to(dev), sh1.to(dev), sc2.to(dev), sh2.to(dev)
    out = torch.zeros(B, H, W, C, dtype=torch.float16, device=dev)

    def run(reps=1, el=None):
        L.check(lib.cvx_chain_pair_unit(L.ptr(xd), B, H, W, C, L.ptr(w1d), L.ptr(sc1d), L.ptr(sh1d), L.ptr(w2d), L.ptr(sc2d), L.ptr(sh2d), int(shortcut),
                                        L.ptr(out), th, tw, reps, el, L.stream_ptr(dev)), "pair")
    run()
    torch.cuda.synchronize()
    err = (out[:nref].float().cpu() - ref).abs().max().item() / ref.abs().max().item()
    tmp = torch.empty_like(out)
    out2 = torch.empty_like(out)

    def run_old():
        old_conv(xd, w1d, C, 3, 1, sc1d, sh1d, tmp)
        old_conv(tmp, w2d, C, 3, 1, sc2d, sh2d, out2)
    t_new, t_old = dev_time(run), timeit(run_old)
    if os.environ.get("CHAIN_CLOCKS"):
        clock_report(run, "pair")
    gf = 2 * 2 * B * H * W * C * C * 9 / 1e9
    print(f"pair   B{B} {H}x{W} c{C} tile {th}x{tw}: rel err {err:.2e}   chain {t_new:7.1f} us ({gf / t_new * 1e3:6.1f} TF/s)   two launches {t_old:7.1f} us (no residual)", flush=True)
    return err


def probe_conv(B, H, W, Ci, Co, k, stride, th, tw):
    g = torch.Generator().manual_seed(1)
    x = mk((B, H, W, Ci), g)
    w = mk((Co, k, k, Ci), g, (k * k * Ci) ** -0.5)
    sc, sh = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.1
    nref = min(B, 2)
    ref = affine_silu(conv_ref(x[:nref], w, k, stride), sc, sh)
    xd, wd, scd, shd = x.to(dev), w.to(dev), sc.to(dev), sh.to(dev)
    out = torch.zeros(B, H // stride, W // stride, Co, dtype=torch.float16, device=dev)
    out2 = torch.empty_like(out)

    def run(reps=1, el=None):
        L.check(lib.cvx_chain_conv_unit(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, k, stride, 0, L.ptr(scd), L.ptr(shd), 0, L.ptr(out), th, tw,
                                        reps, el, L.stream_ptr(dev)), "conv")
    run()
    torch.cuda.synchronize()
    err = (out[:nref].float().cpu() - ref).abs().max().item() / ref.abs().max().item()
    t_new, t_old = dev_time(run), timeit(lambda: old_conv(xd, wd, Co, k, stride, scd, shd, out2))
    gf = 2 * B * (H // stride) * (W // stride) * Co * Ci * k * k / 1e9
    print(f"conv   B{B} {H}x{W} {Ci}->{Co} k{k}s{stride} tile {th}x{tw}: rel err {err:.2e}   chain {t_new:7.1f} us ({gf / t_new * 1e3:6.1f} TF/s)   old {t_old:7.1f} us", flush=True)
    return err


def probe_detect(B, H, W, Cin, th, tw, cb=64, cc=80, ncp=80):
    g = torch.Generator().manual_seed(2)
    x = mk((B, H, W, Cin), g)
    wa = mk((cb + cc, 3, 3, Cin), g, (9 * Cin) ** -0.5)
    wb1, wb2 = mk((cb, 3, 3, cb), g, (9 * cb) ** -0.5), mk((cc, 3, 3, cc), g, (9 * cc) ** -0.5)
    wo1, wo2 = mk((64, 1, 1, cb), g, cb ** -0.5), mk((ncp, 1, 1, cc), g, cc ** -0.5)
    sca, sha = torch.rand(cb + cc, generator=g) + 0.5, torch.randn(cb + cc, generator=g) * 0.1
    scb, shb = torch.rand(cb + cc, generator=g) + 0.5, torch.randn(cb + cc, generator=g) * 0.1
    bias = torch.randn(64 + ncp, generator=g)
    nref = min(B, 2)
    a = affine_silu(conv_ref(x[:nref], wa, 3), sca, sha).half()
    hb = affine_silu(conv_ref(a[..., :cb], wb1, 3), scb[:cb], shb[:cb]).half()
    hc = affine_silu(conv_ref(a[..., cb:], wb2, 3), scb[cb:], shb[cb:]).half()
    ref = torch.cat([conv_ref(hb, wo1, 1) + bias[:64], conv_ref(hc, wo2, 1) + bias[64:]], -1).reshape(nref, H * W, 64 + ncp)
    d = lambda t: t.to(dev)  # noqa: E731
    xd, wad, wb1d, wb2d, wo1d, wo2d, scad, shad, scbd, shbd, biasd = map(d, (x, wa, wb1, wb2, wo1, wo2, sca, sha, scb, shb, bias))
    A = H * W + 64
    pred = torch.zeros(B, A, 64 + ncp, device=dev)

    def run(reps=1, el=None):
        L.check(lib.cvx_chain_detect_unit(L.ptr(xd), B, H, W, Cin, cb, cc, ncp, L.ptr(wad), L.ptr(scad), L.ptr(shad), L.ptr(wb1d), L.ptr(wb2d), L.ptr(scbd),
                                          L.ptr(shbd), L.ptr(wo1d), L.ptr(wo2d), L.ptr(biasd), L.ptr(pred), A, 32, th, tw, reps, el, L.stream_ptr(dev)), "detect")
    run()
    torch.cuda.synchronize()
    got = pred[:nref, 32:32 + H * W].cpu()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    untouched = float(pred[:, :32].abs().max() + pred[:, 32 + H * W:].abs().max())
    t_new = dev_time(run)
    if os.environ.get("CHAIN_CLOCKS"):
        clock_report(run, "detect")
    gf = 2 * B * H * W * ((cb + cc) * 9 * Cin + cb * cb * 9 + cc * cc * 9 + 64 * cb + ncp * cc) / 1e9
    print(f"detect B{B} {H}x{W} cin {Cin} tile {th}x{tw}: rel err {err:.2e} (outside rows {untouched:.1f})   chain {t_new:7.1f} us ({gf / t_new * 1e3:6.1f} TF/s)", flush=True)
    return err


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    errs = []
    if what in ("pair", "all"):
        errs.append(probe_pair(2, 40, 40, 64, 10, 20))
        errs.append(probe_pair(1, 24, 40, 32, 8, 8, shortcut=False))       # ragged: 24 = 3 tiles, 40 = 5 tiles
        errs.append(probe_pair(3, 20, 20, 128, 5, 10))
        errs.append(probe_pair(2, 30, 50, 64, 12, 16))                    # tiles overhang the image
        errs.append(probe_pair(32, 40, 40, 64, 10, 20))
        errs.append(probe_pair(32, 80, 80, 32, 16, 16))
        errs.append(probe_pair(32, 20, 20, 128, 5, 10))
        errs.append(probe_pair(32, 160, 160, 16, 8, 32))
    if what in ("detect", "all"):
        errs.append(probe_detect(2, 16, 24, 64, 8, 16))
        errs.append(probe_detect(32, 80, 80, 64, 8, 16))
        errs.append(probe_detect(32, 40, 40, 128, 8, 8))
        errs.append(probe_detect(32, 20, 20, 256, 5, 10))
    if what in ("conv", "all"):
        errs.append(probe_conv(2, 40, 40, 64, 64, 3, 1, 10, 20))
        errs.append(probe_conv(2, 40, 40, 64, 128, 3, 2, 10, 10))
        errs.append(probe_conv(2, 20, 20, 64, 32, 1, 1, 10, 20))
        errs.append(probe_conv(32, 80, 80, 64, 128, 3, 2, 8, 10))
        errs.append(probe_conv(32, 40, 40, 128, 256, 3, 2, 5, 10))
        errs.append(probe_conv(32, 160, 160, 32, 64, 3, 2, 10, 16))
    if what == "sweep":
        for th, tw in ((10, 20), (8, 20), (10, 10), (5, 20), (8, 16), (20, 10), (14, 14), (6, 20)):
            errs.append(probe_pair(32, 40, 40, 64, th, tw))
        for th, tw in ((16, 16), (10, 20), (8, 40), (8, 16), (16, 20), (14, 20)):
            errs.append(probe_pair(32, 80, 80, 32, th, tw))
        for th, tw in ((5, 10), (10, 10), (5, 20), (4, 10)):
            errs.append(probe_pair(32, 20, 20, 128, th, tw))
        for th, tw in ((8, 16), (8, 8), (4, 16), (8, 10), (10, 10), (5, 16), (4, 20)):
            errs.append(probe_detect(32, 80, 80, 64, th, tw))
        for th, tw in ((8, 8), (8, 10), (4, 20), (5, 8), (5, 10), (4, 8)):
            errs.append(probe_detect(32, 40, 40, 128, th, tw))
        for th, tw in ((5, 10), (5, 5), (4, 10), (2, 20), (4, 5)):
            errs.append(probe_detect(32, 20, 20, 256, th, tw))
    if what == "sweep2":   # single-stage 3x3 s1 on every YOLOv8-n shape: best tile per shape (old = eval profile of the per-layer kernels)
        shapes = [(160, 16, 16, 30), (80, 32, 32, 20), (40, 64, 64, 19), (20, 128, 128, 23), (80, 64, 64, 88), (80, 80, 80, 73), (40, 64, 64, 42),
                  (40, 80, 80, 82), (20, 64, 64, 14), (20, 80, 80, 16.5), (80, 64, 144, 70), (40, 128, 144, 121), (20, 256, 144, 53)]
        tiles = [(8, 16), (8, 20), (10, 20), (16, 16), (16, 20), (20, 20), (8, 32), (8, 40), (16, 32), (16, 40), (10, 40), (20, 40), (5, 20), (10, 10), (4, 20), (5, 10)]
        import io, contextlib
        for hw, ci, co, old in shapes:
            best = None
            for th, tw in tiles:
                if hw % th or hw % tw:
                    continue
                buf = io.StringIO()
                try:
                    with contextlib.redirect_stdout(buf):
                        probe_conv(32, hw, hw, ci, co, 3, 1, th, tw)
                except L.CvxError:
                    continue
                line = buf.getvalue()
                us = float(line.split("chain")[1].split("us")[0])
                if best is None or us < best[0]:
                    best = (us, th, tw)
            print(f"3x3 s1 {hw}x{hw} {ci}->{co}: best chain {best[0]:.1f} us at tile {best[1]}x{best[2]}   (per-layer kernel: {old} us)", flush=True)
        return 0
    bad = [e for e in errs if not e < 5e-3]
    print("worst rel err", max(errs), "FAILED" if bad else "ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
