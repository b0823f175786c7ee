"""On-box diagnostic sweep (not a test): runs every HIP entry point against the CPU oracle / torch-CPU fp32
and prints error metrics, so one gpurun call shows everything.  Usage: python tools/gpu_diag.py [sections...]"""
import os
import sys
import time
import traceback

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

from computervision.pytorch_amd import _lib as L  # noqa: E402
from computervision.pytorch_amd import engine as E  # noqa: E402

dev = torch.device("cuda:0")


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30)), float((a - b).abs().max())


def sect_conv():
    lib = L.load()
    cases = [  # B, H, W, Cin, Cout, k, s
        (2, 16, 16, 16, 16, 3, 1), (2, 20, 20, 8, 16, 3, 2), (1, 24, 24, 32, 64, 3, 2), (2, 10, 10, 64, 144, 3, 1),
        (2, 12, 12, 48, 32, 1, 1), (1, 20, 20, 384, 256, 1, 1), (3, 7, 9, 80, 80, 3, 1), (2, 40, 40, 128, 128, 3, 1),
        (1, 13, 13, 256, 512, 3, 2),
    ]
    for (B, H, W, Ci, Co, k, s) in cases:
        g = torch.Generator().manual_seed(B * 1000 + H + Ci + Co)
        x = torch.randn(B, Ci, H, W, generator=g)
        w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
        x16, w16 = x.half(), w.half()
        ref = F.conv2d(x16.float(), w16.float(), None, s, k // 2)
        OH, OW = ref.shape[2:]
        xd = x16.permute(0, 2, 3, 1).contiguous().to(dev)
        wd = w16.permute(0, 2, 3, 1).contiguous().to(dev)
        out = torch.empty(B, OH, OW, Co, dtype=torch.float16, device=dev)
        L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, k, s, k // 2, 1, 0, None, None, L.ptr(out), L.stream_ptr(dev)), "conv")
        torch.cuda.synchronize()
        r = rel(out.float().cpu().permute(0, 3, 1, 2), ref)
        # dgrad
        dy = torch.randn(B, Co, OH, OW, generator=g).half()
        xr = x16.float().requires_grad_(True)
        wr = w16.float().requires_grad_(True)
        F.conv2d(xr, wr, None, s, k // 2).backward(dy.float())
        dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
        wtd = w16.permute(1, 2, 3, 0).contiguous().to(dev)    # [cin][kh][kw][cout]
        dx = torch.zeros(B, H, W, Ci, dtype=torch.float16, device=dev)
        L.check(lib.cvx_conv2d_dgrad_nhwc(L.ptr(dyd), B, H, W, Ci, L.ptr(wtd), Co, k, s, k // 2, 1, L.ptr(dx), L.stream_ptr(dev)), "dgrad")
        torch.cuda.synchronize()
        r2 = rel(dx.float().cpu().permute(0, 3, 1, 2), xr.grad)
        # wgrad
        need = lib.cvx_conv2d_wgrad_workspace_bytes(B, OH, OW, Ci, Co, k)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        dw = torch.empty(Co, k, k, Ci, dtype=torch.float32, device=dev)
        L.check(lib.cvx_conv2d_wgrad_nhwc(L.ptr(xd), L.ptr(dyd), B, H, W, Ci, Co, k, s, k // 2, 1, L.ptr(dw), L.ptr(ws), need, L.stream_ptr(dev)), "wgrad")
        torch.cuda.synchronize()
        r3 = rel(dw.cpu().permute(0, 3, 1, 2), wr.grad)
        print(f"conv B{B} {H}x{W} {Ci}->{Co} k{k}s{s}: fwd rel {r[0]:.2e} max {r[1]:.2e} | dgrad rel {r2[0]:.2e} | wgrad rel {r3[0]:.2e}", flush=True)


def _model(bs_hw, seed=0):
    from computervision.pytorch_amd.model import Yolo8
    torch.manual_seed(seed)
    return Yolo8("n", 80).to(dev)


def sect_forward():
    from oracle import synth
    from oracle import yolov8_ref as O
    for (B, H) in ((2, 128), (1, 320)):
        x = synth.images(B, H, H, seed=1)
        sd = O.init_state_dict("n", 80, seed=0)
        taps = {}
        ref = O.forward(sd, x, "n", 80, training=True, taps=taps)
        m = _model(None).train()
        with torch.no_grad():
            outs = m(x.to(dev))
        torch.cuda.synchronize()
        for i, (o, r) in enumerate(zip(outs, ref)):
            e = rel(o.cpu(), r.detach())
            print(f"train fwd B{B} {H}px level {i}: rel {e[0]:.3e} max {e[1]:.3e} (ref absmax {float(r.abs().max()):.2f})", flush=True)
        # intermediate layers straight from the engine's buffers
        eng = m._last_engine
        for idx in (0, 1, 2, 4, 6, 9, 12, 15, 18, 21):
            pass
        rm = m.state_dict()["model.0.bn.running_mean"].cpu()
        print("  running_mean[0] rel", rel(rm, sd["model.0.bn.running_mean"]), "running_var rel",
              rel(m.state_dict()["model.22.cv3.2.1.bn.running_var"].cpu(), sd["model.22.cv3.2.1.bn.running_var"]), flush=True)
        # eval with the updated running stats
        m.eval()
        with torch.no_grad():
            y, _ = m(x.to(dev))
            yr, _ = O.forward(sd, x, "n", 80, training=False)
        e = rel(y.cpu()[:, :4], yr[:, :4])
        e2 = rel(y.cpu()[:, 4:], yr[:, 4:])
        print(f"eval fwd B{B} {H}px: boxes rel {e[0]:.3e} max {e[1]:.3e} | scores rel {e2[0]:.3e} max {e2[1]:.3e}", flush=True)


def sect_loss():
    from oracle import synth
    from oracle import yolov8_ref as O
    for (B, H, seed) in ((2, 128, 3), (4, 160, 4), (3, 320, 5)):
        hw = [(H // s, H // s) for s in (8, 16, 32)]
        A = sum(a * b for a, b in hw)
        g = torch.Generator().manual_seed(seed)
        pred = torch.randn(B, A, 144, generator=g)
        pred[..., 64:] = pred[..., 64:] * 2 - 3
        batch = synth.targets(B, seed=seed)
        feats = []
        off = 0
        pr = pred.clone().requires_grad_(True)
        for (h, w) in hw:
            feats.append(pr[:, off:off + h * w].permute(0, 2, 1).reshape(B, 144, h, w))
            off += h * w
        aux = {}
        loss, items = O.v8_loss(feats, batch, 80, aux=aux)
        loss.backward()
        from computervision.pytorch_amd.train import flatten_targets
        op = E.V8LossOp(80)
        its, dpred = op(pred.to(dev), flatten_targets(batch, dev), hw, (8, 16, 32), 256.0)
        torch.cuda.synchronize()
        gi = dpred.float().cpu() / 256.0
        print(f"loss B{B} {H}px: items hip {its.cpu().tolist()} ref {items.tolist()} fg {int(aux['fg_mask'].sum())} tss {aux['score_sum']:.4f}")
        print("   dpred rel", rel(gi, pr.grad), " box part", rel(gi[..., :64], pr.grad[..., :64]), " cls part", rel(gi[..., 64:], pr.grad[..., 64:]), flush=True)
    # no targets
    pred = torch.randn(2, 336, 144)
    its, dpred = E.V8LossOp(80)(pred.to(dev), torch.zeros(0, 6, device=dev), [(16, 16), (8, 8), (4, 4)], (8, 16, 32), 1.0)
    feats = [pred[:, :256].permute(0, 2, 1).reshape(2, 144, 16, 16), pred[:, 256:320].permute(0, 2, 1).reshape(2, 144, 8, 8),
             pred[:, 320:].permute(0, 2, 1).reshape(2, 144, 4, 4)]
    l, it = O.v8_loss(feats, {"batch_idx": torch.zeros(0), "cls": torch.zeros(0, 1), "bboxes": torch.zeros(0, 4)}, 80)
    print("loss (no targets): hip", its.cpu().tolist(), "ref", it.tolist(), flush=True)


def sect_train():
    from oracle import synth
    from oracle import yolov8_ref as O
    from computervision.pytorch_amd.train import FlatAdam, FusedTrainStep, V8DetectionLoss
    from configs import Yolo8DetConfig
    B, H = 4, 160
    x, batch = synth.images(B, H, H, seed=1), synth.targets(B, seed=2)
    sd = O.init_state_dict("n", 80, seed=0)
    state = {}
    m = _model(None).train()
    cfg = Yolo8DetConfig()
    crit = V8DetectionLoss(cfg, m)
    opt = FlatAdam(m, lr=1e-3)
    # step 1 with explicit pieces so gradients can be compared before Adam touches them
    eng = m.engine_for(H, H)
    pred = m._run_forward(x.to(dev), training=True)
    from computervision.pytorch_amd.train import flatten_targets
    its, dpred = crit.op(pred, flatten_targets(batch, dev), m.level_shapes(H, H), (8, 16, 32), crit.loss_scale)
    m.flat_grads.zero_()
    eng.backward(dpred, crit.loss_scale)
    torch.cuda.synchronize()
    loss_ref, items_ref, grads_ref, feats_ref = O.train_step(sd, x, batch, state)
    print("train step1 items hip", its.cpu().tolist(), "ref", items_ref.tolist())
    m.attach_grads()
    named = dict(m.named_parameters())
    worst = []
    for k, gref in grads_ref.items():
        gh = named[k].grad.cpu()
        e = rel(gh, gref)
        worst.append((e[0], k, float(gref.norm())))
    worst.sort(reverse=True)
    print("  worst 12 param-grad rel errors:")
    for e, k, n in worst[:12]:
        print(f"    {k:45s} rel {e:.3e} |g| {n:.3e}")
    print("  median rel", float(np.median([w[0] for w in worst])), flush=True)
    gall_h = torch.cat([named[k].grad.cpu().flatten() for k in grads_ref])
    gall_r = torch.cat([grads_ref[k].flatten() for k in grads_ref])
    print("  global grad rel", rel(gall_h, gall_r), flush=True)
    opt.step(zero_grad=True)
    torch.cuda.synchronize()
    after = m.state_dict()
    for k in ("model.0.conv.weight", "model.22.cv2.1.2.weight", "model.4.m.1.cv2.bn.weight", "model.22.cv3.0.2.bias"):
        print(f"  after Adam {k}: rel {rel(after[k].cpu(), sd[k])[0]:.3e}")
    # fused steps 2..3
    step = FusedTrainStep(m, crit, opt)
    for s in range(2):
        its = step(x.to(dev), batch)
        lr, ir, _, _ = O.train_step(sd, x, batch, state)
        print(f"  step {s + 2}: items hip {its.cpu().tolist()} ref {ir.tolist()}", flush=True)


def sect_nms():
    from oracle import nms_ref, synth
    pred = synth.nms_pred(7)
    t0 = time.time()
    rows, index, counts = E.nms(torch.from_numpy(pred).to(dev), 0.25, 0.7, 300)
    torch.cuda.synchronize()
    ref = nms_ref.non_max_suppression(pred, 0.25, 0.7, 300)
    for b in range(pred.shape[0]):
        k = int(counts[b])
        same = k == len(ref[b][1]) and np.array_equal(index[b, :k].cpu().numpy(), ref[b][1])
        rows_same = k == len(ref[b][1]) and np.array_equal(rows[b, :k].cpu().numpy(), ref[b][0])
        print(f"nms image {b}: kept {k} ref {len(ref[b][1])} indices equal {same} rows bit-equal {rows_same}")
    rows, index, counts = E.nms(torch.from_numpy(pred).to(dev), 0.001, 0.7, 300)
    ref = nms_ref.non_max_suppression(pred, 0.001, 0.7, 300)
    print("nms conf=0.001:", [int(c) for c in counts], "equal", [np.array_equal(index[b, :int(counts[b])].cpu().numpy(), ref[b][1]) for b in range(2)], flush=True)


def sect_perf():
    """quick timing of fwd / fwd+bwd at bs=32 640 (not the bench; just orientation)"""
    from oracle import synth
    from computervision.pytorch_amd.train import FlatAdam, FusedTrainStep, V8DetectionLoss
    from configs import Yolo8DetConfig
    m = _model(None).train()
    cfg = Yolo8DetConfig()
    crit = V8DetectionLoss(cfg, m)
    step = FusedTrainStep(m, crit, FlatAdam(m))
    B = 32
    x = synth.images(B, 640, 640, seed=1).to(dev)
    batch = synth.targets(B, seed=2)
    for _ in range(3):
        its = step(x, batch)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 10
    for _ in range(n):
        its = step(x, batch)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    print(f"train step bs{B} 640: {dt * 1e3:.2f} ms -> {B / dt:.0f} img/s; items {its.cpu().tolist()}; ws {m._last_engine.workspace_bytes() / 2**30:.2f} GiB", flush=True)
    m.eval()
    with torch.no_grad():
        for _ in range(3):
            m._run_forward(x, False)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(n):
            m._run_forward(x, False)
        torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    print(f"eval fwd bs{B} 640: {dt * 1e3:.2f} ms -> {B / dt:.0f} img/s", flush=True)


def sect_train2():
    """gradient error anatomy: (a) engine fwd + oracle dpred, (b) per-layer listing, (c) loss-scale sweep"""
    from oracle import synth
    from oracle import yolov8_ref as O
    from computervision.pytorch_amd.train import flatten_targets, V8DetectionLoss
    from configs import Yolo8DetConfig
    B, H = 4, 160
    x, batch = synth.images(B, H, H, seed=1), synth.targets(B, seed=2)
    sd = O.init_state_dict("n", 80, seed=0)
    keys = O.trainable_keys(sd)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = {k: v.clone() for k, v in sd.items()}
    work.update(leaves)
    feats = O.forward(work, x, "n", 80, training=True)
    for f in feats:
        f.retain_grad()
    loss, items = O.v8_loss(feats, batch, 80)
    loss.backward()
    gref = {k: leaves[k].grad for k in keys}
    dpred_ref = torch.cat([f.grad.reshape(B, 144, -1) for f in feats], 2).permute(0, 2, 1).contiguous()   # (B,A,144)
    m = _model(None).train()
    eng = m.engine_for(H, H)
    named = dict(m.named_parameters())
    for ls in (1024.0, 65536.0):
        pred = m._run_forward(x.to(dev), training=True)
        m.flat_grads.zero_()
        eng.backward((dpred_ref * ls).half().to(dev).contiguous(), ls)
        torch.cuda.synchronize()
        m.attach_grads()
        errs = [(k, rel(named[k].grad.cpu(), gref[k])[0], float(gref[k].norm())) for k in keys]
        gh = torch.cat([named[k].grad.cpu().flatten() for k in keys])
        gr = torch.cat([gref[k].flatten() for k in keys])
        print(f"[oracle dpred, loss_scale {ls}] global rel {rel(gh, gr)[0]:.3e} median {np.median([e[1] for e in errs]):.3e} dpred absmax*ls {float(dpred_ref.abs().max() * ls):.1f}")
        if ls == 1024.0:
            for k, e, n in errs:
                if k.endswith("conv.weight") or k.endswith(".2.weight"):
                    print(f"    {k:42s} rel {e:.3e} |g| {n:.3e}")
        # reset BN running stats drift does not matter here
    sys.stdout.flush()


def sect_anatomy():
    """per-layer activation and activation-gradient errors vs the oracle (oracle dpred fed to the engine)"""
    from oracle import synth
    from oracle import yolov8_ref as O
    B, H = 4, 160
    x, batch = synth.images(B, H, H, seed=1), synth.targets(B, seed=2)
    sd = O.init_state_dict("n", 80, seed=0)
    work = {k: v.clone() for k, v in sd.items()}
    taps = {}
    xin = x.clone()
    feats = O.forward(work, xin, "n", 80, training=True, taps=taps)
    keep = [i for i in taps if i in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 18, 21)]
    for i in keep:
        taps[i].retain_grad() if taps[i].requires_grad else None
    for f in feats:
        f.retain_grad() if f.requires_grad else None
    # need grads w.r.t. activations: make weights require grad so the graph exists
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in O.trainable_keys(sd)}
    work = {k: v.clone() for k, v in sd.items()}
    work.update(leaves)
    taps = {}
    feats = O.forward(work, x, "n", 80, training=True, taps=taps)
    for i in keep:
        taps[i].retain_grad()
    for f in feats:
        f.retain_grad()
    loss, items = O.v8_loss(feats, batch, 80)
    loss.backward()
    dpred_ref = torch.cat([f.grad.reshape(B, 144, -1) for f in feats], 2).permute(0, 2, 1).contiguous()
    m = _model(None).train()
    eng = m.engine_for(H, H)
    ls = 1024.0
    m._run_forward(x.to(dev), training=True)
    m.flat_grads.zero_()
    eng.backward((dpred_ref * ls).half().to(dev).contiguous(), ls)
    torch.cuda.synchronize()
    for i in keep:
        b, off, c = eng.graph.taps[i]
        act = eng.read_buffer(b, B)[..., off:off + c].float().cpu().permute(0, 3, 1, 2)
        grd = eng.read_buffer(b, B, grad=True)[..., off:off + c].float().cpu().permute(0, 3, 1, 2) / ls
        ea, eg = rel(act, taps[i].detach()), rel(grd, taps[i].grad)
        print(f"layer {i:2d}: act rel {ea[0]:.3e} | grad rel {eg[0]:.3e} (|g| {float(taps[i].grad.norm()):.3e})", flush=True)

SECTIONS = dict(conv=sect_conv, forward=sect_forward, loss=sect_loss, train=sect_train, nms=sect_nms, perf=sect_perf, train2=sect_train2, anatomy=sect_anatomy)




if __name__ == "__main__":
    names = sys.argv[1:] or list(SECTIONS)
    print("device:", torch.cuda.get_device_name(0), flush=True)
    for n in names:
        print(f"==== {n} ====", flush=True)
        try:
            SECTIONS[n]()
        except Exception:
            traceback.print_exc()
            print(f"!!!! section {n} failed", flush=True)
