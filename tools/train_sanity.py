"""Sanity run: N fused steps of every registered trainer on fresh synthetic batches; prints the loss curve, the loss scale and skipped steps.
    python tools/train_sanity.py [steps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import builder

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
for name, bs in (("yolo8_det", 16), ("deeplabv3plus", 8), ("centernet", 16), ("ssd", 16), ("yolo7", 8)):
    cfg, _, trainer_cls = builder.export_from_registry(name)
    cfg.train.batch_size = bs
    if hasattr(cfg.train, "pretrained"):
        cfg.train.pretrained = False
    torch.manual_seed(0)
    tr = trainer_cls(cfg, dev)
    tr.model.train()
    losses, it = [], iter(tr.train_dataloader)
    for s in range(steps):
        try:
            batch = next(it)
        except StopIteration:
            it = iter(tr.train_dataloader)
            batch = next(it)
        losses.append(float(tr.train_loop(batch, None)[0]))
    torch.cuda.synchronize()
    sc = getattr(tr._step, "scaler", None)
    if sc is not None:
        sc.poll()
    finite = bool(torch.isfinite(tr.model.flat_params).all())
    print(f"{name:14s} b{bs:<3d} first {losses[0]:.4f} min {min(losses):.4f} last5 {[round(v, 4) for v in losses[-5:]]} "
          f"scale {getattr(sc, 'scale', None)} skipped {getattr(sc, 'skipped', None)} params finite {finite}", flush=True)
    del tr
    torch.cuda.empty_cache()
