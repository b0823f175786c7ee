"""Headline benchmark: images/sec of the 640x640 YOLOv8-n TRAIN STEP (forward + v8 loss + backward + Adam)
on N MI355X of one node.  `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it
under torch.distributed.run (one rank per GPU, RCCL gradient all-reduce).  Rank 0 prints ONE JSON line.

* workload = BASELINE.json configs[1]: YOLOv8-n, batch 32 per GPU, 640x640 synthetic images (U[0,1), seed 1),
  3 synthetic boxes per image, random-init weights (seed 0); inputs are resident in HBM before the timed region
  (`--model s` = the per-rank workload of configs[2]);
* `roofline` (the contract roof is MFMA fp16 dense, 2516.6 TFLOP/s): the dominant kernel family on
  the critical path is the implicit-GEMM convolution -- the forward and data-gradient launches of conv_halo_kernel (3x3
  stride 1), conv_pw_kernel (1x1), conv_gemm_kernel (96+ output channels), conv_igemm_dma_kernel (the rest) and the fp32 stem.  `achieved` = their algorithmic
  FLOPs (2 x MACs of every launch) / their summed launch duration, measured live with HIP events on the
  engine's launch stream (cvx_engine_profile) in a window right after the timed steps; `frac` = achieved / peak.
  `hbm_view` is the same launches against the 8 TB/s roof (algorithmic bytes: fp16 input view + output + weights per launch,
  DESIGN.md section 4) -- every YOLOv8-n layer has 16..256 channels, below the 315 FLOP/B ridge, so this is the nearer roof.
  `traffic` = HBM bytes per launch of that family from rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE,
  tools/pmc_traffic.py): counters cannot be read inside this process, so the JSON under profiles/ carries the SHA-256 of the
  library it was measured on and is REFUSED (traffic = null) when it does not match the library that is running;
  `whole_step` = images/s x 26.140262 GFLOP / 2516.6 TF, the section-8(d) figure for the whole train step;
* `cpu_baseline`: the CPU oracle (torch-CPU fp32 restatement of the reference, kind "port") timed on this
  host's cores on a bounded sample (batch 8 train steps), rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

# Before the HIP runtime loads: one hardware queue per stream PRIORITY.  A process that puts more than four hardware queues to work pays
# 2.2-2.5x on every step on this runtime (tools/micro/queue_count_bench.hip); the engine's three streams have three
# different priorities, so this costs it nothing (same-box A/B of six workloads: equal within noise), while whatever further streams torch,
# RCCL or the caller put to work can then no longer push the process over the budget (six streams at work: 6.4 instead of 14.6 ms).  An explicit
# setting in the environment wins.
# ... for the single-process runs only: with more than one rank the default exchange launches its all-reduces on ProcessGroupNCCL's internal
# stream, which has the main stream's priority -- under one queue per priority the collective (and its wait for the weight gradients) would
# sit in the main chain's in-order queue.  Until an N-GPU A/B exists the runtime default stays for WORLD_SIZE > 1 (CVX_BENCH_HWQ overrides).
if os.environ.get("CVX_BENCH_HWQ"):
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ["CVX_BENCH_HWQ"]
elif int(os.environ.get("WORLD_SIZE", "1")) <= 1:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_FP16_PEAK_TFLOPS = 2516.6                  # MI355X dense fp16
HBM_PEAK_GBS = 8000.0
GFLOP = {"n": (8.742912, 26.140262), "s": (28.601549, 85.627699)}   # (forward, train step) per image, SURVEY section 8(d)
TRAFFIC_FILES = ("r05_conv_traffic.json", "r04_conv_traffic.json", "r03_conv_traffic.json", "r02_conv_traffic.json")      # newest first; PMC passes of a build, keyed by that build's library hash
MFMA_FILES = ("r05_mfma_busy.json", "r04_mfma_busy.json", "r03_mfma_busy.json")                                  # SQ_VALU_MFMA_BUSY_CYCLES pass (tools/pmc_mfma.py), same keying


def src_sha256():
    """Hash of the kernel sources (csrc/*.hip, *.h + the ABI header): what the PMC files of profiles/ are keyed on.  (The library's own hash
    changes with the build directory -- hipcc compilation-unit ids -- so a rebuild elsewhere silently dropped `traffic`.)"""
    import hashlib
    import glob
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "computervision.pytorch_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(ROOT, "include", "cvx_engine.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def lib_sha256():
    import hashlib
    from computervision.pytorch_amd import _lib
    return hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()


def measured_traffic(model, batch):
    """HBM bytes per conv launch from the committed PMC passes -- only if they were taken on THIS build of the library."""
    sha = lib_sha256()
    for name in TRAFFIC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        t = json.load(open(path))
        if (t.get("src_sha256") == src_sha256() or t.get("lib_sha256") == sha) and t.get("model", "n") == model and t.get("batch", 32) == batch:
            return round(t["hbm_bytes_per_launch"]), f"profiles/{name} (library {sha[:12]})"
        return None, f"profiles/{name} was measured on library {str(t.get('lib_sha256'))[:12]}, running {sha[:12]}: refused"
    return None, "no PMC passes committed for this round yet"


def measured_mfma_busy():
    """MFMA-busy fraction of the conv kernels from the committed PMC pass (tools/pmc_mfma.py) -- only if taken on THIS build."""
    sha = lib_sha256()
    for name in MFMA_FILES:
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        t = json.load(open(path))
        if t.get("src_sha256") == src_sha256() or t.get("lib_sha256") == sha:
            return t.get("conv_mfma_busy_frac"), f"profiles/{name} (library {sha[:12]})"
        return None, f"profiles/{name} was measured on library {str(t.get('lib_sha256'))[:12]}, running {sha[:12]}: refused"
    return None, "no MFMA-busy PMC pass committed for this round yet"


def usable_cores() -> int:
    """Cores this process may actually use: the cgroup CPU quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(seconds_budget: float = 25.0):
    from computervision.pytorch_amd import synth
    from oracle import yolov8_ref as O     # the ONLY use of oracle/ in this file: the CPU baseline leg
    bs = 32                                                   # the headline's own batch (rounds 1-4 sampled batch 8)
    torch.set_num_threads(usable_cores())                     # torch defaults to every core of the host, not this box's share
    x, batch = synth.images(bs, 640, 640, seed=1), synth.targets(bs, seed=2)
    sd, state = O.init_state_dict("n", 80, seed=0), {}
    O.train_step(sd, x, batch, state)                         # warm-up (allocator, threads)
    n, t0 = 0, time.time()
    while n < 2 or (time.time() - t0 < seconds_budget and n < 8):
        O.train_step(sd, x, batch, state)
        n += 1
    dt = (time.time() - t0) / n
    return {"value": round(bs / dt, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} YOLOv8-n 640x640 train steps at batch {bs} (fwd+loss+bwd+Adam), torch-CPU fp32 oracle, {dt * 1e3:.0f} ms/step"}


CENTERNET_GFLOP_PER_IMG = 62.24          # DLA-34, nc 80, 512x512 forward


def centernet_main(args):
    """images/sec of CenterNet DLA-34 inference (engine forward + cvx_centernet_decode) on synthetic 512x512 batches; one
    process per GPU, images sharded with no exchange (inference shards by image)."""
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import builder
    from computervision.pytorch_amd import synth
    cfg, algo_cls, _ = builder.export_from_registry("centernet")
    cfg.dataset.num_classes = 80
    cfg.arch.input_size = (3, 512, 512)
    algo = algo_cls(cfg, dev)
    torch.manual_seed(0)
    model, _ = algo.build_model()
    model = model.to(dev).eval()
    B = args.batch if args.batch != 32 else 64
    x = synth.images(B, 512, 512, seed=1 + rank).to(dev)

    def step():
        with torch.no_grad():
            raw = model.forward_raw(x)
            return algo.decode_raw(raw, 128, 128)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 1)):
        out = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    eng = model._last_engine
    eng.profile(True)
    for _ in range(3):
        step()
    sync()
    prof = eng.profile_read()
    eng.profile(False)
    if rank == 0:
        value = B * world * args.steps / elapsed
        conv = prof["conv_fwd"]
        tf = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
        print(json.dumps({
            "metric": "images/sec 512x512 CenterNet DLA-34 inference + decode", "value": round(value, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"weights": "constant between forwards: the engine keeps its fp16 weight images (cvx_engine_keep_shadows), BatchNorm folding still runs every forward", "workload": f"CenterNet DLA-34 (nc 80) inference + heat-map decode (top-100, DIoU-NMS), batch {B}/GPU, 512x512, random init",
                       "global_batch": B * world, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": "implicit-GEMM convolution forward launches (conv_halo / conv_pw / conv_igemm_dma)",
                         "achieved": round(tf, 3), "peak": MFMA_FP16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_FP16_PEAK_TFLOPS, 5),
                         "traffic": None, "avg_launch_us": round(conv["ms"] * 1e3 / max(conv["launches"], 1), 3),
                         "launches_per_step": conv["launches"] // 3},
            "whole_step": {"tflops": round(value / world * CENTERNET_GFLOP_PER_IMG / 1e3, 3),
                           "frac_of_mfma_peak": round(value / world * CENTERNET_GFLOP_PER_IMG / 1e3 / MFMA_FP16_PEAK_TFLOPS, 5)},
            "kernel_classes": {k: {"ms_per_step": round(v["ms"] / 3, 4), "launches_per_step": v["launches"] // 3} for k, v in prof.items() if v["launches"]},
            "detections_image0": int(out["counts"][0]), "engine_workspace_gib": round(eng.workspace_bytes() / 2 ** 30, 3)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


DEEPLAB_GFLOP_PER_IMG = 166.0   # DeepLabv3+ R101 @ 513x513 (2 x MACs of the convolutions, probe)


def deeplab_main(args):
    """images/sec of DeepLabv3+ ResNet-101 INFERENCE (engine forward + final bilinear resize to (B, 21, 513, 513)) on synthetic
    513x513 batches of 16 (the forward half of BASELINE.json configs[5]; `--workload deeplab_train` is the train step); one process per GPU, images sharded with no exchange."""
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import builder
    from computervision.pytorch_amd import synth
    cfg, algo_cls, _ = builder.export_from_registry("deeplabv3plus")
    algo = algo_cls(cfg, dev)
    torch.manual_seed(0)
    model, _ = algo.build_model()
    with torch.no_grad():                                       # a conditioned network (see oracle/make_golden.py section 10): timing only
        for k, v in model.state_dict().items():
            if k.endswith(".bn3.weight"):
                v.fill_(0.1)
    model = model.to(dev).eval()
    B = args.batch if args.batch != 32 else 16
    H, W = cfg.arch.input_size[1:]
    x = synth.images(B, H, W, seed=1 + rank).to(dev)

    def step():
        with torch.no_grad():
            return model(x)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 1)):
        out = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    eng = model._last_engine
    eng.profile(True)
    for _ in range(3):
        step()
    sync()
    prof = eng.profile_read()
    eng.profile(False)
    if rank == 0:
        value = B * world * args.steps / elapsed
        conv = prof["conv_fwd"]
        tf = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
        print(json.dumps({
            "metric": "images/sec 513x513 DeepLabv3+ R101 inference", "value": round(value, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"weights": "constant between forwards: the engine keeps its fp16 weight images (cvx_engine_keep_shadows), BatchNorm folding still runs every forward", "workload": f"DeepLabv3+ ResNet-101 (output stride 16, nc 21) eval forward + bilinear resize to the input size, batch {B}/GPU, "
                                   f"{H}x{W}, random init", "global_batch": B * world, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": "implicit-GEMM convolution forward launches (conv_halo / conv_pw / conv_igemm_dma)",
                         "achieved": round(tf, 3), "peak": MFMA_FP16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_FP16_PEAK_TFLOPS, 5),
                         "traffic": None, "avg_launch_us": round(conv["ms"] * 1e3 / max(conv["launches"], 1), 3),
                         "launches_per_step": conv["launches"] // 3},
            "whole_step": {"tflops": round(value / world * DEEPLAB_GFLOP_PER_IMG / 1e3, 3),
                           "frac_of_mfma_peak": round(value / world * DEEPLAB_GFLOP_PER_IMG / 1e3 / MFMA_FP16_PEAK_TFLOPS, 5)},
            "kernel_classes": {k: {"ms_per_step": round(v["ms"] / 3, 4), "launches_per_step": v["launches"] // 3} for k, v in prof.items() if v["launches"]},
            "output_shape": list(out.shape), "engine_workspace_gib": round(eng.workspace_bytes() / 2 ** 30, 3)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def deeplab_train_main(args):
    """images/sec of the DeepLabv3+ ResNet-101 TRAIN step (BASELINE.json configs[5]: 513x513, batch 16 per GPU, FocalLoss, Adam,
    mixed precision): engine forward (batch-statistics BN, dropout) + cvx_seg_loss + engine backward + [RCCL gradient sum] + fused
    Adam with the GradScaler check -- the reference's train_loop (segmentation_trainer.py:114-131).  One process per GPU, a batch per
    rank, gradients summed over the ranks after the backward pass."""
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29536")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import builder
    from core.trainer.segmentation_trainer import SyntheticSegmentationLoader
    cfg, _, trainer_cls = builder.export_from_registry("deeplabv3plus")
    B = args.batch if args.batch != 32 else cfg.train.batch_size
    cfg.train.batch_size = B
    H, W = cfg.arch.input_size[1:]
    torch.manual_seed(0)
    tr = trainer_cls(cfg, dev, dataloader=SyntheticSegmentationLoader(B, (H, W), cfg.dataset.num_classes, length=1, seed=1 + rank))
    tr.model.train()
    images, targets = next(iter(tr.train_dataloader))
    batch = (images.to(dev), targets.to(dev))

    def step():
        return tr.train_loop(batch, None)[0]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 1)):
        loss = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    eng = tr.model._last_engine
    eng.profile(True)
    for _ in range(2):
        step()
    sync()
    prof = eng.profile_read()
    eng.profile(False)
    tr._step.scaler.poll()
    if rank == 0:
        value = B * world * args.steps / elapsed
        classes = ("conv_fwd", "conv_dgrad", "conv_wgrad")
        ms = sum(prof[k]["ms"] for k in classes)
        fl = sum(prof[k]["flops"] for k in classes)
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        launches = sum(prof[k]["launches"] for k in classes)
        step_tf = value / world * 3 * DEEPLAB_GFLOP_PER_IMG / 1e3
        print(json.dumps({
            "metric": "images/sec 513x513 DeepLabv3+ R101 train", "value": round(value, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"DeepLabv3+ ResNet-101 (output stride 16, nc 21) train step (fwd + focal loss + bwd + Adam, dropout 0.1, dynamic "
                                   f"loss scale), batch {B}/GPU, {H}x{W}, random init", "global_batch": B * world, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": "implicit-GEMM convolution launches: forward, data gradient, weight gradient",
                         "achieved": round(tf, 3), "peak": MFMA_FP16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_FP16_PEAK_TFLOPS, 5),
                         "traffic": None, "avg_launch_us": round(ms * 1e3 / max(launches, 1), 3), "launches_per_step": launches // 2},
            "whole_step": {"tflops": round(step_tf, 3), "frac_of_mfma_peak": round(step_tf / MFMA_FP16_PEAK_TFLOPS, 5),
                           "note": "3 x the forward's 2*MAC count per image (forward + data gradient + weight gradient)"},
            "kernel_classes": {k: {"ms_per_step": round(v["ms"] / 2, 4), "launches_per_step": v["launches"] // 2,
                                   "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 and v["flops"] > 0 else None}
                               for k, v in prof.items() if v["launches"]},
            "loss": round(float(loss), 5), "loss_scale": tr._step.scaler.scale, "skipped_steps": tr._step.scaler.skipped,
            "engine_workspace_gib": round(eng.workspace_bytes() / 2 ** 30, 3)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def centernet_train_main(args):
    """images/sec of the CenterNet DLA-34 TRAIN step at the reference's training configuration (configs/centernet_cfg.py: 384x384, batch 16,
    CombinedLoss, Adam, mixed precision): engine forward (batch-statistics BN) + cvx_centernet_loss + engine backward + [RCCL gradient sum]
    + fused Adam -- the reference's train_loop (centernet_train.py:104-121).  One process per GPU, a batch per rank."""
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29537")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import builder
    from core.trainer.centernet_train import SyntheticCenterNetLoader
    cfg, _, trainer_cls = builder.export_from_registry("centernet")
    B = args.batch if args.batch != 32 else cfg.train.batch_size
    cfg.train.batch_size = B
    H, W = cfg.arch.input_size[1:]
    torch.manual_seed(0)
    tr = trainer_cls(cfg, dev, dataloader=SyntheticCenterNetLoader(B, (H, W), cfg.dataset.num_classes, length=1, seed=1 + rank))
    tr.model.train()
    images, targets = next(iter(tr.train_dataloader))
    batch = (images.to(dev), [t.to(dev) for t in targets])

    def step():
        return tr.train_loop(batch, None)[0]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 1)):
        loss = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    eng = tr.model._last_engine
    eng.profile(True)
    for _ in range(2):
        step()
    sync()
    prof = eng.profile_read()
    eng.profile(False)
    tr._step.scaler.poll()
    if rank == 0:
        value = B * world * args.steps / elapsed
        classes = ("conv_fwd", "conv_dgrad", "conv_wgrad")
        ms = sum(prof[k]["ms"] for k in classes)
        fl = sum(prof[k]["flops"] for k in classes)
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        launches = sum(prof[k]["launches"] for k in classes)
        print(json.dumps({
            "metric": f"images/sec {H}x{W} CenterNet DLA-34 train", "value": round(value, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"CenterNet DLA-34 (nc {cfg.dataset.num_classes}) train step (fwd + CombinedLoss + bwd + Adam, dynamic loss scale), "
                                   f"batch {B}/GPU, {H}x{W}, random init", "global_batch": B * world, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": "implicit-GEMM convolution launches: forward, data gradient, weight gradient",
                         "achieved": round(tf, 3), "peak": MFMA_FP16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_FP16_PEAK_TFLOPS, 5),
                         "traffic": None, "avg_launch_us": round(ms * 1e3 / max(launches, 1), 3), "launches_per_step": launches // 2},
            "kernel_classes": {k: {"ms_per_step": round(v["ms"] / 2, 4), "launches_per_step": v["launches"] // 2} for k, v in prof.items() if v["launches"]},
            "loss": round(float(loss), 5), "loss_scale": tr._step.scaler.scale, "skipped_steps": tr._step.scaler.skipped,
            "engine_workspace_gib": round(eng.workspace_bytes() / 2 ** 30, 3)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def ssd_train_main(args):
    """images/sec of the SSD300 VGG16-BN TRAIN step (configs/ssd_cfg.py: 300x300, MultiBoxLossV2, Adam, mixed precision): engine forward
    (batch-statistics BN) + cvx_multibox_loss + engine backward + [RCCL gradient sum] + fused Adam -- the reference's train_loop
    (ssd_train.py:96-115).  One process per GPU, a batch per rank."""
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29538")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import builder
    from core.trainer.ssd_train import SyntheticSsdLoader
    cfg, _, trainer_cls = builder.export_from_registry("ssd")
    cfg.train.pretrained = False
    B = args.batch
    cfg.train.batch_size = B
    H, W = cfg.arch.input_size[1:]
    torch.manual_seed(0)
    tr = trainer_cls(cfg, dev, dataloader=SyntheticSsdLoader(B, (H, W), cfg.dataset.num_classes, length=1, seed=1 + rank))
    tr.model.train()
    images, y_true = next(iter(tr.train_dataloader))
    batch = (images.to(dev), y_true.to(dev))

    def step():
        return tr.train_loop(batch, None)[0]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 1)):
        loss = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    eng = tr.model._last_engine
    eng.profile(True)
    for _ in range(2):
        step()
    sync()
    prof = eng.profile_read()
    eng.profile(False)
    tr._step.scaler.poll()
    if rank == 0:
        value = B * world * args.steps / elapsed
        classes = ("conv_fwd", "conv_dgrad", "conv_wgrad")
        ms = sum(prof[k]["ms"] for k in classes)
        fl = sum(prof[k]["flops"] for k in classes)
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        launches = sum(prof[k]["launches"] for k in classes)
        print(json.dumps({
            "metric": f"images/sec {H}x{W} SSD300 VGG16-BN train", "value": round(value, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"SSD300 VGG16-BN (nc {cfg.dataset.num_classes}) train step (fwd + MultiBoxLossV2 + bwd + Adam, dynamic loss scale), "
                                   f"batch {B}/GPU, {H}x{W}, random init", "global_batch": B * world, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": "implicit-GEMM convolution launches: forward, data gradient, weight gradient",
                         "achieved": round(tf, 3), "peak": MFMA_FP16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_FP16_PEAK_TFLOPS, 5),
                         "traffic": None, "avg_launch_us": round(ms * 1e3 / max(launches, 1), 3), "launches_per_step": launches // 2},
            "kernel_classes": {k: {"ms_per_step": round(v["ms"] / 2, 4), "launches_per_step": v["launches"] // 2} for k, v in prof.items() if v["launches"]},
            "loss": round(float(loss), 5), "loss_scale": tr._step.scaler.scale, "skipped_steps": tr._step.scaler.skipped,
            "engine_workspace_gib": round(eng.workspace_bytes() / 2 ** 30, 3)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def yolov7_train_main(args):
    """images/sec of the YOLOv7-l TRAIN step (configs/yolo7_cfg.py: 640x640, Yolo7Loss with SimOTA assignment, Adam, mixed precision):
    engine forward (batch-statistics BN) + cvx_yolo7_loss + engine backward + [RCCL gradient sum] + fused Adam -- the reference's train_loop
    (yolo7_train.py:79-97).  One process per GPU, a batch per rank."""
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29539")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import builder
    from core.trainer.yolo7_train import SyntheticYolo7Loader
    cfg, _, trainer_cls = builder.export_from_registry("yolo7")
    cfg.train.pretrained = False
    B = args.batch
    cfg.train.batch_size = B
    H, W = cfg.arch.input_size[1:]
    torch.manual_seed(0)
    tr = trainer_cls(cfg, dev, dataloader=SyntheticYolo7Loader(B, (H, W), cfg.dataset.num_classes, length=1, seed=1 + rank))
    tr.model.train()
    images, targets = next(iter(tr.train_dataloader))
    batch = (images.to(dev), targets.to(dev))

    def step():
        return tr.train_loop(batch, None)[0]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 1)):
        loss = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    eng = tr.model._last_engine
    eng.profile(True)
    for _ in range(2):
        step()
    sync()
    prof = eng.profile_read()
    eng.profile(False)
    tr._step.scaler.poll()
    if rank == 0:
        value = B * world * args.steps / elapsed
        classes = ("conv_fwd", "conv_dgrad", "conv_wgrad")
        ms = sum(prof[k]["ms"] for k in classes)
        fl = sum(prof[k]["flops"] for k in classes)
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        launches = sum(prof[k]["launches"] for k in classes)
        print(json.dumps({
            "metric": f"images/sec {H}x{W} YOLOv7-l train", "value": round(value, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"YOLOv7-l (nc {cfg.dataset.num_classes}) train step (fwd + Yolo7Loss with SimOTA + bwd + Adam, dynamic loss scale), "
                                   f"batch {B}/GPU, {H}x{W}, random init", "global_batch": B * world, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": "implicit-GEMM convolution launches: forward, data gradient, weight gradient",
                         "achieved": round(tf, 3), "peak": MFMA_FP16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_FP16_PEAK_TFLOPS, 5),
                         "traffic": None, "avg_launch_us": round(ms * 1e3 / max(launches, 1), 3), "launches_per_step": launches // 2},
            "kernel_classes": {k: {"ms_per_step": round(v["ms"] / 2, 4), "launches_per_step": v["launches"] // 2} for k, v in prof.items() if v["launches"]},
            "loss": round(float(loss), 5), "loss_scale": tr._step.scaler.scale, "skipped_steps": tr._step.scaler.skipped,
            "assignment_overflow": tr.criterion.overflowed(), "engine_workspace_gib": round(eng.workspace_bytes() / 2 ** 30, 3)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


YOLOV7_GFLOP_PER_IMG = 105.8    # YOLOv7-l @ 640x640 (probe at nc = 80; the VOC head is 0.3 % smaller)


def yolov7_main(args):
    """images/sec of YOLOv7-l INFERENCE (engine forward + anchor decode + per-class NMS) on synthetic 640x640 batches of 32; one
    process per GPU, images sharded with no exchange."""
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29535")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import builder
    from computervision.pytorch_amd import synth
    cfg, algo_cls, _ = builder.export_from_registry("yolo7")
    cfg.train.pretrained = False
    algo = algo_cls(cfg, dev)
    torch.manual_seed(0)
    model, _ = algo.build_model()
    model = model.to(dev).eval()
    B = args.batch
    H, W = cfg.arch.input_size[1:]
    x = synth.images(B, H, W, seed=1 + rank).to(dev)

    # --nms-load K: K candidates per image planted into the head rows after every forward (random-init logits are ~0: objectness * class =
    # 0.25 everywhere, nothing passes 0.6 and the suppression kernel would idle).  K/4 clusters of 2 x 2 neighbouring cells of the first
    # level, first anchor, one class per cluster, objectness and class logit +4 (score 0.96): neighbours' boxes overlap, so every cluster
    # is real suppression work; the scatter is two tiny device ops inside the timed step.
    plant = None
    if args.nms_load > 0:
        g = torch.Generator().manual_seed(7)
        ncl = max(args.nms_load // 4, 1)
        with torch.no_grad():
            model.forward_rows(x[:1])
        lh, lw = model._last_engine.graph.level_hw[0]             # the first level of the head rows (its rows start at 0)
        cy, cx = torch.randint(0, lh - 1, (B, ncl), generator=g), torch.randint(0, lw - 1, (B, ncl), generator=g)
        cls = torch.randint(0, cfg.dataset.num_classes, (B, ncl), generator=g)
        pix = torch.stack([(cy + dy) * lw + (cx + dx) for dy in (0, 1) for dx in (0, 1)], 2).reshape(B, -1)       # (B, 4 * ncl) rows of level 0
        plant = (torch.arange(B).unsqueeze(1).expand_as(pix).to(dev), pix.to(dev), (5 + cls).repeat_interleave(4, 1).to(dev))

    def step():
        with torch.no_grad():
            rows = model.forward_rows(x)
            if plant is not None:
                rows[plant[0], plant[1], 4] = 4.0
                rows[plant[0], plant[1], plant[2]] = 4.0
            dec, y = algo.decode_rows(model, rows)
            return algo.nms_device(y, dec, 0.6)      # (without --nms-load: untrained logits ~ 0, nothing passes the threshold)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 1)):
        out = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    eng = model._last_engine
    eng.profile(True)
    for _ in range(3):
        step()
    sync()
    prof = eng.profile_read()
    eng.profile(False)
    planted_in = int(plant[1].shape[1]) if plant is not None else 0
    kept_out = round(sum(0 if d is None else int(d.shape[0]) for d, _ in out) / B, 1)
    if rank == 0:
        value = B * world * args.steps / elapsed
        conv = prof["conv_fwd"]
        tf = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
        print(json.dumps({
            "metric": ("images/sec 640x640 YOLOv7-l inference + decode + NMS (" + (f"{planted_in} planted candidates/image above the threshold, {kept_out} kept" if plant is not None else
                       "0 candidates pass the threshold at random init: suppression not exercised") + ")"), "value": round(value, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"weights": "constant between forwards: the engine keeps its fp16 weight images (cvx_engine_keep_shadows), BatchNorm folding still runs every forward", "workload": f"YOLOv7-l (nc 20) eval forward + anchor decode + per-class NMS, batch {B}/GPU, {H}x{W}, random init",
                       "global_batch": B * world, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": "implicit-GEMM convolution forward launches (conv_halo / conv_pw / conv_igemm_dma)",
                         "achieved": round(tf, 3), "peak": MFMA_FP16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_FP16_PEAK_TFLOPS, 5),
                         "traffic": None, "avg_launch_us": round(conv["ms"] * 1e3 / max(conv["launches"], 1), 3),
                         "launches_per_step": conv["launches"] // 3},
            "whole_step": {"tflops": round(value / world * YOLOV7_GFLOP_PER_IMG / 1e3, 3),
                           "frac_of_mfma_peak": round(value / world * YOLOV7_GFLOP_PER_IMG / 1e3 / MFMA_FP16_PEAK_TFLOPS, 5)},
            "kernel_classes": {k: {"ms_per_step": round(v["ms"] / 3, 4), "launches_per_step": v["launches"] // 3} for k, v in prof.items() if v["launches"]},
            "engine_workspace_gib": round(eng.workspace_bytes() / 2 ** 30, 3)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


SSD_GFLOP_PER_IMG = 62.8        # SSD300 VGG16 @ 300x300 (probe)


def ssd_main(args):
    """images/sec of SSD300 (VGG16-BN) INFERENCE (engine forward + NCHW-order head tensors + softmax / prior decode) on synthetic
    300x300 batches of 32; one process per GPU, images sharded with no exchange.  (At random init no class score passes the 0.7
    threshold, so the per-class NMS launches are skipped, as they would be on background images.)"""
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29536")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import builder
    from computervision.pytorch_amd import synth
    cfg, algo_cls, _ = builder.export_from_registry("ssd")
    algo = algo_cls(cfg, dev)
    torch.manual_seed(0)
    model, _ = algo.build_model()
    model = model.to(dev).eval()
    B = args.batch
    x = synth.images(B, 300, 300, seed=1 + rank).to(dev)

    # --nms-load K: K candidates per image planted into the class logits after every forward (at random init no score passes 0.7 and the
    # per-class NMS launches are skipped): K/4 runs of four consecutive priors of the 38 x 38 level (the four aspect ratios of one cell --
    # overlapping boxes), one class per run, logit +8 (softmax 0.99)
    plant = None
    if args.nms_load > 0:
        g = torch.Generator().manual_seed(7)
        ncl = max(args.nms_load // 4, 1)
        base = torch.randint(0, 38 * 38, (B, ncl), generator=g) * 4
        cls = torch.randint(1, cfg.dataset.num_classes + 1, (B, ncl), generator=g)
        pri = (base.unsqueeze(2) + torch.arange(4)).reshape(B, -1)
        plant = (torch.arange(B).unsqueeze(1).expand_as(pri).to(dev), pri.to(dev), cls.repeat_interleave(4, 1).to(dev))

    def step():
        with torch.no_grad():
            loc, conf = model(x)
            if plant is not None:
                conf = conf.view(B, -1, cfg.dataset.num_classes + 1)
                conf[plant[0], plant[1], plant[2]] = 8.0
            return algo.decode_device((loc, conf))

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 1)):
        out = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    eng = model._last_engine
    eng.profile(True)
    for _ in range(3):
        step()
    sync()
    prof = eng.profile_read()
    eng.profile(False)
    planted_in = int(plant[1].shape[1]) if plant is not None else 0
    kept_out = round(sum(int(d.shape[0]) for d, _ in out) / B, 1)
    if rank == 0:
        value = B * world * args.steps / elapsed
        conv = prof["conv_fwd"]
        tf = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
        print(json.dumps({
            "metric": ("images/sec 300x300 SSD300-VGG16 inference + decode + per-class NMS (" + (f"{planted_in} planted candidates/image above the threshold, {kept_out} kept" if plant is not None else
                       "no score passes the threshold at random init: NMS not exercised") + ")"), "value": round(value, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"weights": "constant between forwards: the engine keeps its fp16 weight images (cvx_engine_keep_shadows), BatchNorm folding still runs every forward", "workload": f"SSD300 VGG16-BN (nc 20) eval forward + softmax / prior decode (+ per-class NMS when a score passes), batch {B}/GPU, "
                                   "300x300, random init", "global_batch": B * world, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": "implicit-GEMM convolution forward launches (conv_halo / conv_pw / conv_igemm_dma)",
                         "achieved": round(tf, 3), "peak": MFMA_FP16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_FP16_PEAK_TFLOPS, 5),
                         "traffic": None, "avg_launch_us": round(conv["ms"] * 1e3 / max(conv["launches"], 1), 3),
                         "launches_per_step": conv["launches"] // 3},
            "whole_step": {"tflops": round(value / world * SSD_GFLOP_PER_IMG / 1e3, 3),
                           "frac_of_mfma_peak": round(value / world * SSD_GFLOP_PER_IMG / 1e3 / MFMA_FP16_PEAK_TFLOPS, 5)},
            "kernel_classes": {k: {"ms_per_step": round(v["ms"] / 3, 4), "launches_per_step": v["launches"] // 3} for k, v in prof.items() if v["launches"]},
            "engine_workspace_gib": round(eng.workspace_bytes() / 2 ** 30, 3)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def yolov8_eval_main(args):
    """images/sec of the YOLOv8-n EVAL forward (the pass north_star quotes the MFMA fraction of): engine forward with folded BatchNorm, the
    eval-mode fusion groups and the DFL decode to (B, 84, 8400) on synthetic 640x640 batches of 32; one process per GPU, images sharded
    with no exchange.  `--fusion 0` runs the same forward layer by layer (A/B of the cross-layer fusion on one box)."""
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29537")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from computervision.pytorch_amd.model import Yolo8
    from computervision.pytorch_amd import synth
    torch.manual_seed(0)
    model = Yolo8(args.model, 80).to(dev).eval()
    B = args.batch
    x = synth.images(B, 640, 640, seed=1 + rank).to(dev)

    def step():
        with torch.no_grad():
            return model(x)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    step()
    eng = model._last_engine
    if args.fusion:  # tuning build only (CVX_LIB=build/libcvx_tuning.so): the release library raises here
        eng.set_fusion(True)
    for _ in range(max(args.warmup, 1)):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    eng.profile(True)
    for _ in range(3):
        step()
    sync()
    prof = eng.profile_read()
    eng.profile(False)
    if rank == 0:
        fwd_gf = GFLOP.get(args.model, (float("nan"), float("nan")))[0]
        value = B * world * args.steps / elapsed
        conv = prof["conv_fwd"]
        tf = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
        launches = sum(v["launches"] for v in prof.values()) // 3
        print(json.dumps({
            "metric": f"images/sec 640x640 YOLOv8-{args.model} eval forward + decode", "value": round(value, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"weights": "constant between forwards: the engine keeps its fp16 weight images (cvx_engine_keep_shadows), BatchNorm folding still runs every forward", "workload": f"YOLOv8-{args.model} eval forward (folded BN + SiLU epilogues, fusion groups {'on' if args.fusion else 'off'}) + DFL decode, "
                                   f"batch {B}/GPU, 640x640, nc=80, random init", "global_batch": B * world, "parallelism": f"dp{world}",
                       "fused_groups": eng.fused_groups()},
            "roofline": {"bound": "mfma", "kernel": "convolution launches of the eval forward (conv_chain / conv_halo / conv_pw / conv_igemm_dma + the fp32 stem), "
                                                    "HIP events on the engine's streams, serialised window after the timed steps",
                         "achieved": round(tf, 3), "peak": MFMA_FP16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_FP16_PEAK_TFLOPS, 5),
                         "traffic": None, "avg_launch_us": round(conv["ms"] * 1e3 / max(conv["launches"], 1), 3),
                         "launches_per_step": conv["launches"] // 3},
            "whole_step": {"tflops": round(value / world * fwd_gf / 1e3, 3), "frac_of_mfma_peak": round(value / world * fwd_gf / 1e3 / MFMA_FP16_PEAK_TFLOPS, 5),
                           "gflop_per_image": fwd_gf, "engine_launches_per_forward": launches},
            "kernel_classes": {k: {"ms_per_step": round(v["ms"] / 3, 4), "launches_per_step": v["launches"] // 3} for k, v in prof.items() if v["launches"]},
            "engine_workspace_gib": round(eng.workspace_bytes() / 2 ** 30, 3)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--model", default="n")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nms-load", type=int, default=0, help="yolov7 / ssd: plant this many above-threshold candidates per image (clusters of four overlapping boxes) "
                                                            "into the head output after every forward, so that the suppression kernels have work at random init")
    ap.add_argument("--profile-steps", type=int, default=10, help="steps of the per-kernel HIP-event window after the timed region")
    ap.add_argument("--workload", default="yolov8_train", choices=["yolov8_train", "yolov8_eval", "centernet", "centernet_train", "deeplab", "deeplab_train", "yolov7", "yolov7_train", "ssd", "ssd_train"],
                    help="centernet: BASELINE.json configs[3] -- CenterNet DLA-34 (nc 80) 512x512 inference + heat-map decode, batch 64 per GPU")
    ap.add_argument("--exchange", default="torch", choices=["torch", "c"], help="N > 1: gradient exchange through torch.distributed (default) or "
                    "entirely behind the C ABI (RCCL communicator owned by the engine library, no Python between the backward ranges)")
    ap.add_argument("--fusion", type=int, default=0, help="yolov8_eval: 1 runs the eval forward with the cross-layer fusion groups (Bottleneck pairs, Detect levels as one launch each; measured 1-6 %% slower: TUNING build only, CVX_LIB=build/libcvx_tuning.so)")
    ap.add_argument("--stream", default="default", choices=["default", "own"], help="own: launch on a torch.cuda.Stream of the bench's own instead of "
                    "the legacy default stream (A/B switch for the hardware-queue budget, DESIGN.md section 6)")
    args = ap.parse_args()
    if args.stream == "own":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        torch.cuda.set_stream(torch.cuda.Stream())
    if args.workload == "yolov8_eval":
        return yolov8_eval_main(args)
    if args.workload == "centernet":
        return centernet_main(args)
    if args.workload == "deeplab":
        return deeplab_main(args)
    if args.workload == "deeplab_train":
        return deeplab_train_main(args)
    if args.workload == "centernet_train":
        return centernet_train_main(args)
    if args.workload == "ssd_train":
        return ssd_train_main(args)
    if args.workload == "yolov7_train":
        return yolov7_train_main(args)
    if args.workload == "yolov7":
        return yolov7_main(args)
    if args.workload == "ssd":
        return ssd_main(args)

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus N > 1 must be launched with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    use_dist = world > 1 or bool(os.environ.get("CVX_FORCE_DIST"))   # CVX_FORCE_DIST=1: 1-rank RCCL group (exchange-path rehearsal)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from computervision.pytorch_amd.model import Yolo8
    from computervision.pytorch_amd.train import FlatAdam, FusedTrainStep, V8DetectionLoss
    from configs import Yolo8DetConfig
    from computervision.pytorch_amd import synth

    cfg = Yolo8DetConfig()
    cfg.arch.model_type = args.model
    torch.manual_seed(0)
    model = Yolo8(args.model, 80, loss_scale=cfg.engine.loss_scale).to(dev).train()
    crit = V8DetectionLoss(cfg, model)
    comm = None
    if use_dist and args.exchange == "c":       # the exchange entirely behind the C ABI (cvx_engine_backward_exchange); default: torch.distributed
        from computervision.pytorch_amd.train import CvxComm
        comm = CvxComm(dev)
    step = FusedTrainStep(model, crit, FlatAdam(model, lr=cfg.train.initial_lr), n_buckets=cfg.engine.allreduce_buckets,
                          comm=comm)
    B = args.batch
    x = synth.images(B, 640, 640, seed=1 + rank).to(dev)
    batch = synth.targets(B, seed=2 + rank)
    batch = {k: v.to(dev) for k, v in batch.items()}         # labels resident in HBM too

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        items = step(x, batch)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        items = step(x, batch)
    sync()
    elapsed = time.perf_counter() - t0
    eng = model._last_engine                                  # exists after the first step (also with --warmup 0)
    # Per-kernel timing: HIP events recorded on the engine's launch streams around every kernel class.  The ~1000 event
    # records per step cost ~20 % of wall time, so they run on their own steps directly after the timed region (same
    # process, same buffers, same clocks) instead of inside it; `value` is never measured with them on.
    prof_steps = max(1, args.profile_steps)
    step(x, batch)
    sync()
    eng.profile(True)
    for _ in range(prof_steps):
        step(x, batch)
    sync()
    prof = eng.profile_read()
    eng.profile(False)
    # inference forward (BN folded into the conv epilogues, no BN passes): the north_star quotes its MFMA fraction
    model.eval()
    with torch.no_grad():
        for _ in range(3):
            model._run_forward(x, False)
        sync()
        t1 = time.perf_counter()
        n_eval = 20
        for _ in range(n_eval):
            model._run_forward(x, False)
        sync()
    eval_ms = (time.perf_counter() - t1) / n_eval * 1e3
    model.train()
    elapsed_rank0 = elapsed
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_items = [float(v) for v in items.cpu()]

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        fwd_gf, train_gf = GFLOP.get(args.model, (float('nan'), float('nan')))
        value = B * world * args.steps / elapsed
        conv = {k: prof[k] for k in ("conv_fwd", "conv_dgrad")}
        conv_ms = sum(v["ms"] for v in conv.values())
        conv_fl = sum(v["flops"] for v in conv.values())
        conv_launches = sum(v["launches"] for v in conv.values())
        achieved = conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        conv_by = sum(v["bytes"] for v in conv.values())
        conv_gbs = conv_by / (conv_ms * 1e-3) / 1e9 if conv_ms > 0 else 0.0
        traffic, traffic_note = measured_traffic(args.model, B)
        mfma_busy, mfma_note = measured_mfma_busy()
        classes = {}
        for k, v in prof.items():
            if v["launches"] == 0:
                continue
            sec = v["ms"] * 1e-3
            classes[k] = {"ms_per_step": round(v["ms"] / prof_steps, 4), "launches_per_step": v["launches"] // prof_steps,
                          "tflops": round(v["flops"] / sec / 1e12, 2) if v["flops"] else None,
                          "algorithmic_gbs": round(v["bytes"] / sec / 1e9, 1)}
        out = {
            "metric": f"images/sec 640x640 YOLOv8-{args.model} train step", "value": round(value, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"YOLOv8-{args.model} train step (fwd + v8 loss + bwd + Adam), batch {B}/GPU, 640x640, nc=80, random init",
                       "global_batch": B * world, "parallelism": f"dp{world}", "loss_scale": cfg.engine.loss_scale,
                       "launch": "eager",
                       # the first multi-GPU record should be read as N = 2 against N = 1: rank 0's own time beside the
                       # max over ranks, which exchange path ran, and the hardware-queue setting it ran under
                       "exchange": ("none (1 rank)" if not use_dist else ("C ABI: cvx_engine_backward_exchange (RCCL through dlopen)" if comm is not None
                                                                          else "torch.distributed all_reduce per bucket (ProcessGroupNCCL)")),
                       "ms_per_step_rank0": round(elapsed_rank0 / args.steps * 1e3, 4),
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default")},
            "roofline": {"bound": "mfma",
                         "kernel": "implicit-GEMM convolution, forward + data-gradient launches: conv_halo_kernel + conv_tile_kernel + conv_pw_kernel + "
                                   "conv_gemm_kernel + conv_igemm_dma_kernel (+ the fp32 stem passes)",
                         "achieved": round(achieved, 3), "peak": MFMA_FP16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_FP16_PEAK_TFLOPS, 5), "traffic": traffic, "traffic_source": traffic_note,
                         "mfma_busy_frac": mfma_busy, "mfma_busy_source": mfma_note,
                         "algorithmic_flops_per_launch": round(conv_fl / max(conv_launches, 1)),
                         "algorithmic_bytes_per_launch": round(conv_by / max(conv_launches, 1)),
                         "avg_launch_us": round(conv_ms * 1e3 / max(conv_launches, 1), 3), "launches_per_step": conv_launches // prof_steps,
                         "hbm_view": {"achieved_gbs": round(conv_gbs, 1), "peak_gbs": HBM_PEAK_GBS, "frac": round(conv_gbs / HBM_PEAK_GBS, 5)}},
            "whole_step": {"train_tflops": round(value / world * train_gf / 1e3, 3),
                           "frac_of_mfma_peak": round(value / world * train_gf / 1e3 / MFMA_FP16_PEAK_TFLOPS, 5),
                           "gflop_per_image": train_gf},
            "forward_eval": {"ms_per_batch": round(eval_ms, 4), "images_per_sec": round(B / eval_ms * 1e3, 1),
                             "tflops": round(B * fwd_gf / eval_ms, 2),
                             "frac_of_mfma_peak": round(B * fwd_gf / eval_ms / MFMA_FP16_PEAK_TFLOPS, 5),
                             "note": "per GPU; eval-mode forward of the same batch, folded BN + SiLU in the conv epilogues; weights constant between these forwards: fp16 weight images kept (cvx_engine_keep_shadows)"},
            "kernel_classes": classes,
            "loss_items_last_step": loss_items,
            "engine_workspace_gib": round(eng.workspace_bytes() / 2 ** 30, 3),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
