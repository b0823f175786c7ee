"""Loss, optimiser and the fused train step on top of the engine.

* ``V8DetectionLoss`` -- the reference's ``Loss(cfg, model)`` callable (core/algorithms/yolo_v8.py:25-124):
  ``loss, items = criterion(preds, batch)``; value and gradient come from ``cvx_loss_v8``.
* ``FlatAdam`` -- ``torch.optim.Optimizer``-shaped wrapper over ``cvx_adam_step`` on the flat arenas
  (the reference builds ``torch.optim.Adam`` over all parameters, core/trainer/lr_scheduler.py:37-43).
* ``FusedTrainStep`` -- zero_grad -> forward -> loss -> backward -> [all-reduce] -> Adam as five C-ABI
  calls with no per-parameter Python work (the reference's ``train_loop``,
  core/trainer/yolo8_train.py:93-111).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

import ctypes as C
import os

from . import _lib as L
from .engine import V8LossOp, _note_param_write, adam_step, adam_step_dev, check_finite
from .graph import STRIDES
from .model import PredList, Yolo8


def flatten_targets(batch: Dict[str, torch.Tensor], device) -> torch.Tensor:
    """yolo8_collate dict (core/data/collate.py:25-29) -> (N, 6) [batch_idx, cls, cx, cy, w, h] on `device`,
    grouped by image in stable order (what Loss.preprocess does row by row, yolo_v8.py:51-65).

    Device-resident dicts go through ``cvx_pack_targets`` (one small HIP kernel, no ATen sort/index/cat on the hot path);
    host dicts -- what a DataLoader hands over -- are ordered on the host and shipped with one copy."""
    n = int(batch["batch_idx"].numel())
    device = torch.device(device)
    if n == 0:
        return torch.zeros(0, 6, device=device)
    bi, cls, box = batch["batch_idx"].reshape(-1), batch["cls"].reshape(-1), batch["bboxes"].reshape(-1, 4)
    if bi.is_cuda and device.type == "cuda":
        bi, cls, box = (t if t.dtype == torch.float32 and t.is_contiguous() else t.float().contiguous() for t in (bi, cls, box))
        rows = torch.empty(n, 6, device=device)
        lib = L.load()
        L.check(lib.cvx_pack_targets(L.ptr(bi), L.ptr(cls), L.ptr(box), n, L.ptr(rows), L.stream_ptr(device)), "cvx_pack_targets")
        return rows
    t = torch.cat((bi.reshape(-1, 1).float().cpu(), cls.reshape(-1, 1).float().cpu(), box.float().cpu()), 1)
    order = torch.sort(t[:, 0], stable=True).indices
    return t[order].contiguous().to(device, non_blocking=True)


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, targets, owner, level_hw):
        items, dpred = owner.op(pred.detach(), targets, level_hw, STRIDES, owner.loss_scale)
        dpred = dpred[..., :pred.shape[2]]                       # row-padded buffer -> the gradient of pred itself
        ctx.save_for_backward(dpred)
        ctx.inv_scale = 1.0 / owner.loss_scale
        owner.last_items = items
        return items.sum() * pred.shape[0]                       # yolo_v8.py:124

    @staticmethod
    def backward(ctx, gout):
        (dpred,) = ctx.saved_tensors
        return dpred.float() * (gout * ctx.inv_scale), None, None, None


class V8DetectionLoss:
    def __init__(self, cfg, model: Yolo8):
        m = model.model[-1]
        self.nc, self.no, self.reg_max = m.nc, m.no, m.reg_max
        self.stride = m.stride
        self.box, self.cls, self.dfl = cfg.loss.box, cfg.loss.cls, cfg.loss.dfl
        self.loss_scale = float(getattr(getattr(cfg, "engine", None), "loss_scale", 1024.0))
        self.op = V8LossOp(self.nc, (self.box, self.cls, self.dfl))
        self.device = next(model.parameters()).device
        self.last_items = None

    def __call__(self, preds, batch):
        feats = preds[1] if isinstance(preds, tuple) else preds
        pred = getattr(feats, "pred", None)
        if pred is None:                                           # plain list of NCHW tensors: re-fuse (plumbing)
            b = feats[0].shape[0]
            pred = torch.cat([f.reshape(b, self.no, -1) for f in feats], 2).permute(0, 2, 1).contiguous()
            level_hw = [tuple(f.shape[2:]) for f in feats]
        else:
            level_hw = feats.level_hw
        targets = flatten_targets(batch, pred.device)
        loss = _LossFn.apply(pred, targets, self, level_hw)
        return loss, self.last_items.detach()


class FlatAdam(torch.optim.Optimizer):
    """Adam over the model's flat arenas; ``param_groups`` keeps torch's shape so LR schedulers work."""

    def __init__(self, model: Yolo8, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.model = model
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__([{"params": params, "initial_lr": lr}], dict(lr=lr, betas=betas, eps=eps))
        self._m = None
        self._v = None
        self._step = 0
        self._state = None          # device [lr, step, lr/(1-b1^t), 1/sqrt(1-b2^t)]: the step advances on the device
        self._lr_on_device = None
        self.found_inf: Optional[torch.Tensor] = None

    def _ensure_state(self):
        p = self.model.flat_params
        if self._m is None or self._m.device != p.device:
            self._m, self._v = torch.zeros_like(p), torch.zeros_like(p)
            self._state = torch.tensor([self.param_groups[0]["lr"], float(self._step), 0.0, 0.0], device=p.device)
            self._lr_on_device = self.param_groups[0]["lr"]

    def sync_lr(self):
        """Push a learning rate changed by a scheduler to the device state (call outside graph capture)."""
        self._ensure_state()
        lr = self.param_groups[0]["lr"]
        if lr != self._lr_on_device:
            self._state[0:1].fill_(lr)
            self._lr_on_device = lr

    @torch.no_grad()
    def step(self, closure=None, zero_grad: bool = False, grad_scale: float = 1.0):
        self._ensure_state()
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()               # a scheduler (or GradScaler.step -> optimizer.step) may have changed param_groups["lr"]
        g = self.param_groups[0]
        self._step += 1                  # host mirror only; the authoritative count lives in the device state
        adam_step_dev(self.model.flat_params, self.model.flat_grads, self._m, self._v, g["betas"], g["eps"], self._state, self.found_inf,
                      zero_grad, grad_scale)

    def zero_grad(self, set_to_none: bool = True):
        self.model.flat_grads.zero_()

    def device_step(self) -> int:
        """Steps actually applied: read from the device state (overflow-skipped steps and graph replays are counted there)."""
        return self._step if self._state is None else int(self._state[1].item())

    # ---- checkpoint interchange: torch.optim.Adam's state_dict layout (what the reference saves and loads, core/utils/ckpt.py:41-66) ----
    def _all_params(self):
        return list(self.model.parameters())   # the reference hands model.parameters() to Adam: frozen tensors (DFL) keep their index, without state

    def _moment_views(self, p):
        """the slices of the flat moment arenas that belong to parameter `p` (a strided view of the flat parameter arena)"""
        flat = self.model.flat_params
        if p.untyped_storage().data_ptr() != flat.untyped_storage().data_ptr():
            return None
        off = p.storage_offset() - flat.storage_offset()
        return (torch.as_strided(self._m, p.shape, p.stride(), off), torch.as_strided(self._v, p.shape, p.stride(), off))

    def state_dict(self):
        """torch.optim.Adam's layout: per-parameter ``state[i] = {step, exp_avg, exp_avg_sq}`` in ``model.parameters()`` order, with the
        reference's logical shapes (copies of the flat arenas' slices), so ``torch.optim.Adam(model.parameters()).load_state_dict`` of the
        reference accepts it -- and load_state_dict below accepts the reference's."""
        self._ensure_state()
        step = float(self.device_step())
        state = {}
        params = self._all_params()
        for i, p in enumerate(params):
            mv = self._moment_views(p) if p.requires_grad else None
            if mv is not None and step > 0:
                state[i] = {"step": torch.tensor(step), "exp_avg": mv[0].detach().clone().contiguous(), "exp_avg_sq": mv[1].detach().clone().contiguous()}
        g = self.param_groups[0]
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None,
                 "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                 "params": list(range(len(params)))}
        for k, v in g.items():                       # scheduler bookkeeping (initial_lr ...) rides along like in torch
            if k not in group and k != "params":
                group[k] = v
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts torch.optim.Adam's layout (a reference checkpoint, or state_dict() above) and the round-1/2 flat format
        ``{step, exp_avg, exp_avg_sq, param_groups}``."""
        for g, s in zip(self.param_groups, sd.get("param_groups", [])):
            g.update({k: v for k, v in s.items() if k != "params"})
        self._ensure_state()
        if "state" in sd:
            params = self._all_params()
            self._m.zero_()
            self._v.zero_()
            step = 0
            skipped = []
            for i, st in sd["state"].items():
                i = int(i)
                if i >= len(params):
                    raise L.CvxError(f"optimizer checkpoint has state for parameter {i}, the model has {len(params)} parameters")
                mv = self._moment_views(params[i])
                if mv is None:
                    skipped.append(i)
                    continue
                for key in ("exp_avg", "exp_avg_sq"):
                    if tuple(st[key].shape) != tuple(params[i].shape):
                        raise L.CvxError(f"optimizer checkpoint: {key} of parameter {i} has shape {tuple(st[key].shape)}, the model's is {tuple(params[i].shape)}")
                mv[0].copy_(st["exp_avg"])
                mv[1].copy_(st["exp_avg_sq"])
                step = max(step, int(float(st["step"])))
            self._step = step
            if skipped:
                import warnings
                warnings.warn(f"FlatAdam.load_state_dict: state of {len(skipped)} parameter(s) (indices {skipped[:8]}...) was NOT restored: they are "
                              "not views of the flat parameter arena (their moments start from zero)", stacklevel=2)
        else:
            self._step = int(sd["step"])
            if sd.get("exp_avg") is not None:
                self._m.copy_(sd["exp_avg"])
                self._v.copy_(sd["exp_avg_sq"])
            else:
                self._m.zero_()
                self._v.zero_()
        # rebuild the device state unconditionally: [lr, step, -, -] (the derived factors are recomputed by the next step)
        self._state.copy_(torch.tensor([self.param_groups[0]["lr"], float(self._step), 0.0, 0.0]))
        self._lr_on_device = self.param_groups[0]["lr"]


class DynamicLossScale:
    """torch.cuda.amp.GradScaler's policy (the reference's mixed-precision loop, core/trainer/yolo8_train.py:99-104,
    base.py:193-194) without its per-step host synchronisation: a step whose gradients contain inf/nan is skipped ON
    THE DEVICE (``found_inf`` read by the fused Adam kernel); the host learns about it through a pinned flag copied
    asynchronously and adjusts the scale when the copy has landed -- backoff x0.5 per observed overflow, growth x2 after
    ``growth_interval`` clean steps.  The only deviation from GradScaler: the new scale takes effect one or two steps
    after the overflow instead of on the very next step (those steps overflow again and are skipped as well)."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000,
                 min_scale=1.0, max_scale=2.0 ** 24):
        self.scale = float(init_scale)
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, int(growth_interval)
        self.min_scale, self.max_scale = float(min_scale), float(max_scale)
        self.device = torch.device(device)
        self.found_inf = torch.zeros(1, dtype=torch.int32, device=device)
        self._host = torch.zeros(1, dtype=torch.int32).pin_memory() if torch.device(device).type == "cuda" else torch.zeros(1, dtype=torch.int32)
        self._event = None
        self._good = 0
        self.skipped = 0

    def poll(self):
        """Apply the verdict of the last finished step, if its flag has arrived (never blocks)."""
        if self._event is None or not self._event.query():
            return
        self._event = None
        if int(self._host[0]) != 0:
            self.scale = max(self.scale * self.backoff_factor, self.min_scale)
            self._good = 0
            self.skipped += 1
        else:
            self._good += 1
            if self._good >= self.growth_interval:
                self.scale = min(self.scale * self.growth_factor, self.max_scale)
                self._good = 0

    def begin_step(self) -> float:
        self.poll()
        self.found_inf.zero_()
        return self.scale

    def end_step(self):
        """After the optimiser step was queued: ship the flag to the host without waiting for it."""
        if self._event is None:                      # one verdict in flight at a time
            with torch.cuda.device(self.device):   # the copy and the event belong to the scaler's device, whatever is current
                self._host.copy_(self.found_inf, non_blocking=True)
                self._event = torch.cuda.Event()
                self._event.record(torch.cuda.current_stream(self.device))

    def state_dict(self):
        return {"scale": self.scale, "good_steps": self._good}

    def load_state_dict(self, sd):
        self.scale, self._good = float(sd["scale"]), int(sd.get("good_steps", 0))


class CvxComm:
    """An RCCL communicator owned by the engine library (include/cvx_engine.h: cvx_comm_*): the gradient exchange then runs entirely
    behind the C ABI (``cvx_engine_backward_exchange``: no Python between the op ranges of the backward pass).  The 128-byte unique id
    is created on rank 0 and shipped through torch.distributed (any backend -- gloo is enough: it only carries these bytes)."""

    def __init__(self, device: torch.device, process_group=None, rank: Optional[int] = None, world: Optional[int] = None, unique_id: Optional[bytes] = None):
        import torch.distributed as dist
        lib = L.load()
        have_pg = dist.is_available() and dist.is_initialized()
        self.rank = rank if rank is not None else (dist.get_rank(process_group) if have_pg else 0)
        self.world = world if world is not None else (dist.get_world_size(process_group) if have_pg else 1)
        if unique_id is None:
            buf = C.create_string_buffer(128)
            if self.rank == 0:
                L.check(lib.cvx_comm_unique_id(buf), "cvx_comm_unique_id")
            box = [bytes(buf.raw)]
            if self.world > 1:
                dist.broadcast_object_list(box, src=0, group=process_group)
            unique_id = box[0]
        self.handle = C.c_void_p()
        dev_index = device.index if device.index is not None else torch.cuda.current_device()
        L.check(lib.cvx_comm_create(C.byref(self.handle), C.c_char_p(unique_id), self.rank, self.world, dev_index), "cvx_comm_create")
        self.stream = None                # the exchange is queued on the engine's weight-gradient stream (Engine.exchange_stream)
        self._lib = lib

    def all_reduce_(self, t: torch.Tensor, stream: Optional[torch.cuda.Stream] = None):
        st = stream or torch.cuda.current_stream(t.device)
        L.check(self._lib.cvx_allreduce_f32(L.ptr(t), t.numel(), self.handle, C.c_void_p(st.cuda_stream)), "cvx_allreduce_f32")
        _note_param_write(t)              # a raw-pointer write torch's version counter does not see (Engine.forward: keep_shadows)
        return t

    def backward_exchange(self, eng, buckets, dpred: torch.Tensor, loss_scale: float):
        """the data-parallel backward pass in one C call (buckets: graph.grad_buckets rows)"""
        flat = (C.c_int64 * (4 * len(buckets)))(*[int(v) for b in buckets for v in b])
        eng._use_current_stream()
        self.stream = eng.exchange_stream()
        L.check(self._lib.cvx_engine_backward_exchange(eng.handle, L.ptr(dpred), float(loss_scale), self.handle, flat, len(buckets),
                                                       C.c_void_p(self.stream.cuda_stream)), "cvx_engine_backward_exchange")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.cvx_comm_destroy(self.handle)
        except Exception:
            pass


class FusedTrainStep:
    """One optimisation step = forward, loss(+grad), backward, optional DP all-reduce, Adam.

    With ``world_size > 1`` (``torch.distributed`` initialised, backend nccl == RCCL) the flat gradient
    arena is averaged across ranks in ``n_buckets`` contiguous slices on the engine's weight-gradient stream, each slice as
    soon as the backward pass has produced it; BN statistics stay per rank (the reference has no SyncBN).
    """

    def __init__(self, model: Yolo8, criterion: V8DetectionLoss, optimizer: FlatAdam, process_group=None, n_buckets: int = 4,
                 scaler: Optional[DynamicLossScale] = None, comm: Optional["CvxComm"] = None):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.comm = comm               # CvxComm: the exchange runs behind the C ABI (cvx_engine_backward_exchange) instead of torch.distributed
        self.scaler = scaler           # None: static loss scale (criterion.loss_scale)
        self.pg = process_group
        self.n_buckets = n_buckets
        # (Launches are eager.  The hipGraph replay of the step was removed in round 5: the multi-stream capture could not run under the
        # package's own default of one hardware queue, measured 13.9 ms under four, and even a single-stream chain replays no faster than it
        # launches -- profiles/r05_graph_main_chain_ab.txt.)
        self.world = 1
        self.distributed = False
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
            self.distributed = True     # a 1-rank group still exercises the RCCL exchange path (bench.py CVX_FORCE_DIST)
        if comm is not None:
            self.world, self.distributed = comm.world, True
        self._pred = None
        self._dpred = None
        self._side = None
        self._buckets, self._buckets_key = None, None
        self.found_inf = None

    def __call__(self, images: torch.Tensor, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        self.optimizer.sync_lr()
        return self._eager(images, batch)

    def _eager(self, images: torch.Tensor, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        m, crit = self.model, self.criterion
        dev = m.flat_params.device
        B, _, H, W = images.shape
        eng = m.engine_for(H, W)
        A, no = eng.graph.anchors, m.layout.no_pad          # rows of the engine's pred buffer (class columns padded to 8)
        if self._pred is None or self._pred.shape[0] != B or self._pred.shape[1] != A:
            self._pred = torch.empty(B, A, no, device=dev)
            self._dpred = torch.zeros(B, A, no, device=dev, dtype=torch.float16)   # the loss never writes the padding
        targets = flatten_targets(batch, dev)
        dynamic = self.scaler is not None
        scale = self.scaler.begin_step() if dynamic else crit.loss_scale
        pred = m._run_forward(images, training=True, pred=self._pred)
        items, dpred = crit.op(pred, targets, m.level_shapes(H, W), STRIDES, scale, self._dpred)
        if self.distributed and dev.type == "cuda":
            self._backward_overlapped(eng, dpred, scale)
        else:
            eng.backward(dpred, scale)
        if dynamic:                                  # GradScaler.step: skip the update when a gradient is not finite
            check_finite(m.flat_grads, self.scaler.found_inf)
            self.optimizer.found_inf = self.scaler.found_inf
        self.optimizer.step(zero_grad=True, grad_scale=1.0 / self.world)
        if dynamic:
            self.scaler.end_step()
        return items

    def _backward_overlapped(self, eng, dpred, loss_scale):
        """Backward in `n_buckets` op ranges (head first).  As soon as a range's gradients are final, its slice of the flat
        gradient arena is SUM-all-reduced (RCCL) on the engine's weight-gradient stream while the main stream runs the next range; the mean's
        1/world is folded into the Adam kernel.  BASELINE.json north_star: exchange overlapped with the backward pass."""
        m = self.model
        g = m.flat_grads
        # The exchange is queued on the engine's own weight-gradient stream: fold and all-reduce of a range behind the weight gradients they
        # depend on.  A stream of its own for it (a torch pool stream, beside an engine that then had three) was the process's fifth hardware
        # queue and took the step from 6.5 to 16 ms on the ROCm 7 runtime (1-rank RCCL group, bench.py CVX_FORCE_DIST=1).
        self._side = eng.exchange_stream()
        key = id(eng)
        if self._buckets_key != key:
            self._buckets = eng.grad_buckets(m.layout, self.n_buckets)
            self._buckets_key = key
        if self.comm is not None:        # whole pass in one C call: ranges, slab folds and RCCL all-reduces enqueued without host code between them
            self.comm.backward_exchange(eng, self._buckets, dpred, loss_scale)
            return
        cur = torch.cuda.current_stream(g.device)
        backward_with_overlapped_exchange(eng, self._buckets, g, dpred, loss_scale, self.pg, self._side)
        cur.wait_stream(self._side)


def backward_with_overlapped_exchange(eng, buckets, g: torch.Tensor, dpred, loss_scale: float, group=None, side_stream=None):
    """The bucket loop of the data-parallel step, engine-agnostic: ``eng`` provides the segmented backward of
    include/cvx_engine.h (``backward_begin / backward_range(op_hi, op_lo) / grads_ready(op_hi, op_lo, stream) / backward_end``),
    ``buckets`` = ``graph.grad_buckets`` = op ranges in backward order with their contiguous slice [p0, p1) of the flat
    gradient arena ``g``.  Each slice is SUM-all-reduced as soon as its range has run -- on ``side_stream`` (HIP: RCCL,
    overlapping the next range on the current stream) or inline (CPU tensors: the gloo tests drive exactly this loop with a
    fake engine).  The caller folds the mean's 1/world into the optimiser step."""
    import torch.distributed as dist
    eng.backward_begin(dpred, loss_scale)
    for op_hi, op_lo, p0, p1 in buckets:
        eng.backward_range(op_hi, op_lo)
        eng.grads_ready(op_hi, op_lo, side_stream)               # side stream: waits for the range, folds its slabs
        if side_stream is not None:
            with torch.cuda.stream(side_stream):
                dist.all_reduce(g[p0:p1], group=group)            # ordered after the fold on the same stream
        else:
            dist.all_reduce(g[p0:p1], group=group)
    eng.backward_end()


class OverlappedExchange:
    """The data-parallel backward of the engine-backed train steps other than YOLOv8's: buckets from the graph's own parameter offsets
    (``graph.generic_grad_buckets``), each bucket's slice of the flat gradient arena SUM-all-reduced (RCCL) on the engine's weight-gradient stream
    as soon as its op range has run, while the main stream runs the next range; the caller folds 1/world into the optimiser step."""

    def __init__(self, process_group=None, n_buckets: int = 4):
        self.pg, self.n_buckets = process_group, n_buckets
        self._side = None
        self._buckets, self._key = None, None

    def backward(self, eng, flat_grads: torch.Tensor, dpred: torch.Tensor, loss_scale: float):
        from .graph import generic_grad_buckets
        self._side = eng.exchange_stream()          # the engine's weight-gradient stream: see FusedTrainStep._backward_overlapped
        if self._key != id(eng):
            self._buckets, self._key = generic_grad_buckets(eng.graph, self.n_buckets), id(eng)
        cur = torch.cuda.current_stream(flat_grads.device)
        backward_with_overlapped_exchange(eng, self._buckets, flat_grads, dpred, loss_scale, self.pg, self._side)
        cur.wait_stream(self._side)


def broadcast_bn_statistics(model: Yolo8, group=None, src: int = 0):
    """BatchNorm running statistics are per rank (the reference has no SyncBN) and drift apart; before a checkpoint is
    written every rank takes rank ``src``'s, so that a resumed job starts from ONE consistent model."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(model.flat_stats, src=src, group=group)
        dist.broadcast(model._flat["nbt"], src=src, group=group)


def bucket_bounds(n: int, n_buckets: int):
    """Contiguous [start, end) slices of a flat arena, 16-byte aligned starts."""
    per = (n + n_buckets - 1) // max(n_buckets, 1)
    per = max((per + 3) & ~3, 4)
    return [(s, min(n, s + per)) for s in range(0, n, per)]


def allreduce_mean_flat(g: torch.Tensor, world: int, group=None, n_buckets: int = 4, side_stream=None, average: bool = True):
    """Average the flat gradient arena over the ranks: a few large contiguous all-reduces (xGMI is
    point-to-point, so few big messages beat many small ones), issued on a side HIP stream."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return
    bounds = bucket_bounds(g.numel(), n_buckets)
    if not g.is_cuda or side_stream is None:                      # CPU tensors: gloo tests of the bucketing logic
        for s, e in bounds:
            dist.all_reduce(g[s:e], group=group)
            if average:
                g[s:e].div_(world)
        return
    cur = torch.cuda.current_stream(g.device)
    side_stream.wait_stream(cur)
    with torch.cuda.stream(side_stream):
        for s, e in bounds:
            chunk = g[s:e]
            dist.all_reduce(chunk, group=group)
            if average:
                chunk.div_(world)
    cur.wait_stream(side_stream)
