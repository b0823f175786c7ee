"""Trainer template -- the hook API of the reference's ``BaseTrainer`` (core/trainer/base.py:48-163):
``set_model_algorithm, load_data, initialize_model, set_optimizer, set_lr_scheduler, set_criterion,
train_loop, evaluate_loop`` and ``train()`` with the reference's schedule and checkpoint semantics:

* milestones are given in epochs and converted to ITERATIONS with ``(m + 1) * len(train_dataloader)``, ``last_iter =
  (last_epoch + 1) * len(train_dataloader)`` (reference base.py:121-122);
* the LR scheduler is stepped once per iteration and ONLY inside the warm-up dampening context, i.e. never when
  ``warmup_iters == 0`` (reference base.py:261-263 -- a quirk of the reference that is kept: a config without warm-up
  trains at constant LR there too);
* a checkpoint ``{"model", "optimizer", "scheduler", "warm_up"}`` every ``save_interval`` epochs and at the last epoch, a
  bare ``state_dict`` at the end (reference base.py:277-292, core/utils/ckpt.py:38-51); ``cfg.train.resume_training`` +
  ``last_epoch`` resume from one (base.py:180-191).

Dataset readers and TensorBoard are outside the hot path; ``load_data`` may be
overridden or a dataloader injected, and defaults to seeded synthetic batches so the loop is runnable anywhere.
"""
import logging
import os
import time
from contextlib import contextmanager
from typing import Dict, List

import torch

from core.utils.ckpt import CheckPoint


class MeanMetric:
    def __init__(self):
        self.total, self.count = 0.0, 0

    def update(self, v):
        self.total += float(v)
        self.count += 1

    def result(self):
        return self.total / max(self.count, 1)

    def reset(self):
        self.total, self.count = 0.0, 0


class LinearWarmup:
    """The reference's warm-up (core/trainer/warm_up.py:31-120, pytorch_warmup's LinearWarmup): after every scheduler step
    the group learning rates are multiplied by ``min(1, (step + 1) / warmup_period)``; ``dampening()`` restores the
    undamped rates around the scheduler's own step."""

    def __init__(self, optimizer, warmup_period: int, last_step: int = -1):
        self.optimizer, self.warmup_period, self.last_step = optimizer, int(warmup_period), last_step
        self.lrs = [g["lr"] for g in optimizer.param_groups]
        self.dampen()

    def warmup_factor(self, step):
        return min(1.0, (step + 1) / self.warmup_period)

    def dampen(self, step=None):
        step = self.last_step + 1 if step is None else step
        self.last_step = step
        for g in self.optimizer.param_groups:
            g["lr"] *= self.warmup_factor(step)

    @contextmanager
    def dampening(self):
        for g, lr in zip(self.optimizer.param_groups, self.lrs):
            g["lr"] = lr
        yield
        self.lrs = [g["lr"] for g in self.optimizer.param_groups]
        self.dampen()

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


class BaseTrainer:
    def __init__(self, cfg, device, use_iter_milestones=True):
        self.cfg, self.device = cfg, device
        self.dataset_name = cfg.dataset.dataset_name
        self.input_image_size = cfg.arch.input_size
        self.last_epoch = cfg.train.last_epoch
        self.start_epoch = cfg.train.last_epoch + 1
        self.total_epoch = cfg.train.epoch
        self.batch_size = cfg.train.batch_size
        self.initial_lr = cfg.train.initial_lr
        self.warmup_iters = cfg.train.warmup_iters
        self.milestones, self.gamma = list(cfg.train.milestones), cfg.train.gamma
        self.mixed_precision = cfg.train.mixed_precision
        self.num_workers = cfg.train.num_workers
        self.optimizer_name = cfg.optimizer.name
        self.print_interval = cfg.log.print_interval
        self.save_interval = max(int(getattr(cfg.train, "save_interval", 0) or 0), 0)
        self.eval_interval = int(getattr(cfg.train, "eval_interval", 0) or 0)
        self.save_path = getattr(cfg.train, "save_path", "saves")
        self.resume_training_weights = getattr(cfg.train, "resume_training", "") or None
        self.metric_names: List[str] = []
        self.train_dataloader = None
        self.val_dataloader = None
        self.model = None
        self.model_name = None
        self.optimizer = None
        self.lr_scheduler = None
        self.warmup_scheduler = None
        self.logger = logging.getLogger(type(self).__name__)
        self.set_model_algorithm()
        self.load_data()
        n_iter = max(len(self.train_dataloader), 1)
        self.last_iter = (self.last_epoch + 1) * n_iter                       # reference base.py:121
        if use_iter_milestones:
            self.milestones = [(m + 1) * n_iter for m in self.milestones]      # reference base.py:122
        self.initialize_model()
        self.set_optimizer()
        self.set_lr_scheduler()
        self.set_criterion()

    # hooks ---------------------------------------------------------------------------------------
    def set_model_algorithm(self):
        raise NotImplementedError

    def load_data(self):
        raise NotImplementedError

    def initialize_model(self):
        raise NotImplementedError

    def set_optimizer(self):
        raise NotImplementedError

    def set_lr_scheduler(self):
        self.lr_scheduler = None

    def set_criterion(self):
        raise NotImplementedError

    def train_loop(self, batch_data, scaler) -> List:
        raise NotImplementedError

    def evaluate_loop(self) -> Dict:
        return {}

    # checkpoints ---------------------------------------------------------------------------------
    def _is_rank0(self):
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0

    def save_checkpoint(self, path, full=True):
        """Rank 0 writes; under data parallelism rank 0's BatchNorm running statistics are broadcast first, so that every
        rank continues from (and a resumed job starts from) one consistent model."""
        from computervision.pytorch_amd.train import broadcast_bn_statistics
        broadcast_bn_statistics(self.model)
        if not self._is_rank0():
            return
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        if full:
            CheckPoint.save(self.model, path, optimizer=self.optimizer, scheduler=self.lr_scheduler, warm_up=self.warmup_scheduler)
        else:
            CheckPoint.save(self.model, path)

    # driver --------------------------------------------------------------------------------------
    def train(self, max_iters=None):
        metrics = [MeanMetric() for _ in self.metric_names]
        if CheckPoint.check(self.resume_training_weights):                     # reference base.py:180-191
            CheckPoint.load(self.resume_training_weights, self.device, self.model, pure=False, optimizer=self.optimizer,
                            scheduler=self.lr_scheduler, warm_up=self.warmup_scheduler)
            self.logger.info("resumed from %s at epoch %d", self.resume_training_weights, self.last_epoch)
        it = 0
        tag = f"{self.model_name}_{str(self.dataset_name).lower()}"
        for epoch in range(self.start_epoch, self.total_epoch):
            self.model.train()
            for m in metrics:
                m.reset()
            t0 = time.time()
            for batch in self.train_dataloader:
                values = self.train_loop(batch, None)
                it += 1
                if self.warmup_scheduler is not None:                            # reference base.py:261-263: the scheduler
                    with self.warmup_scheduler.dampening():                      # steps only under warm-up dampening
                        self.lr_scheduler.step()
                if it % self.print_interval == 0 or (max_iters and it >= max_iters):
                    for m, v in zip(metrics, values):      # one host sync per print interval, not per step
                        m.update(v.item() if torch.is_tensor(v) else v)
                    self.logger.info("epoch %d iter %d lr %.3g %s (%.1fs)", epoch, it, self.optimizer.param_groups[0]["lr"],
                                     {n: round(m.result(), 4) for n, m in zip(self.metric_names, metrics)}, time.time() - t0)
                if max_iters and it >= max_iters:
                    return
            if self.eval_interval and epoch % self.eval_interval == 0:
                self.logger.info("eval %s", self.evaluate_loop())
            if self.save_interval and (epoch % self.save_interval == 0 or epoch == self.total_epoch - 1):
                self.save_checkpoint(os.path.join(self.save_path, f"{tag}_epoch-{epoch}.pth"), full=True)
        if self.save_interval:
            self.save_checkpoint(os.path.join(self.save_path, f"{tag}_final.pth"), full=False)
