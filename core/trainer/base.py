"""Trainer template -- the hook API of the reference's ``BaseTrainer`` (core/trainer/base.py:48-163):
``set_model_algorithm, load_data, initialize_model, set_optimizer, set_lr_scheduler, set_criterion,
train_loop, evaluate_loop`` and ``train()``.  Dataset readers, TensorBoard and checkpoint files are
outside the hot path; ``load_data`` may be overridden or a dataloader
injected, and defaults to seeded synthetic batches so the loop is runnable anywhere.
"""
import logging
import time
from typing import Dict, List

import torch


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


class BaseTrainer:
    def __init__(self, cfg, device, use_iter_milestones=True):
        self.cfg, self.device = cfg, device
        self.dataset_name = cfg.dataset.dataset_name
        self.input_image_size = cfg.arch.input_size
        self.start_epoch = cfg.train.last_epoch + 1
        self.total_epoch = cfg.train.epoch
        self.batch_size = cfg.train.batch_size
        self.initial_lr = cfg.train.initial_lr
        self.warmup_iters = cfg.train.warmup_iters
        self.milestones, self.gamma = cfg.train.milestones, cfg.train.gamma
        self.mixed_precision = cfg.train.mixed_precision
        self.num_workers = cfg.train.num_workers
        self.optimizer_name = cfg.optimizer.name
        self.print_interval = cfg.log.print_interval
        self.metric_names: List[str] = []
        self.train_dataloader = None
        self.val_dataloader = None
        self.logger = logging.getLogger(type(self).__name__)
        self.set_model_algorithm()
        self.load_data()
        self.last_iter = (self.start_epoch - 1) * max(len(self.train_dataloader), 1) if self.start_epoch > 0 else -1
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

    # driver --------------------------------------------------------------------------------------
    def train(self, max_iters=None):
        metrics = [MeanMetric() for _ in self.metric_names]
        it = 0
        for epoch in range(self.start_epoch, self.total_epoch):
            self.model.train()
            for m in metrics:
                m.reset()
            t0 = time.time()
            for batch in self.train_dataloader:
                values = self.train_loop(batch, None)
                it += 1
                if it % self.print_interval == 0 or (max_iters and it >= max_iters):
                    for m, v in zip(metrics, values):      # one host sync per print interval, not per step
                        m.update(v.item() if torch.is_tensor(v) else v)
                    self.logger.info("epoch %d iter %d %s (%.1fs)", epoch, it,
                                     {n: round(m.result(), 4) for n, m in zip(self.metric_names, metrics)}, time.time() - t0)
                if max_iters and it >= max_iters:
                    return
            if self.cfg.train.eval_interval and (epoch + 1) % self.cfg.train.eval_interval == 0:
                self.logger.info("eval %s", self.evaluate_loop())
