"""``Yolo8Trainer`` -- registered as ``trainer_yolo8_det`` like the reference's
(core/trainer/yolo8_train.py:19-129).  ``train_loop`` keeps the reference's step semantics
(zero_grad -> forward -> loss -> backward -> Adam over all parameters, :93-111) and runs it as the
engine's fused step; with ``torch.distributed`` initialised the step also averages gradients (RCCL).
"""
from typing import Dict, List

import torch

from computervision.pytorch_amd.train import DynamicLossScale, FlatAdam, FusedTrainStep
from configs import Yolo8DetConfig
from core.algorithms.yolo_v8 import YOLOv8
from core.trainer.base import BaseTrainer, LinearWarmup
from registry import trainer_registry


class SyntheticDetectionLoader:
    """Seeded stand-in for DetectionDataset + yolo8_collate (core/data/collate.py:17-29): yields
    ``(images (B,3,H,W) in [0,1), {"batch_idx","cls","bboxes"})``."""

    def __init__(self, batch_size, hw, num_classes, length=64, boxes_per_img=3, seed=1):
        self.b, self.hw, self.nc, self.length, self.k, self.seed = batch_size, hw, num_classes, length, boxes_per_img, seed

    def __len__(self):
        return self.length

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.length):
            n = self.b * self.k
            images = torch.rand(self.b, 3, *self.hw, generator=g)
            cls = torch.randint(0, self.nc, (n, 1), generator=g).float()
            boxes = torch.cat((torch.rand(n, 2, generator=g) * 0.5 + 0.25, torch.rand(n, 2, generator=g) * 0.3 + 0.1), 1)
            yield images, {"batch_idx": torch.arange(self.b).repeat_interleave(self.k).float(), "cls": cls, "bboxes": boxes}


def get_optimizer(optimizer_name, model, initial_lr):
    """reference core/trainer/lr_scheduler.py:37-43 (Adam only)."""
    if optimizer_name.lower() == "adam":
        return FlatAdam(model, lr=initial_lr)
    raise ValueError(f"{optimizer_name} is not supported")


@trainer_registry("yolo8_det")
class Yolo8Trainer(BaseTrainer):
    def __init__(self, cfg: Yolo8DetConfig, device, dataloader=None):
        self._injected_loader = dataloader
        self.metric_names = ["loss"]
        self.show_option = [True]
        super().__init__(cfg, device, True)
        self.metric_names = ["loss"]

    def set_model_algorithm(self):
        self.model_algorithm = YOLOv8(self.cfg, self.device)

    def initialize_model(self):
        self.model, self.model_name = self.model_algorithm.build_model()
        self.model.to(device=self.device)

    def load_data(self):
        loader = self._injected_loader or SyntheticDetectionLoader(self.batch_size, self.input_image_size[1:],
                                                                   self.cfg.dataset.num_classes)
        self.train_dataloader = self.val_dataloader = loader

    def set_optimizer(self):
        self.optimizer = get_optimizer(self.optimizer_name, self.model, self.initial_lr)

    def set_lr_scheduler(self):
        """EnhancedMultiStepLR over ITERATION milestones + LinearWarmup (reference yolo8_train.py:76-88,
        lr_scheduler.py:87-91: an empty milestone list means 'never')."""
        milestones = list(self.milestones) or [int(1e8), int(1e8) + 1]
        self.lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=milestones, gamma=self.gamma,
                                                                 last_epoch=self.last_iter if self.last_iter > 0 else -1)
        if self.warmup_iters > 0:
            self.warmup_scheduler = LinearWarmup(self.optimizer, warmup_period=self.warmup_iters,
                                                 last_step=self.last_iter if self.last_iter > 0 else -1)

    def set_criterion(self):
        self.criterion = self.model_algorithm.build_loss(model=self.model)
        eng_cfg = self.cfg.engine
        scaler = None
        if self.mixed_precision and getattr(eng_cfg, "dynamic_loss_scale", False) and not getattr(eng_cfg, "graph_capture", False):
            scaler = DynamicLossScale(self.device, init_scale=getattr(eng_cfg, "init_loss_scale", 65536.0))
        self._step = FusedTrainStep(self.model, self.criterion, self.optimizer, n_buckets=getattr(eng_cfg, "allreduce_buckets", 4),
                                    scaler=scaler)

    def train_loop(self, batch_data, scaler) -> List:
        images = batch_data[0].to(self.device, non_blocking=True)
        items = self._step(images, batch_data[1])
        return [items.sum() * images.shape[0]]          # the reference's scalar: sum(box,cls,dfl) * batch

    def evaluate_loop(self) -> Dict:
        self.model.eval()
        total, n = 0.0, 0
        with torch.no_grad():
            for images, targets in self.val_dataloader:
                preds = self.model(images.to(self.device))
                loss, _ = self.criterion(preds, targets)
                total += float(loss)
                n += 1
        return {"val_loss": total / max(n, 1)}
