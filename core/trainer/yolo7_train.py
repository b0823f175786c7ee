"""``Yolo7Trainer`` -- registered as ``trainer_yolo7`` like the reference's (core/trainer/yolo7_train.py).  ``train_loop`` keeps the
reference's step semantics (zero_grad -> forward -> Yolo7Loss(preds, targets, images) -> backward -> Adam under AMP, :79-97) and runs it
as the engine's fused step (``Yolo7TrainStep``: engine forward, ``cvx_yolo7_loss`` -- candidate generation, SimOTA assignment and the loss
terms on the device --, engine backward, fused Adam with GradScaler's skip-on-overflow); with ``torch.distributed`` initialised the step
also sums the gradients over the ranks (RCCL).  The dataset readers / mosaic augmentation are outside the hot path: a dataloader yielding
``(images, targets (N, 6) [image, class, cx, cy, w, h])`` (yolo7_collate's format) is injected, or seeded synthetic batches stand in."""
from typing import Dict, List

import torch

from computervision.pytorch_amd.train import DynamicLossScale, FlatAdam
from computervision.pytorch_amd.yolov7 import Yolo7TrainStep
from configs import Yolo7Config
from core.algorithms.yolo_v7 import YOLOv7
from core.trainer.base import BaseTrainer, LinearWarmup
from registry import trainer_registry


class SyntheticYolo7Loader:
    """Seeded stand-in for DetectionDataset + yolo7_collate (core/data/collate.py:5-14): images (B,3,H,W) in [0,1) and targets (N, 6)
    [image index, class, cx, cy, w, h] normalised, grouped by image."""

    def __init__(self, batch_size, hw, num_classes, boxes_per_img=4, length=16, seed=1):
        self.b, self.hw, self.nc, self.k, self.length, self.seed = batch_size, hw, num_classes, boxes_per_img, length, seed

    def __len__(self):
        return self.length

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.length):
            images = torch.rand(self.b, 3, *self.hw, generator=g)
            n = self.b * self.k
            t = torch.zeros(n, 6)
            t[:, 0] = torch.arange(self.b).repeat_interleave(self.k).float()
            t[:, 1] = torch.randint(0, self.nc, (n,), generator=g).float()
            t[:, 2:4] = torch.rand(n, 2, generator=g) * 0.7 + 0.15
            t[:, 4:6] = torch.rand(n, 2, generator=g) * 0.4 + 0.05
            yield images, t


def get_optimizer(optimizer_name, model, initial_lr):
    """reference core/trainer/lr_scheduler.py:37-43 (Adam only)."""
    if optimizer_name.lower() == "adam":
        return FlatAdam(model, lr=initial_lr)
    raise ValueError(f"{optimizer_name} is not supported")


@trainer_registry("yolo7")
class Yolo7Trainer(BaseTrainer):
    def __init__(self, cfg: Yolo7Config, device, dataloader=None):
        self._injected_loader = dataloader
        super().__init__(cfg, device, True)
        self.metric_names = ["loss", "box_loss", "obj_loss", "cls_loss"]
        self.show_option = [True, True, True, True]

    def set_model_algorithm(self):
        self.model_algorithm = YOLOv7(self.cfg, self.device)

    def initialize_model(self):
        self.model, self.model_name = self.model_algorithm.build_model()
        self.model.to(device=self.device)

    def load_data(self):
        loader = self._injected_loader or SyntheticYolo7Loader(self.batch_size, self.input_image_size[1:], self.cfg.dataset.num_classes)
        self.train_dataloader = self.val_dataloader = loader

    def set_optimizer(self):
        self.optimizer = get_optimizer(self.optimizer_name, self.model, self.initial_lr)

    def set_lr_scheduler(self):
        milestones = list(self.milestones) or [int(1e8), int(1e8) + 1]
        self.lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=milestones, gamma=self.gamma,
                                                                 last_epoch=self.last_iter if self.last_iter > 0 else -1)
        if self.warmup_iters > 0:
            self.warmup_scheduler = LinearWarmup(self.optimizer, warmup_period=self.warmup_iters,
                                                 last_step=self.last_iter if self.last_iter > 0 else -1)

    def set_criterion(self):
        self.criterion = self.model_algorithm.build_loss()
        scaler = DynamicLossScale(self.device, init_scale=self.model.loss_scale) if self.mixed_precision else None
        self._step = Yolo7TrainStep(self.model, self.criterion, self.optimizer, scaler=scaler)

    def train_loop(self, batch_data, scaler) -> List:
        images = batch_data[0].to(self.device, non_blocking=True)
        targets = batch_data[1].to(self.device, non_blocking=True)
        items = self._step(images, targets)
        return [items[0], items[1], items[2], items[3]]

    def evaluate_loop(self) -> Dict:
        self.model.eval()
        total, n = 0.0, 0
        with torch.no_grad():
            for images, targets in self.val_dataloader:
                images = images.to(self.device)
                preds = self.model(images)
                total += float(self.criterion(preds, targets.to(self.device), images)[0])
                n += 1
        return {"val_loss": total / max(n, 1)}
