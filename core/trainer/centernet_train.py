"""``CenterNetTrainer`` -- registered as ``trainer_centernet`` like the reference's (core/trainer/centernet_train.py:21-135).
``train_loop`` keeps the reference's step semantics (zero_grad -> forward -> CombinedLoss -> backward -> Adam under AMP, :104-121) and
runs it as the engine's fused step (``CenterNetTrainStep``: engine forward, ``cvx_centernet_loss``, engine backward, fused Adam with
GradScaler's skip-on-overflow); with ``torch.distributed`` initialised the step also sums the gradients over the ranks (RCCL).
The dataset readers and ``centernet_collate``'s CPU target drawing (core/data/collate.py:52-68, core/algorithms/centernet.py:66-120) are
outside the hot path: a dataloader yielding ``(images, [heatmap, reg, wh, reg_mask, indices])`` is injected, or seeded synthetic
batches of that format stand in."""
from typing import Dict, List

import torch

from computervision.pytorch_amd.dla import CenterNetTrainStep
from computervision.pytorch_amd.train import DynamicLossScale, FlatAdam
from configs import CenternetConfig
from core.algorithms.centernet import CenterNetA
from core.trainer.base import BaseTrainer, LinearWarmup
from registry import trainer_registry


class SyntheticCenterNetLoader:
    """Seeded stand-in for DetectionDataset + centernet_collate: images (B,3,H,W) in [0,1) and the five target tensors in the format of
    CenterNet.generate_targets -- Gaussian bumps with an exact 1 at each centre, sub-pixel offsets, sizes, mask, flat indices."""

    def __init__(self, batch_size, hw, num_classes, max_boxes=30, ratio=4, length=16, seed=1):
        self.b, self.hw, self.nc, self.k, self.ratio, self.length, self.seed = batch_size, hw, num_classes, max_boxes, ratio, length, seed

    def __len__(self):
        return self.length

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        H, W = self.hw
        h, w = H // self.ratio, W // self.ratio
        ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
        for _ in range(self.length):
            images = torch.rand(self.b, 3, H, W, generator=g)
            heat = torch.zeros(self.b, h, w, self.nc)
            reg, wh = torch.zeros(self.b, self.k, 2), torch.zeros(self.b, self.k, 2)
            mask, idx = torch.zeros(self.b, self.k), torch.zeros(self.b, self.k, dtype=torch.long)
            for b in range(self.b):
                for k in range(3):
                    cx, cy = float(torch.rand(1, generator=g)) * (w - 1), float(torch.rand(1, generator=g)) * (h - 1)
                    bw, bh = 2 + float(torch.rand(1, generator=g)) * w / 3, 2 + float(torch.rand(1, generator=g)) * h / 3
                    c, ix, iy = int(torch.randint(0, self.nc, (1,), generator=g)), int(cx), int(cy)
                    sigma = max(1.0, min(bw, bh) / 6)
                    heat[b, :, :, c] = torch.maximum(heat[b, :, :, c], torch.exp(-((xs - ix) ** 2 + (ys - iy) ** 2) / (2 * sigma * sigma)))
                    reg[b, k], wh[b, k] = torch.tensor([cx - ix, cy - iy]), torch.tensor([bw, bh])
                    mask[b, k], idx[b, k] = 1.0, iy * w + ix
            yield images, [heat, reg, wh, mask, idx]


def get_optimizer(optimizer_name, model, initial_lr):
    """reference core/trainer/lr_scheduler.py:37-43 (Adam only)."""
    if optimizer_name.lower() == "adam":
        return FlatAdam(model, lr=initial_lr)
    raise ValueError(f"{optimizer_name} is not supported")


@trainer_registry("centernet")
class CenterNetTrainer(BaseTrainer):
    def __init__(self, cfg: CenternetConfig, device, dataloader=None):
        self._injected_loader = dataloader
        super().__init__(cfg, device, True)
        self.metric_names = ["loss"]
        self.show_option = [True]

    def set_model_algorithm(self):
        self.model_algorithm = CenterNetA(self.cfg, self.device)

    def initialize_model(self):
        self.model, self.model_name = self.model_algorithm.build_model()
        self.model.to(device=self.device)

    def load_data(self):
        loader = self._injected_loader or SyntheticCenterNetLoader(self.batch_size, self.input_image_size[1:], self.cfg.dataset.num_classes,
                                                                   getattr(self.cfg.train, "max_num_boxes", 30), self.cfg.arch.downsampling_ratio)
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
        self._step = CenterNetTrainStep(self.model, self.criterion, self.optimizer, scaler=scaler)

    def train_loop(self, batch_data, scaler) -> List:
        images = batch_data[0].to(self.device, non_blocking=True)
        targets = [t.to(self.device, non_blocking=True) for t in batch_data[1]]
        return [self._step(images, targets)[0]]

    def evaluate_loop(self) -> Dict:
        self.model.eval()
        total, n = 0.0, 0
        with torch.no_grad():
            for images, targets in self.val_dataloader:
                preds = self.model(images.to(self.device))
                total += float(self.criterion(preds, [t.to(self.device) for t in targets]))
                n += 1
        return {"val_loss": total / max(n, 1)}
