"""``SsdTrainer`` -- registered as ``trainer_ssd`` like the reference's (core/trainer/ssd_train.py).  ``train_loop`` keeps the reference's
step semantics (zero_grad -> forward -> MultiBoxLossV2 -> backward -> Adam under AMP, :96-115) and runs it as the engine's fused step
(``SsdTrainStep``: engine forward, ``cvx_multibox_loss``, engine backward, fused Adam with GradScaler's skip-on-overflow); with
``torch.distributed`` initialised the step also sums the gradients over the ranks (RCCL).  The dataset readers and ``ssd_collate``'s CPU
prior matching / target encoding (core/data/collate.py:32-49, core/algorithms/ssd.py:327-480) are outside the hot path: a dataloader
yielding ``(images, y_true (B, 8732, 4 + (nc + 1) + 1))`` is injected, or seeded synthetic batches of that format stand in."""
from typing import Dict, List

import torch

from computervision.pytorch_amd.ssd import SsdTrainStep
from computervision.pytorch_amd.train import DynamicLossScale, FlatAdam
from configs import SsdConfig
from core.algorithms.ssd import Ssd
from core.trainer.base import BaseTrainer, LinearWarmup
from registry import trainer_registry


class SyntheticSsdLoader:
    """Seeded stand-in for DetectionDataset + ssd_collate: images (B,3,300,300) in [0,1) and encoded targets (B, 8732, 4 + (nc+1) + 1):
    a few positive priors per image with box regression targets and a one-hot class, background one-hot elsewhere."""

    def __init__(self, batch_size, hw, num_classes, anchors=8732, n_pos=24, length=16, seed=1):
        self.b, self.hw, self.nc, self.a, self.n_pos, self.length, self.seed = batch_size, hw, num_classes, anchors, n_pos, length, seed

    def __len__(self):
        return self.length

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.length):
            images = torch.rand(self.b, 3, *self.hw, generator=g)
            y = torch.zeros(self.b, self.a, 4 + self.nc + 1 + 1)
            y[:, :, 4] = 1.0
            for b in range(self.b):
                idx = torch.randperm(self.a, generator=g)[:self.n_pos]
                y[b, idx, :4] = torch.randn(self.n_pos, 4, generator=g)
                y[b, idx, 4] = 0.0
                y[b, idx, 5 + torch.randint(0, self.nc, (self.n_pos,), generator=g)] = 1.0
                y[b, idx, -1] = 1.0
            yield images, y


def get_optimizer(optimizer_name, model, initial_lr):
    """reference core/trainer/lr_scheduler.py:37-43 (Adam only)."""
    if optimizer_name.lower() == "adam":
        return FlatAdam(model, lr=initial_lr)
    raise ValueError(f"{optimizer_name} is not supported")


@trainer_registry("ssd")
class SsdTrainer(BaseTrainer):
    def __init__(self, cfg: SsdConfig, device, dataloader=None):
        self._injected_loader = dataloader
        super().__init__(cfg, device, True)
        self.metric_names = ["loss", "loc_loss", "conf_loss"]
        self.show_option = [True, True, True]

    def set_model_algorithm(self):
        self.model_algorithm = Ssd(self.cfg, self.device)

    def initialize_model(self):
        self.model, self.model_name = self.model_algorithm.build_model()
        self.model.to(device=self.device)

    def load_data(self):
        loader = self._injected_loader or SyntheticSsdLoader(self.batch_size, self.input_image_size[1:], self.cfg.dataset.num_classes)
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
        self._step = SsdTrainStep(self.model, self.criterion, self.optimizer, scaler=scaler)

    def train_loop(self, batch_data, scaler) -> List:
        images = batch_data[0].to(self.device, non_blocking=True)
        targets = batch_data[1].to(self.device, non_blocking=True)
        items = self._step(images, targets)
        return [items[0], items[1], items[2]]

    def evaluate_loop(self) -> Dict:
        self.model.eval()
        total, n = 0.0, 0
        with torch.no_grad():
            for images, targets in self.val_dataloader:
                preds = self.model(images.to(self.device))
                total += float(self.criterion(y_true=targets.to(self.device), y_pred=preds)[0])
                n += 1
        return {"val_loss": total / max(n, 1)}
