"""``DeeplabV3PlusTrainer`` -- registered as ``trainer_deeplabv3plus`` like the reference's
(core/trainer/segmentation_trainer.py:21-159).  ``train_loop`` keeps the reference's step semantics (zero_grad -> forward ->
criterion -> backward -> Adam under AMP, :114-131) and runs it as the engine's fused step (``SegTrainStep``: engine forward,
``cvx_seg_loss``, engine backward, fused Adam with GradScaler's skip-on-overflow); with ``torch.distributed`` initialised the step
also sums the gradients over the ranks (RCCL).  ``evaluate_loop`` reports the reference's numbers (loss, Overall / Mean / FreqW
accuracy, Mean IoU, :133-159) from a confusion matrix accumulated on the device.  The VOC / Cityscapes / SBD readers are outside
the hot path: a dataloader is injected, or seeded synthetic batches stand in.
"""
from typing import Dict, List

import torch

from computervision.pytorch_amd.deeplab import SegTrainStep
from computervision.pytorch_amd.train import DynamicLossScale, FlatAdam
from configs import DeeplabV3PlusConfig
from core.algorithms.segmentation_2d import DeeplabV3PlusA
from core.trainer.base import BaseTrainer, LinearWarmup
from registry import trainer_registry


class SyntheticSegmentationLoader:
    """Seeded stand-in for get_voc_dataloader (core/data/segmentation_dataset.py): yields ``(images (B,3,H,W) in [0,1),
    targets (B,H,W) int64 in [0, num_classes) with blocky regions and a few ignored pixels)``."""

    def __init__(self, batch_size, hw, num_classes, length=32, seed=1, ignore_index=-100):
        self.b, self.hw, self.nc, self.length, self.seed, self.ignore = batch_size, hw, num_classes, length, seed, ignore_index

    def __len__(self):
        return self.length

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        h, w = self.hw
        for _ in range(self.length):
            images = torch.rand(self.b, 3, h, w, generator=g)
            coarse = torch.randint(0, self.nc, (self.b, 1, (h + 31) // 32, (w + 31) // 32), generator=g).float()
            targets = torch.nn.functional.interpolate(coarse, size=(h, w), mode="nearest")[:, 0].long()
            targets[torch.rand(self.b, h, w, generator=g) < 0.02] = self.ignore
            yield images, targets


class SegmentationMetrics:
    """core/metrics/seg_metrics.py:4-44 with the confusion matrix kept on the device (one bincount per batch)."""

    def __init__(self, num_classes, device="cpu"):
        self.num_classes = num_classes
        self.confusion_matrix = torch.zeros(num_classes, num_classes, dtype=torch.float64, device=device)

    def reset(self):
        self.confusion_matrix.zero_()

    def add_batch(self, predictions, gts):
        predictions, gts = torch.as_tensor(predictions).reshape(-1), torch.as_tensor(gts).reshape(-1)
        mask = (gts >= 0) & (gts < self.num_classes)
        idx = (self.num_classes * gts[mask].long() + predictions[mask].long()).to(self.confusion_matrix.device)
        self.confusion_matrix += torch.bincount(idx, minlength=self.num_classes ** 2).reshape(self.num_classes, self.num_classes).double()

    def get_results(self):
        hist = self.confusion_matrix.cpu()
        diag = torch.diag(hist)
        acc = float(diag.sum() / hist.sum())
        acc_cls = diag / hist.sum(1)
        iu = diag / (hist.sum(1) + hist.sum(0) - diag)
        freq = hist.sum(1) / hist.sum()
        return {"Overall Acc": acc, "Mean Acc": float(torch.nanmean(acc_cls)), "FreqW Acc": float((freq[freq > 0] * iu[freq > 0]).sum()),
                "Mean IoU": float(torch.nanmean(iu)), "Class IoU": dict(zip(range(self.num_classes), iu.tolist()))}


def get_optimizer(optimizer_name, model, initial_lr):
    """reference core/trainer/lr_scheduler.py:37-43 (Adam only)."""
    if optimizer_name.lower() == "adam":
        return FlatAdam(model, lr=initial_lr)
    raise ValueError(f"{optimizer_name} is not supported")


@trainer_registry("deeplabv3plus")
class DeeplabV3PlusTrainer(BaseTrainer):
    def __init__(self, cfg: DeeplabV3PlusConfig, device, dataloader=None):
        self._injected_loader = dataloader
        super().__init__(cfg, device, False)
        self.metrics = SegmentationMetrics(num_classes=cfg.dataset.num_classes, device=device)
        self.metric_names = ["loss"]
        self.show_option = [True]

    def set_model_algorithm(self):
        self.model_algorithm = DeeplabV3PlusA(self.cfg, self.device)

    def initialize_model(self):
        self.model, self.model_name = self.model_algorithm.build_model()
        self.model.to(device=self.device)

    def load_data(self):
        loader = self._injected_loader or SyntheticSegmentationLoader(self.batch_size, self.cfg.arch.crop_size, self.cfg.dataset.num_classes)
        self.train_dataloader = self.val_dataloader = loader

    def set_optimizer(self):
        self.optimizer = get_optimizer(self.optimizer_name, self.model, self.initial_lr)

    def set_lr_scheduler(self):
        """EnhancedMultiStepLR + LinearWarmup (reference :98-111; an empty milestone list means 'never', lr_scheduler.py:87-91)."""
        milestones = list(self.milestones) or [int(1e8), int(1e8) + 1]
        self.lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=milestones, gamma=self.gamma,
                                                                 last_epoch=self.last_iter if self.last_iter > 0 else -1)
        if self.warmup_iters > 0:
            self.warmup_scheduler = LinearWarmup(self.optimizer, warmup_period=self.warmup_iters,
                                                 last_step=self.last_iter if self.last_iter > 0 else -1)

    def set_criterion(self):
        # dropout masks (aspp.project.3, deeplabv3plus.py:67) are a counter-based hash of (seed, training pass, op, element) on the engine:
        # derive the seed from torch's global seed, the rank (the reference's ranks draw independent masks) and the iteration a resumed run
        # starts at (a resumed job must not replay the mask sequence from pass 0)
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.model.seed = (int(torch.initial_seed()) * 0x9E3779B97F4A7C15 + rank * 0xBF58476D1CE4E5B9 + int(self.last_iter) * 0x94D049BB133111EB) & (2 ** 64 - 1)
        self.criterion = self.model_algorithm.build_loss()
        scaler = DynamicLossScale(self.device, init_scale=self.model.loss_scale) if self.mixed_precision else None   # GradScaler()
        self._step = SegTrainStep(self.model, self.criterion, self.optimizer, scaler=scaler)

    def train_loop(self, batch_data, scaler) -> List:
        images = batch_data[0].to(self.device, non_blocking=True)
        targets = batch_data[1].to(self.device, non_blocking=True)
        return [self._step(images, targets)]

    def evaluate_loop(self) -> Dict:
        self.model.eval()
        self.metrics.reset()
        total, n = 0.0, 0
        with torch.no_grad():
            for images, targets in self.val_dataloader:
                images, targets = images.to(self.device), targets.to(self.device)
                preds = self.model(images)
                total += float(self.criterion(preds, targets))
                self.metrics.add_batch(torch.argmax(preds, dim=1), targets)
                n += 1
        r = self.metrics.get_results()
        return {"Loss": total / max(n, 1), "Overall Acc": r["Overall Acc"], "Mean Acc": r["Mean Acc"], "FreqW Acc": r["FreqW Acc"],
                "Mean IoU": r["Mean IoU"]}
