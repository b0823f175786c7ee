"""Seeded synthetic inputs shared by bench.py, the tests and oracle/make_golden.py.

Data generators only -- no algorithm of the hot path lives here.
"""
from __future__ import annotations

import numpy as np
import torch


def images(b: int, h: int, w: int, seed: int = 1) -> torch.Tensor:
    """U[0,1) fp32 images, the range ``TF.to_tensor`` produces (detection_dataset.py:101)."""
    return torch.rand(b, 3, h, w, generator=torch.Generator().manual_seed(seed))


def targets(b: int, seed: int = 2, boxes_per_img: int = 3, nc: int = 80) -> dict:
    """Labels in the yolo8_collate dict format (core/data/collate.py:25-29): 3 boxes per image,
    centre U(0.25,0.75), size U(0.1,0.4), normalised cxcywh."""
    g = torch.Generator().manual_seed(seed)
    n = b * boxes_per_img
    cls = torch.randint(0, nc, (n, 1), generator=g).float()
    cxcy = torch.rand(n, 2, generator=g) * 0.5 + 0.25
    wh = torch.rand(n, 2, generator=g) * 0.3 + 0.1
    return {"batch_idx": torch.arange(b).repeat_interleave(boxes_per_img).float(), "cls": cls,
            "bboxes": torch.cat((cxcy, wh), 1)}


def nms_pred(seed: int = 7, b: int = 2, a: int = 8400, nc: int = 80, hot: int = 900) -> np.ndarray:
    """(B, 4+nc, A) eval-head style predictions with clustered boxes and deliberately tied scores
    (a random-init network never clears conf 0.25, so the NMS tail gets its own generator)."""
    rng = np.random.default_rng(seed)
    pred = np.zeros((b, 4 + nc, a), np.float32)
    centers = rng.uniform(40, 600, size=(b, 40, 2))
    for i in range(b):
        which = rng.integers(0, 40, a)
        pred[i, 0:2] = (centers[i, which] + rng.normal(0, 6, (a, 2))).T
        pred[i, 2:4] = rng.uniform(20, 200, (2, a))
        pred[i, 4:] = rng.uniform(0, 0.2, (nc, a))
        idx = rng.choice(a, hot, replace=False)
        pred[i, 4 + rng.integers(0, 6, hot), idx] = np.round(rng.uniform(0.2, 0.95, hot), 2)
    return pred


def nms_pred_borderline(seed: int = 3, a: int = 2048, nc: int = 80, pairs: int = 600, iou: float = 0.7) -> np.ndarray:
    """(1, 4+nc, A) predictions made of box PAIRS whose IoU sits within ~1e-6 of ``iou`` (the second box is the first one
    shifted along x by w * (1 - iou) / (1 + iou), plus sub-pixel jitter), all in high class indices: the case where
    torchvision's coordinate-offset ``batched_nms`` (boxes re-rounded after adding cls * (max + 1)) and its per-class
    variant can disagree."""
    rng = np.random.default_rng(seed)
    pred = np.zeros((1, 4 + nc, a), np.float32)
    pred[0, 2:4] = 1.0
    pred[0, 4:] = rng.uniform(0, 0.05, (nc, a))
    slots = rng.choice(a // 2, pairs, replace=False) * 2
    for k, s0 in enumerate(slots):
        w, h = rng.uniform(40, 160, 2)
        cx, cy = rng.uniform(100, 540, 2)
        dx = w * (1 - iou) / (1 + iou) + rng.uniform(-2e-4, 2e-4)
        c = int(rng.integers(nc // 2, nc))
        s = float(np.round(rng.uniform(0.3, 0.9), 3))
        for j, (x, sc) in enumerate(((cx, s), (cx + dx, s - 0.01))):
            pred[0, 0:4, s0 + j] = (x, cy, w, h)
            pred[0, 4 + c, s0 + j] = sc
    return pred
