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
