"""CPU oracle for the box-decode + NMS tail -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

PARITY UNPINNED UPSTREAM: the suppression arithmetic of the reference lives in
``torchvision.ops.batched_nms`` (call site core/utils/ultralytics_ops.py:247), a third-party
dependency that is neither vendored under /root/reference nor installed in this image (README.md:8
names torchvision 0.14.1; requirements.txt leaves it unpinned) and the reference holds no test or
golden vector at that boundary.  This file therefore *restates* torchvision-0.14.1's published
algorithm (torchvision/ops/boxes.py ``batched_nms`` + csrc/ops/{cpu,cuda}/nms_kernel) and is itself
the bit-exact reference for the HIP NMS indices.  Both strategies of 0.14.1 are restated:

* ``offset``  = ``_batched_nms_coordinate_trick``: every box is shifted by ``cls * (boxes.max() + 1)`` (fp32) and ONE
  class-agnostic greedy pass runs over the shifted boxes -- areas and intersections are computed from the shifted,
  re-rounded coordinates, which can flip a borderline ``IoU > thr``;
* ``vanilla`` = ``_batched_nms_vanilla``: one greedy pass per class on the unshifted boxes, survivors re-sorted by score;
* ``tv0141_cuda`` / ``tv0141_cpu`` = the library's own switch: vanilla when ``boxes.numel()`` exceeds 20000 (CUDA tensors,
  i.e. more than 5000 boxes -- the reference's predict path keeps the tensor on the GPU) or 4000 (CPU), else the
  coordinate trick.  With conf 0.25 an image has a few hundred candidates: 0.14.1 takes the OFFSET path there.

Common to both: candidates sorted by score, descending; ties broken by the lower original anchor index
(``torch.argsort(descending=True)`` at ultralytics_ops.py:240 and the sort inside ``nms`` are unstable, so the reference
leaves tie order undefined -- we define it); a box is suppressed when ``inter / (area_i + area_j - inter) > iou_thres``
evaluated in fp32 in exactly this operation order; kept boxes are returned in descending-score order, truncated to
``max_det``.

Everything around that call is reference Python (ultralytics_ops.py:131-264, 360-375;
core/algorithms/yolo_v8.py:210-242; core/utils/image_process.py:69-97) and is restated line by
line, minus the wall-clock early exit (ultralytics_ops.py:260-262), which is nondeterministic.
"""
from __future__ import annotations

import numpy as np


def xywh2xyxy(b: np.ndarray) -> np.ndarray:
    """ultralytics_ops.py:360-375 (fp32: x -/+ w/2)."""
    b = b.astype(np.float32)
    half = b[..., 2:4] / np.float32(2)
    return np.concatenate((b[..., 0:2] - half, b[..., 0:2] + half), -1)


def _greedy(boxes: np.ndarray, same: "np.ndarray | None", iou_thres: float) -> np.ndarray:
    """torchvision nms kernel on boxes (n,4) xyxy fp32 already in descending-score order -> kept positions (ascending).
    ``same`` (optional, (n,) ints): only boxes with equal entries interact."""
    n = boxes.shape[0]
    x1, y1, x2, y2 = (boxes[:, i].astype(np.float32) for i in range(4))
    area = (x2 - x1) * (y2 - y1)
    thr = np.float32(iou_thres)
    dead = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(i)
        j = np.arange(i + 1, n)
        j = j[~dead[j]] if same is None else j[(~dead[j]) & (same[j] == same[i])]
        if j.size == 0:
            continue
        w = np.maximum(np.float32(0), np.minimum(x2[i], x2[j]) - np.maximum(x1[i], x1[j]))
        h = np.maximum(np.float32(0), np.minimum(y2[i], y2[j]) - np.maximum(y1[i], y1[j]))
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (area[i] + area[j] - inter)
        dead[j[ovr > thr]] = True
    return np.asarray(keep, dtype=np.int64)


def greedy_nms_per_class(boxes: np.ndarray, cls: np.ndarray, iou_thres: float) -> np.ndarray:
    """``_batched_nms_vanilla``: per-class greedy passes on the unshifted boxes (input in descending-score order, so the
    union of the per-class survivors in input order IS the final descending-score order)."""
    return _greedy(boxes, cls, iou_thres)


def greedy_nms_offset(boxes: np.ndarray, cls: np.ndarray, iou_thres: float) -> np.ndarray:
    """``_batched_nms_coordinate_trick``: boxes + cls * (boxes.max() + 1) in fp32, one class-agnostic pass."""
    if boxes.shape[0] == 0:
        return np.zeros((0,), np.int64)
    boxes = boxes.astype(np.float32)
    off = cls.astype(np.float32) * (boxes.max() + np.float32(1))
    return _greedy(boxes + off[:, None], None, iou_thres)


VARIANTS = ("tv0141_cuda", "tv0141_cpu", "offset", "vanilla")


def batched_nms(boxes: np.ndarray, cls: np.ndarray, iou_thres: float, variant: str = "tv0141_cuda") -> np.ndarray:
    """torchvision 0.14.1 ``batched_nms`` on score-sorted boxes -> kept positions, ascending = descending score."""
    assert variant in VARIANTS, variant
    if variant.startswith("tv0141"):
        limit = 20000 if variant.endswith("cuda") else 4000
        variant = "vanilla" if boxes.size > limit else "offset"
    return greedy_nms_per_class(boxes, cls, iou_thres) if variant == "vanilla" else greedy_nms_offset(boxes, cls, iou_thres)


def non_max_suppression(pred: np.ndarray, conf_thres: float = 0.25, iou_thres: float = 0.7,
                        max_det: int = 300, max_nms: int = 30000, variant: str = "tv0141_cuda"):
    """pred (B, 4+nc, A) fp32 [cx,cy,w,h, cls scores] -> per image (rows (k,6), anchor_idx (k,)).

    rows = [x1,y1,x2,y2,conf,cls] as ultralytics_ops.py:225-226,248; anchor_idx = the column of
    ``pred`` each kept row came from (the "NMS indices" the GPU path must match bit-exactly).
    """
    assert 0 <= conf_thres <= 1 and 0 <= iou_thres <= 1
    pred = np.asarray(pred, dtype=np.float32)
    out = []
    for x in pred:                                           # (4+nc, A)
        scores = x[4:]
        cand = np.nonzero(scores.max(0) > np.float32(conf_thres))[0]        # :190,204
        if cand.size == 0:
            out.append((np.zeros((0, 6), np.float32), np.zeros((0,), np.int64)))
            continue
        box = xywh2xyxy(x[:4, cand].T)
        conf = scores[:, cand].max(0)
        cls = scores[:, cand].argmax(0)                      # first max, as torch.max(1)
        sel = conf > np.float32(conf_thres)                  # :226 (redundant with :190, kept)
        box, conf, cls, cand = box[sel], conf[sel], cls[sel], cand[sel]
        order = np.lexsort((cand, -conf.astype(np.float64)))[:max_nms]      # desc score, asc index
        box, conf, cls, cand = box[order], conf[order], cls[order], cand[order]
        keep = batched_nms(box, cls, iou_thres, variant)[:max_det]
        rows = np.concatenate((box[keep], conf[keep, None], cls[keep, None].astype(np.float32)), 1)
        out.append((rows.astype(np.float32), cand[keep].astype(np.int64)))
    return out


def decode_box(rows: np.ndarray, input_hw, image_hw, letterbox: bool = True):
    """YOLOv8.decode_box after NMS (yolo_v8.py:229-242) + reverse_letter_box_numpy
    (image_process.py:69-97): network-input pixels -> original-image pixels.

    The reference's ``astype(np.int)`` (yolo_v8.py:231) does not exist on NumPy >= 1.24; int64 is
    what it meant.
    """
    rows = np.asarray(rows, dtype=np.float32)
    bbox, conf, cls = rows[:, :4].copy(), rows[:, 4], rows[:, 5].astype(np.int64)
    in_h, in_w = input_hw
    img_h, img_w = image_hw
    bbox[:, 0::2] /= in_w
    bbox[:, 1::2] /= in_h
    xy, wh = (bbox[:, 0:2] + bbox[:, 2:4]) / 2, bbox[:, 2:4] - bbox[:, 0:2]
    new = np.concatenate((xy - wh / 2, xy + wh / 2), -1)
    if letterbox:
        new[:, 0::2] *= in_w
        new[:, 1::2] *= in_h
        scale = max(img_h / in_h, img_w / in_w)
        top = (in_h - img_h / scale) // 2
        left = (in_w - img_w / scale) // 2
        new[:, 0::2] -= left
        new[:, 1::2] -= top
        new *= scale
    else:
        new[:, 0::2] *= img_w
        new[:, 1::2] *= img_h
    return new, conf, cls
