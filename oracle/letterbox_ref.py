"""CPU oracle for the letterbox pre-processing -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of ``letter_box`` + ``TF.to_tensor`` (core/utils/image_process.py:48-66, :41 of the reference):

    scale = min(H / h, W / w); new_h, new_w = int(h * scale), int(w * scale)
    image = cv2.resize(image, (new_w, new_h), interpolation=cv2.INTER_NEAREST)
    top = (H - new_h) // 2; left = (W - new_w) // 2; pad with (128, 128, 128)
    tensor = HWC uint8 -> CHW float32 / 255

Parity pin: the size / padding arithmetic is the reference's own source, line for line.  ``cv2.resize(INTER_NEAREST)`` lives in
OpenCV (opencv-python, not vendored by the reference and NOT installed in this image -- ``import cv2`` fails): PARITY UNPINNED for
that one step; it restates OpenCV's published ``resizeNN`` (modules/imgproc/src/resize.cpp: ``x_ofs[x] = min(cvFloor(x * ifx),
ssize.width - 1)`` with ``ifx = 1 / (dsize.width / ssize.width)`` in double, rows likewise).
"""
import math

import numpy as np


def geometry(h, w, H, W):
    scale = min(H / h, W / w)
    new_h, new_w = int(h * scale), int(w * scale)
    top, left = (H - new_h) // 2, (W - new_w) // 2
    return new_h, new_w, top, left, scale


def resize_nearest(image, new_h, new_w):
    h, w = image.shape[:2]
    ify, ifx = 1.0 / (new_h / h), 1.0 / (new_w / w)
    ys = np.array([min(math.floor(y * ify), h - 1) for y in range(new_h)], dtype=np.int64)
    xs = np.array([min(math.floor(x * ifx), w - 1) for x in range(new_w)], dtype=np.int64)
    return image[ys][:, xs]


def letter_box(image, size):
    """uint8 (h, w, 3) -> uint8 (H, W, 3), scale, [top, bottom, left, right] (image_process.py:48-66)."""
    h, w, _ = image.shape
    H, W = size
    new_h, new_w, top, left, scale = geometry(h, w, H, W)
    out = np.full((H, W, 3), 128, dtype=np.uint8)
    out[top:top + new_h, left:left + new_w] = resize_nearest(image, new_h, new_w)
    return out, scale, [top, H - new_h - top, left, W - new_w - left]


def to_tensor(image_u8):
    """TF.to_tensor: HWC uint8 -> CHW float32 in [0, 1] (a float32 division by 255)."""
    return (image_u8.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)).astype(np.float32)
