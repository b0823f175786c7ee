"""Input side of the detection path: the reference's ``letter_box`` + ``TF.to_tensor``
(core/utils/image_process.py:29-66) as one HIP kernel on images that are already in device memory, and the batch assembly
``torch.stack`` does in the reference's collate functions (core/data/collate.py:9,24).  Image FILE I/O (cv2.imread) stays with the
caller, as in the reference."""
from typing import Sequence, Tuple

import numpy as np
import torch

from computervision.pytorch_amd.engine import letterbox_geometry, letterbox_u8


def letter_box(image, size, device=None, swap_rb: bool = False):
    """``letter_box`` (reference :48-66) fused with ``TF.to_tensor``: uint8 (h, w, 3) numpy array or tensor -> ((1, 3, H, W) fp32
    tensor in [0, 1] on the device, scale, [top, bottom, left, right])."""
    img = torch.as_tensor(np.ascontiguousarray(image) if isinstance(image, np.ndarray) else image)
    if device is not None:
        img = img.to(device, non_blocking=True)
    h, w = int(img.shape[0]), int(img.shape[1])
    H, W = size
    new_h, new_w, top, left, scale = letterbox_geometry(h, w, H, W)
    out = torch.empty(1, 3, H, W, device=img.device)
    letterbox_u8(img, out[0], letterbox=True, swap_rb=swap_rb)
    return out, scale, [top, H - new_h - top, left, W - new_w - left]


def read_image_and_convert_to_tensor(image, size, mode="rgb", letterbox=True, device=None):
    """Reference :29-45: image file (or an already decoded uint8 (h, w, 3) array) -> ((1, 3, H, W) fp32 tensor in [0, 1], h, w).  With
    ``letterbox`` the nearest-neighbour resize, the grey border and ``to_tensor`` are one kernel on the device (``letter_box`` above); the
    reference's other branch, a bicubic stretch, stays with OpenCV as it does there.  Decoding a file needs OpenCV (imported lazily)."""
    if isinstance(image, (str, bytes)):
        import cv2
        image = cv2.imread(image, cv2.IMREAD_COLOR | cv2.IMREAD_IGNORE_ORIENTATION)
        if mode == "rgb":
            image = cv2.cvtColor(image, cv2.COLOR_BGR2RGB)
    h, w = int(image.shape[0]), int(image.shape[1])
    if letterbox:
        return letter_box(image, size, device=device)[0], h, w
    import cv2
    arr = cv2.resize(np.asarray(image), tuple(size[::-1]), interpolation=cv2.INTER_CUBIC)
    x = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1).float().div(255.0).unsqueeze(0)
    return (x.to(device) if device is not None else x), h, w


def images_to_batch(images: Sequence, size: Tuple[int, int], device, letterbox: bool = True, swap_rb: bool = False) -> torch.Tensor:
    """A list of uint8 (h_i, w_i, 3) images of any sizes -> the (B, 3, H, W) fp32 network input, one kernel per image writing
    straight into its slot of the batch tensor (no per-image tensors, no ``torch.stack``)."""
    H, W = size
    batch = torch.empty(len(images), 3, H, W, device=device)
    for i, im in enumerate(images):
        img = torch.as_tensor(np.ascontiguousarray(im) if isinstance(im, np.ndarray) else im).to(device, non_blocking=True)
        letterbox_u8(img, batch[i], letterbox=letterbox, swap_rb=swap_rb)
    return batch
