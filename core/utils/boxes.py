"""Host-side box bookkeeping after the GPU NMS: normalise to the network input and undo the letterbox.

Follows YOLOv8.decode_box (reference core/algorithms/yolo_v8.py:229-242) and reverse_letter_box_numpy
(reference core/utils/image_process.py:69-97).  A few dozen boxes per image: numpy on the host, as in
the reference.
"""
import numpy as np


def undo_letterbox(rows: np.ndarray, input_hw, image_hw, letterbox: bool = True):
    """rows (k,6) [x1,y1,x2,y2,conf,cls] in network-input pixels -> (boxes (k,4) in original-image pixels, conf, cls)."""
    rows = np.asarray(rows, dtype=np.float32).reshape(-1, 6)
    in_h, in_w = (float(v) for v in input_hw)
    img_h, img_w = (float(v) for v in image_hw)
    conf, cls = rows[:, 4].copy(), rows[:, 5].astype(np.int64)     # the reference's np.int is gone from NumPy >= 1.24
    box = rows[:, :4].copy()
    if letterbox:
        gain = max(img_h / in_h, img_w / in_w)
        pad_top = (in_h - img_h / gain) // 2
        pad_left = (in_w - img_w / gain) // 2
        box[:, 0::2] -= pad_left
        box[:, 1::2] -= pad_top
        box *= gain
    else:
        box[:, 0::2] *= img_w / in_w
        box[:, 1::2] *= img_h / in_h
    return box, conf, cls
