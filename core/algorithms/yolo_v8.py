"""YOLOv8 algorithm wrapper -- same duck-typed interface as the reference's ``YOLOv8``
(core/algorithms/yolo_v8.py:127-242): ``__init__(cfg, device)``, ``build_model() -> (nn.Module, name)``,
``build_loss(model)``, ``predict(...)``, ``decode_box(preds, image_h, image_w, conf_threshold=None)``.
Model, loss and the decode + NMS tail run on the MI355X engine (``computervision.pytorch_amd``).
"""
import numpy as np
import torch

from computervision.pytorch_amd._lib import CvxError

from computervision.pytorch_amd import engine as _engine
from computervision.pytorch_amd.model import Yolo8
from computervision.pytorch_amd.train import V8DetectionLoss
from configs import Yolo8DetConfig
from core.utils.boxes import undo_letterbox
from registry import model_registry

Loss = V8DetectionLoss          # the reference exposes the criterion class under this name (yolo_v8.py:25)

_NAMES = {"n": "YOLOv8n", "s": "YOLOv8s", "m": "YOLOv8m", "l": "YOLOv8l", "x": "YOLOv8x"}


@model_registry("yolo8_det")
class YOLOv8:
    def __init__(self, cfg: Yolo8DetConfig, device):
        self.cfg = cfg
        self.device = device
        self.model_type = cfg.arch.model_type
        self.num_classes = cfg.dataset.num_classes
        self.input_image_size = cfg.arch.input_size[1:]
        self.conf_threshold = cfg.decode.conf_threshold
        self.iou_threshold = cfg.decode.nms_threshold
        self.max_det = cfg.decode.max_det
        self.letterbox_image = cfg.decode.letterbox_image

    def build_model(self):
        if self.model_type not in _NAMES:
            raise ValueError(f"model_type: {self.model_type} is not supported")
        loss_scale = float(getattr(getattr(self.cfg, "engine", None), "loss_scale", 1024.0))
        return Yolo8(self.model_type, self.num_classes, loss_scale=loss_scale), _NAMES[self.model_type]

    def build_loss(self, model):
        return V8DetectionLoss(cfg=self.cfg, model=model)

    # ---- inference tail -------------------------------------------------------------------------
    def non_max_suppression(self, preds, conf_threshold=None):
        """(y, feats) or y (B, 4+nc, A) -> list of (k_i, 6) device tensors [x1,y1,x2,y2,conf,cls]
        (ultralytics_ops.non_max_suppression's return format, class-aware, max_det rows)."""
        y = preds[0] if isinstance(preds, (list, tuple)) else preds
        conf = self.conf_threshold if conf_threshold is None else conf_threshold
        rows, _, counts = _engine.nms(y, conf, self.iou_threshold, self.max_det)
        counts = counts.cpu().tolist()
        if min(counts, default=0) < 0:   # cvx_nms_variant signals "more candidates than the in-LDS sort holds" with -1: never slice rows[:-1]
            raise CvxError(f"non_max_suppression: image {counts.index(min(counts))} has more than 16384 candidates above conf {conf} "
                           "(the in-LDS sort capacity of cvx_nms); raise the confidence threshold")
        return [rows[i, :k] for i, k in enumerate(counts)]

    def decode_box(self, preds, image_h, image_w, conf_threshold=None):
        out = self.non_max_suppression(preds, conf_threshold)
        assert len(out) == 1, "仅支持单张图片的预测"
        return undo_letterbox(out[0].cpu().numpy(), self.input_image_size, (image_h, image_w), self.letterbox_image)

    def predict_tensor(self, model, image: torch.Tensor, image_h: int, image_w: int):
        """``predict`` minus file I/O and drawing: image (1,3,H,W) in [0,1] already letterboxed."""
        model.eval()
        with torch.no_grad():
            preds = model(image.to(self.device))
            return self.decode_box(preds, image_h, image_w)

    def predict(self, model, image_path, print_on, save_result):
        """Reads and letterboxes the image with OpenCV, as the reference does (image_process.py:29-66);
        OpenCV is I/O plumbing outside the hot path and is imported lazily."""
        try:
            import cv2
        except ImportError as e:  # pragma: no cover
            raise ImportError("predict() needs opencv-python for image I/O; use predict_tensor() with a prepared tensor") from e
        img = cv2.cvtColor(cv2.imread(image_path, cv2.IMREAD_COLOR | cv2.IMREAD_IGNORE_ORIENTATION), cv2.COLOR_BGR2RGB)
        h, w, _ = img.shape
        H, W = self.input_image_size
        if self.letterbox_image:                        # letter_box + to_tensor on the GPU (one kernel; image_process.py:48-66)
            from core.utils.image_process import letter_box
            x, _, _ = letter_box(img, (H, W), device=self.device)
        else:
            img = cv2.resize(img, (W, H), interpolation=cv2.INTER_CUBIC)
            x = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float().div(255.0).unsqueeze(0)
        boxes, scores, classes = self.predict_tensor(model, x, h, w)
        bgr = cv2.imread(image_path, cv2.IMREAD_COLOR | cv2.IMREAD_IGNORE_ORIENTATION)
        if boxes.shape[0] == 0:
            print("No object detected")
            return bgr                                  # the reference returns the untouched BGR image (yolo_v8.py:193-195)
        # the reference returns the image with the detections drawn on it (show_detection_results, visualize.py:15)
        for b, s_, c in zip(boxes, scores, classes):
            x1, y1, x2, y2 = (int(round(float(v))) for v in b)
            cv2.rectangle(bgr, (x1, y1), (x2, y2), (0, 255, 0), 2)
            cv2.putText(bgr, f"{int(c)}: {float(s_):.2f}", (x1, max(y1 - 4, 10)), cv2.FONT_HERSHEY_SIMPLEX, 0.5, (0, 255, 0), 1)
            if print_on:
                print(f"class {int(c)} score {float(s_):.3f} box {[float(v) for v in b]}")
        if save_result:
            import os
            os.makedirs(self.cfg.decode.test_results, exist_ok=True)
            cv2.imwrite(os.path.join(self.cfg.decode.test_results, os.path.basename(image_path)), bgr)
        return bgr
