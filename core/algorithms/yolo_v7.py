"""YOLOv7 algorithm wrapper -- the duck-typed interface of the reference's ``YOLOv7`` (core/algorithms/yolo_v7.py:27-424) for the
INFERENCE path: ``__init__(cfg, device)``, ``get_anchors``, ``build_model() -> (nn.Module, name)``,
``decode_box(preds, image_h, image_w, conf_threshold=None)``, ``predict``.  Network, anchor decode and per-class NMS run on
the MI355X engine (``computervision.pytorch_amd.yolov7``, ``cvx_yolo7_decode``, ``cvx_nms_variant``); ``build_loss``
(Yolo7Loss with SimOTA matching) raises.
"""
import numpy as np
import torch

from computervision.pytorch_amd import _lib as L
from computervision.pytorch_amd import engine as _engine
from computervision.pytorch_amd.yolov7 import Yolo7L, Yolo7Loss
from configs import Yolo7Config
from registry import model_registry

MAX_DET = 1024          # rows per image cvx_nms_variant is first asked for; a full block is retried with 4x the room, up to ...
MAX_CANDIDATES = 16384  # ... the candidates per image the in-LDS sort holds (the reference's _nms has no limit: only beyond this it raises)


@model_registry("yolo7")
class YOLOv7:
    def __init__(self, cfg: Yolo7Config, device) -> None:
        self.cfg, self.device = cfg, device
        self.anchors = self.get_anchors()
        self.num_classes = cfg.dataset.num_classes
        self.input_image_size = list(cfg.arch.input_size[1:])
        self.bbox_attrs = 5 + self.num_classes
        self.anchors_mask = cfg.arch.anchors_mask
        self.letterbox_image = cfg.decode.letterbox_image
        self.conf_threshold = cfg.decode.conf_threshold
        self.nms_threshold = cfg.decode.nms_threshold

    def get_anchors(self) -> np.ndarray:
        return np.array(self.cfg.arch.anchors, dtype=np.float32).reshape(-1, 2)

    def build_model(self):
        if self.cfg.arch.phi != "l":
            raise L.CvxError("the MI355X engine builds YOLOv7-l (phi = 'l'), the reference's configuration")
        model = Yolo7L(self.num_classes)
        if self.cfg.train.pretrained:                            # Yolo7.__init__ loads the checkpoint itself (yolov7_model.py:444-445)
            from core.utils.ckpt import CheckPoint
            CheckPoint.load_pretrained(model, self.cfg.train.pretrained_weights)
        return model, "YOLOv7"

    def build_loss(self):
        """Reference :57-64: Yolo7Loss(anchors, num_classes, input_shape, anchors_mask, label_smoothing) -- here the engine's fused loss
        (``cvx_yolo7_loss``: candidates, SimOTA assignment, the three terms and their gradient on the head rows)."""
        return Yolo7Loss(anchors=np.asarray(self.anchors, dtype=np.float32).reshape(-1, 2), num_classes=self.num_classes,
                         input_shape=self.input_image_size, anchors_mask=self.anchors_mask, label_smoothing=self.cfg.loss.label_smoothing)

    # ---- decode ---------------------------------------------------------------------------------------
    def _levels(self, model):
        g = model._last_engine.graph
        return g.level_hw, [[tuple(self.anchors[i]) for i in mask] for mask in self.anchors_mask]

    def decode_rows(self, model, rows: torch.Tensor):
        """The engine's head rows -> (decoded (B, 3*sum, 5+nc) = the reference's ``decoded_outputs``, NMS input)."""
        level_hw, anchors = self._levels(model)
        return _engine.yolo7_decode(rows, self.num_classes, level_hw, anchors, self.input_image_size)

    def nms_device(self, y: torch.Tensor, dec: torch.Tensor, conf_threshold=None):
        """Per-class greedy NMS with score = objectness * best class probability (``_nms``, yolo_v7.py:348-415) on the device:
        per image an (n, 7) tensor [x1, y1, x2, y2, obj_conf, class_conf, class_pred] in normalised corner coordinates -- classes
        ascending, scores descending inside a class, like the reference's concatenation -- and the kept rows of ``dec``."""
        conf = self.conf_threshold if conf_threshold is None else conf_threshold
        # cvx_nms keeps scores > threshold (ultralytics_ops.py:190), the reference here keeps >= : the next float below
        thr = float(np.nextafter(np.float32(conf), np.float32(-1.0)))
        max_det = MAX_DET
        while True:
            rows, index, counts = _engine.nms(y, thr, self.nms_threshold, max_det=max_det, variant="vanilla")
            counts_h = counts.cpu().tolist()                       # the ONE host read of the tail (one more per retry)
            for b, n in enumerate(counts_h):
                if n < 0:  # the in-LDS sort holds 16384 candidates per image: only then is the reference's unlimited _nms out of reach
                    raise L.CvxError(f"cvx_nms: more than {MAX_CANDIDATES} candidates above the confidence threshold in image {b}")
            # a full row block may have been cut short: ask again with room for every candidate (mAP-style runs at conf 0.001)
            if max(counts_h, default=0) < max_det or max_det >= MAX_CANDIDATES:
                break
            max_det = min(max_det * 4, MAX_CANDIDATES)
        # re-order every image's kept rows at once (no per-image kernels): class ascending, cvx_nms's descending score inside a class;
        # rows past an image's count sort to the end
        B, K = rows.shape[0], rows.shape[1]
        valid = torch.arange(K, device=rows.device).unsqueeze(0) < counts.unsqueeze(1)
        order = torch.argsort(torch.where(valid, rows[:, :, 5], torch.full_like(rows[:, :, 5], float("inf"))), dim=1, stable=True)
        r = rows.gather(1, order.unsqueeze(2).expand(-1, -1, rows.shape[2]))
        idx = index.long().gather(1, order).clamp_(min=0)
        d = dec.gather(1, idx.unsqueeze(2).expand(-1, -1, dec.shape[2]))
        cconf = d[:, :, 5:5 + self.num_classes].gather(2, r[:, :, 5:6].long().clamp_(0, self.num_classes - 1))
        det = torch.cat((r[:, :, :4], d[:, :, 4:5], cconf, r[:, :, 5:6]), 2)
        return [(det[b, :n], idx[b, :n]) if n > 0 else (None, None) for b, n in enumerate(counts_h)]

    def decode_box(self, preds, image_h, image_w, conf_threshold=None, model=None):
        """Reference signature (yolo_v7.py:234).  ``preds``: the model's output tuple; the decode reads the engine's rows of that
        forward directly (``model`` defaults to the model ``predict`` was called with)."""
        model = model or self._model
        dec, y = self.decode_rows(model, model.last_rows)
        results = []
        for det, _ in self.nms_device(y, dec, conf_threshold):
            if det is None:
                results.append(None)
                continue
            o = det.cpu().numpy()
            xy, wh = (o[:, 0:2] + o[:, 2:4]) / 2, o[:, 2:4] - o[:, 0:2]
            o[:, :4] = self._correct_boxes(xy, wh, self.input_image_size, [image_h, image_w])
            results.append(o)
        return results

    def _correct_boxes(self, box_xy, box_wh, input_shape, image_shape):
        """yolo_correct_boxes (core/utils/image_process.py:161-181): letterbox inverse, or plain scaling to the image size."""
        xywh = np.concatenate([box_xy, box_wh], axis=-1)
        if self.letterbox_image:
            ih, iw = image_shape
            h, w = input_shape
            scale = max(ih / h, iw / w)
            top, left = (h - ih / scale) // 2, (w - iw / scale) // 2
            cx, cy = xywh[:, 0] * w - left, xywh[:, 1] * h - top
            bw, bh = xywh[:, 2] * w, xywh[:, 3] * h
            out = np.stack([(cx - bw / 2) * scale, (cy - bh / 2) * scale, (cx + bw / 2) * scale, (cy + bh / 2) * scale], -1)
            return out
        out = np.stack([xywh[:, 0] - xywh[:, 2] / 2, xywh[:, 1] - xywh[:, 3] / 2, xywh[:, 0] + xywh[:, 2] / 2, xywh[:, 1] + xywh[:, 3] / 2], -1)
        out[:, ::2] *= image_shape[1]
        out[:, 1::2] *= image_shape[0]
        return out

    def predict_tensor(self, model, images: torch.Tensor, image_h, image_w, conf_threshold=None):
        model.eval()
        self._model = model
        with torch.no_grad():
            preds = model(images)
        return self.decode_box(preds, image_h, image_w, conf_threshold, model)

    def predict(self, model, image_path, print_on, save_result):
        """Reference :70-112: read + letterbox to the network size, forward, decode, draw.  Image I/O needs OpenCV (lazy import)."""
        import cv2
        img = cv2.cvtColor(cv2.imread(image_path), cv2.COLOR_BGR2RGB)
        from core.utils.image_process import read_image_and_convert_to_tensor
        x, h, w = read_image_and_convert_to_tensor(img, self.input_image_size, letterbox=self.letterbox_image, device=self.device)
        results = self.predict_tensor(model, x, h, w)
        out = cv2.cvtColor(img, cv2.COLOR_RGB2BGR)
        if results[0] is None:
            return out
        for x1, y1, x2, y2, oc, cc, cls in results[0]:
            cv2.rectangle(out, (int(x1), int(y1)), (int(x2), int(y2)), (0, 255, 0), 2)
            cv2.putText(out, f"{int(cls)}:{oc * cc:.2f}", (int(x1), max(int(y1) - 3, 0)), cv2.FONT_HERSHEY_SIMPLEX, 0.5, (0, 255, 0), 1)
        return out
