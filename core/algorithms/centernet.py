"""CenterNet algorithm wrapper -- the duck-typed interface of the reference's ``CenterNetA``
(core/algorithms/centernet.py:25-338) for the INFERENCE path: ``__init__(cfg, device)``, ``build_model() -> (nn.Module, name)``,
``decode_boxes(pred, h, w, conf_threshold=None) -> (boxes, scores, classes)``, ``predict``.  The DLA-34 network and the
heat-map decode run on the MI355X engine (``computervision.pytorch_amd.dla``, ``cvx_centernet_decode``); ``build_loss`` returns the
engine's fused CombinedLoss; the heat-map target drawing (``generate_targets``, CPU work inside the reference's collate) is one
launch for the whole batch (``draw_targets`` -> ``cvx_centernet_draw_targets``).
"""
import numpy as np
import torch

from computervision.pytorch_amd import _lib as L
from computervision.pytorch_amd import engine as _engine
from computervision.pytorch_amd.dla import CenterNetDLA34, CenterNetLoss
from configs import CenternetConfig
from registry import model_registry


@model_registry("centernet")
class CenterNetA:
    def __init__(self, cfg: CenternetConfig, device):
        self.cfg, self.device = cfg, device
        self.num_classes = cfg.dataset.num_classes
        self.input_size = cfg.arch.input_size[1:]
        self.downsampling_ratio = cfg.arch.downsampling_ratio
        self.feature_size = [self.input_size[0] // self.downsampling_ratio, self.input_size[1] // self.downsampling_ratio]
        self.K = cfg.decode.max_boxes_per_img
        self.conf_threshold = cfg.decode.score_threshold
        self.nms_threshold = cfg.decode.nms_threshold
        self.use_nms = cfg.decode.use_nms
        self.letterbox_image = cfg.decode.letterbox_image

    def build_model(self):
        return CenterNetDLA34(self.num_classes), "CenterNet"

    def build_loss(self):
        """Reference :63-64: CombinedLoss(num_classes, hm_weight, wh_weight, off_weight) -- here the engine's fused CenterNetLoss
        (``cvx_centernet_loss``: value and gradient on the head rows)."""
        lc = self.cfg.loss
        return CenterNetLoss(self.num_classes, lc.hm_weight, lc.wh_weight, lc.off_weight)

    def draw_targets(self, labels):
        """What centernet_collate does image by image on the CPU (core/data/collate.py:52-68), for the batch in one launch: a list of
        (N_i, 6) label arrays [_, class id, cx, cy, w, h] -> [heatmap (B,h,w,nc), reg (B,K,2), wh (B,K,2), reg_mask (B,K), indices (B,K)] on
        the device (``cvx_centernet_draw_targets``), K = cfg.train.max_num_boxes (longer lists are truncated, as in the reference)."""
        dev = torch.device(self.device)
        K = int(getattr(self.cfg.train, "max_num_boxes", 30))
        packed = torch.zeros(len(labels), K, 5)
        counts = []
        for i, l in enumerate(labels):
            l = torch.as_tensor(np.asarray(l, dtype=np.float32) if not torch.is_tensor(l) else l).float()[:K]
            if l.shape[0]:
                packed[i, :l.shape[0]] = l[:, 1:6]
            counts.append(int(l.shape[0]))
        ratio = int(self.cfg.arch.downsampling_ratio)
        fh, fw = self.cfg.arch.input_size[1] // ratio, self.cfg.arch.input_size[2] // ratio
        return _engine.centernet_draw_targets(packed.to(dev), torch.tensor(counts, dtype=torch.int32, device=dev), (fh, fw), self.num_classes)

    def generate_targets(self, label):
        """Reference :66-112 for one image -> (heatmap (h,w,nc), reg (K,2), wh (K,2), reg_mask (K,), indices (K,)) on the device."""
        return tuple(t[0] for t in self.draw_targets([label]))

    # ---- decode ---------------------------------------------------------------------------------------
    def decode_raw(self, raw: torch.Tensor, fh: int, fw: int, conf_threshold=None):
        """The engine's head tensor (B, fh*fw, ld) -> device dict of cvx_centernet_decode (normalised boxes, before the
        letterbox inverse).  Batches decode image by image in one launch (the reference's decode_boxes is only meaningful for
        B = 1: it flattens the batch before the score mask and the NMS, centernet.py:300-307)."""
        nc_pad = (self.num_classes + 7) & ~7
        conf = self.conf_threshold if conf_threshold is None else conf_threshold
        return _engine.centernet_decode(raw, fh, fw, self.num_classes, nc_pad, nc_pad + 8, self.K, conf, self.nms_threshold, self.use_nms)

    def _finish(self, out, b, h, w):
        n = int(out["counts"][b])
        if n < 0:
            raise L.CvxError("cvx_centernet_decode: more than 2048 scores tie at the K-th value (flat heat-map)")
        keep = out["keep"][b, :n].long()
        boxes, scores, classes = out["boxes"][b][keep].cpu(), out["scores"][b][keep].cpu(), out["classes"][b][keep].cpu()
        # reverse_letter_box(xywh=False), core/utils/image_process.py:100-129 -- same fp32 tensor operations as the reference
        nb = boxes.clone()
        nb[..., ::2] *= self.input_size[1]
        nb[..., 1::2] *= self.input_size[0]
        scale = max(h / self.input_size[0], w / self.input_size[1])
        top = (self.input_size[0] - h / scale) // 2
        left = (self.input_size[1] - w / scale) // 2
        nb[..., 0] -= left
        nb[..., 2] -= left
        nb[..., 1] -= top
        nb[..., 3] -= top
        nb *= scale
        return nb.numpy(), scores.numpy(), classes.numpy().astype(np.int64)

    def decode_boxes(self, pred, h, w, conf_threshold=None):
        """pred: the model's (1, H/4, W/4, nc + 4) output (the reference's contract) or the engine's raw head tensor."""
        if pred.dim() == 4:                                        # reference layout -> the decode kernel's padded rows
            B, fh, fw, ch = pred.shape
            assert B == 1 and ch == self.num_classes + 4, "decode_boxes handles one image, as the reference does"
            nc, nc_pad = self.num_classes, (self.num_classes + 7) & ~7
            raw = torch.zeros(B, fh * fw, nc_pad + 16, device=pred.device)
            flat = pred.reshape(B, fh * fw, ch).float()
            raw[..., :nc] = flat[..., :nc]
            raw[..., nc_pad:nc_pad + 2] = flat[..., nc:nc + 2]
            raw[..., nc_pad + 8:nc_pad + 10] = flat[..., nc + 2:]
        else:
            raw, (fh, fw) = pred, self.feature_size
        return self._finish(self.decode_raw(raw, fh, fw, conf_threshold), 0, h, w)

    def predict_tensor(self, model, image: torch.Tensor, image_h: int, image_w: int):
        """``predict`` minus file I/O and drawing: image (1,3,H,W) in [0,1], already letterboxed."""
        model.eval()
        with torch.no_grad():
            raw = model.forward_raw(image.to(self.device))
            return self._finish(self.decode_raw(raw, image.shape[2] // 4, image.shape[3] // 4), 0, image_h, image_w)

    def predict(self, model, image_path, print_on, save_result):
        try:
            import cv2
        except ImportError as e:  # pragma: no cover
            raise ImportError("predict() needs opencv-python for image I/O; use predict_tensor() with a prepared tensor") from e
        bgr = cv2.imread(image_path, cv2.IMREAD_COLOR | cv2.IMREAD_IGNORE_ORIENTATION)
        img = cv2.cvtColor(bgr, cv2.COLOR_BGR2RGB)
        from core.utils.image_process import read_image_and_convert_to_tensor
        x, h, w = read_image_and_convert_to_tensor(img, self.input_size, letterbox=self.letterbox_image, device=self.device)
        boxes, scores, classes = self.predict_tensor(model, x, h, w)
        for b, s_, c in zip(boxes, scores, classes):
            x1, y1, x2, y2 = (int(round(float(v))) for v in b)
            cv2.rectangle(bgr, (x1, y1), (x2, y2), (0, 255, 0), 2)
            if print_on:
                print(f"class {int(c)} score {float(s_):.3f} box {[float(v) for v in b]}")
        return bgr
