"""SSD algorithm wrapper -- the duck-typed interface of the reference's ``Ssd`` (core/algorithms/ssd.py:27-535) for the INFERENCE
path: ``__init__(cfg, device)`` (prior boxes included), ``build_model() -> (nn.Module, name)``,
``decode_boxes(preds, h, w, conf_threshold=None)``, ``predict``.  Network, softmax + box decode and the per-class NMS run on the
MI355X engine (``computervision.pytorch_amd.ssd``, ``cvx_ssd_decode``, ``cvx_nms_variant``); ``build_loss`` (MultiBoxLossV2) raises.
"""
import numpy as np
import torch

from computervision.pytorch_amd import _lib as L
from computervision.pytorch_amd import engine as _engine
from computervision.pytorch_amd.ssd import MultiBoxLoss, SSD300VGG
from configs import SsdConfig
from registry import model_registry

MAX_DET = 1024          # rows per image and class cvx_nms_variant is first asked for (a full block is retried with 4x the room)


@model_registry("ssd")
class Ssd:
    def __init__(self, cfg: SsdConfig, device):
        self.cfg, self.device = cfg, device
        self.input_image_size = cfg.arch.input_size[1:]
        self.anchor_sizes, self.feature_shapes, self.aspect_ratios = cfg.arch.anchor_sizes, cfg.arch.feature_shapes, cfg.arch.aspect_ratios
        self.anchors = self._get_ssd_anchors()
        self.num_anchors = self.anchors.shape[0]
        self.num_classes = cfg.dataset.num_classes
        self.neg_pos_ratio = cfg.loss.neg_pos
        self.variance = np.repeat(np.array(cfg.loss.variance, dtype=np.float32), 2, axis=0)
        self.overlap_threshold = cfg.loss.overlap_threshold
        self.conf_threshold = cfg.decode.confidence_threshold
        self.nms_threshold = cfg.decode.nms_threshold
        self.letterbox_image = cfg.decode.letterbox_image
        self._priors_dev = None

    def _get_ssd_anchors(self):
        """Prior boxes (reference :482-535): per feature map a grid of centres, per centre one box per aspect ratio plus the
        sqrt(min * max) square; corners normalised and clipped to [0, 1]; float32 (8732, 4)."""
        image_h, image_w = self.input_image_size
        out = []
        for i, fh in enumerate(self.feature_shapes):
            mn, mx = self.anchor_sizes[i], self.anchor_sizes[i + 1]
            ws, hs = [], []
            for ar in self.aspect_ratios[i]:
                if ar == 1:
                    ws += [mn, np.sqrt(mn * mx)]
                    hs += [mn, np.sqrt(mn * mx)]
                else:
                    ws.append(mn * np.sqrt(ar))
                    hs.append(mn / np.sqrt(ar))
            half_w, half_h = np.array(ws) / 2.0, np.array(hs) / 2.0
            step = [image_h / fh, image_w / fh]
            gx, gy = np.meshgrid(np.linspace(0.5 * step[1], image_w - 0.5 * step[1], fh), np.linspace(0.5 * step[0], image_h - 0.5 * step[0], fh))
            a = np.tile(np.concatenate((gx.reshape(-1, 1), gy.reshape(-1, 1)), 1), (1, (len(self.aspect_ratios[i]) + 1) * 2))
            a[:, ::4] -= half_w
            a[:, 1::4] -= half_h
            a[:, 2::4] += half_w
            a[:, 3::4] += half_h
            a[:, ::2] /= image_w
            a[:, 1::2] /= image_h
            out.append(np.clip(a, 0.0, 1.0).reshape(-1, 4))
        return np.concatenate(out, 0).astype(np.float32)

    def build_model(self):
        if self.cfg.arch.backbone != "vgg" or self.input_image_size[0] != 300:
            raise L.CvxError("the MI355X engine builds SSD300 with the VGG16-BN backbone (the reference's configuration)")
        if self.cfg.train.pretrained:
            raise L.CvxError("train.pretrained needs a torchvision download; load a checkpoint with load_state_dict instead")
        return SSD300VGG(self.num_classes), f"SSD{self.input_image_size[0]}_vgg"

    def build_loss(self):
        """Reference :62-66: MultiBoxLossV2(neg_pos_ratio=cfg.loss.neg_pos, num_classes) -- here the engine's fused MultiBoxLoss
        (``cvx_multibox_loss``: values and gradient w.r.t. (loc, conf), radix-select hard-negative mining)."""
        return MultiBoxLoss(self.cfg.loss.neg_pos, self.num_classes)

    def generate_targets(self, label):
        """Reference :327-388 for one image: label (N, 6) [_, class id, cx, cy, w, h] (numpy or tensor) -> (8732, 4 + (nc + 1) + 1) tensor on
        the device, by the batch kernel below."""
        lab = torch.as_tensor(np.asarray(label, dtype=np.float32) if not torch.is_tensor(label) else label).float()
        return self.encode_targets([lab])[0]

    def encode_targets(self, labels):
        """What ssd_collate does image by image on the CPU (core/data/collate.py:32-49), for the batch in two launches: a list of (N_i, 6)
        label arrays -> y_true (B, 8732, 4 + (nc + 1) + 1) on the device (``cvx_ssd_encode_targets``)."""
        dev = torch.device(self.device)
        nmax = max([int(l.shape[0]) for l in labels] + [1])
        packed = torch.zeros(len(labels), nmax, 5)
        for i, l in enumerate(labels):
            l = torch.as_tensor(l).float()
            if l.shape[0]:
                packed[i, :l.shape[0]] = l[:, 1:6]
        counts = torch.tensor([int(l.shape[0]) for l in labels], dtype=torch.int32)
        if self._priors_dev is None or self._priors_dev.device != dev:
            self._priors_dev = torch.from_numpy(self.anchors).to(dev)
        return _engine.ssd_encode_targets(packed.to(dev), counts.to(dev), self._priors_dev, self.num_classes, self.overlap_threshold,
                                          self.variance[::2].tolist())

    def decode_device(self, preds, conf_threshold=None):
        """(loc, conf) on the device -> per image ((n, 6) tensor [x1, y1, x2, y2, label, conf], (n, 2) kept (prior, class column)):
        classes ascending, scores descending inside a class, like the reference's loop (reference :246-274)."""
        conf_thr = self.conf_threshold if conf_threshold is None else conf_threshold
        loc, conf = preds
        dev = loc.device
        if self._priors_dev is None or self._priors_dev.device != dev:
            self._priors_dev = torch.from_numpy(self.anchors).to(dev)
        boxes, prob, cmax = _engine.ssd_decode(loc, conf, self._priors_dev, self.variance[::2].tolist(), with_class_max=True)
        B, A, _ = boxes.shape
        bt = boxes.transpose(1, 2).contiguous()                                  # (B, 4, A): corner boxes, channel-major for cvx_nms
        active = (cmax > conf_thr).cpu().tolist()       # the decode kernel's per-class maxima: one host read decides every class
        classes = [c for c in range(1, self.num_classes + 1) if active[c]]
        empty = (torch.zeros(0, 6, device=dev), torch.zeros(0, 2, dtype=torch.long, device=dev))
        if not classes:
            return [empty for _ in range(B)]
        # one NMS launch per class that has a score above the threshold, all queued back to back; ONE host read (the counts of every
        # class and image) afterwards, and one masked gather per image instead of a cat per (class, image)
        # ONE NMS launch for all active classes: (class, image) pairs are the launch's batch -- a workgroup per pair, C' * B of them, instead
        # of C' launches of B workgroups (measured with 256 planted candidates per image, 20 classes active: 4.7 -> 0.6 ms per batch of 32)
        nc_act = len(classes)
        cidx = torch.tensor(classes, device=dev)
        stacked = torch.cat((bt.unsqueeze(0).expand(nc_act, -1, -1, -1), prob.index_select(2, cidx).permute(2, 0, 1).unsqueeze(2)), 2).reshape(nc_act * B, 5, A)
        max_det = MAX_DET
        while True:
            r_, i_, c_ = _engine.nms(stacked, float(conf_thr), self.nms_threshold, max_det=max_det, variant="vanilla", boxes_xyxy=True)
            rows = r_.view(nc_act, B, *r_.shape[1:])                             # (C', B, max_det, 6)
            index = i_.long().view(nc_act, B, -1)                                # (C', B, max_det)
            counts = c_.view(nc_act, B)                                          # (C', B)
            counts_h = counts.cpu()                                              # the one host read (one more per retry)
            if bool((counts_h < 0).any()):  # 8732 priors never exceed the 16384 candidates the in-LDS sort holds; kept for other prior sets
                raise L.CvxError("cvx_nms: more than 16384 candidates of one class above the confidence threshold in one image")
            # a full row block may have been cut short: ask again with room for every prior (the reference's decode_boxes has no limit)
            if int(counts_h.max()) < max_det or max_det >= 16384:
                break
            max_det = min(max_det * 4, 16384)
        cls_col = torch.tensor(classes, device=dev).view(-1, 1, 1).expand(-1, B, rows.shape[2])      # class column per slot
        valid = torch.arange(rows.shape[2], device=dev).view(1, 1, -1) < counts.unsqueeze(2)         # (C', B, MAX_DET)
        det = torch.cat((rows[..., :4], (cls_col - 1).unsqueeze(3).to(rows.dtype), rows[..., 4:5]), 3)
        pairs = torch.stack((index, cls_col), 3)
        per_image = counts_h.sum(0).tolist()
        # ONE boolean-mask gather for the whole batch (each one is a host round trip: a gather per image was 64 of them per batch): the
        # mask walks (image, class, rank) in order -- classes ascending, scores descending inside a class, like the reference's loop
        vb = valid.permute(1, 0, 2)
        det_all, pairs_all = det.permute(1, 0, 2, 3)[vb], pairs.permute(1, 0, 2, 3)[vb]
        return [(d, p) if n > 0 else empty for d, p, n in zip(det_all.split(per_image), pairs_all.split(per_image), per_image)]

    def decode_boxes(self, preds, h, w, conf_threshold=None):
        results = []
        for det, _ in self.decode_device(preds, conf_threshold):
            o = det.cpu().numpy()
            if len(o):
                xy, wh = (o[:, 0:2] + o[:, 2:4]) / 2, o[:, 2:4] - o[:, 0:2]
                o[:, :4] = self._correct_boxes(xy, wh, self.input_image_size, [h, w])
            results.append(o if len(o) else [])
        return results

    def _correct_boxes(self, box_xy, box_wh, input_shape, image_shape):
        """yolo_correct_boxes (core/utils/image_process.py:161-181)."""
        xywh = np.concatenate([box_xy, box_wh], axis=-1)
        if self.letterbox_image:
            ih, iw = image_shape
            h, w = input_shape
            scale = max(ih / h, iw / w)
            top, left = (h - ih / scale) // 2, (w - iw / scale) // 2
            cx, cy, bw, bh = xywh[:, 0] * w - left, xywh[:, 1] * h - top, xywh[:, 2] * w, xywh[:, 3] * h
            return np.stack([(cx - bw / 2) * scale, (cy - bh / 2) * scale, (cx + bw / 2) * scale, (cy + bh / 2) * scale], -1)
        out = np.stack([xywh[:, 0] - xywh[:, 2] / 2, xywh[:, 1] - xywh[:, 3] / 2, xywh[:, 0] + xywh[:, 2] / 2, xywh[:, 1] + xywh[:, 3] / 2], -1)
        out[:, ::2] *= image_shape[1]
        out[:, 1::2] *= image_shape[0]
        return out

    def predict_tensor(self, model, images: torch.Tensor, h, w, conf_threshold=None):
        model.eval()
        with torch.no_grad():
            return self.decode_boxes(model(images), h, w, conf_threshold)

    def predict(self, model, image_path, print_on, save_result):
        """Reference :69-100: read + letterbox to 300 x 300, forward, decode, draw.  Image I/O needs OpenCV (lazy import)."""
        import cv2
        img = cv2.cvtColor(cv2.imread(image_path), cv2.COLOR_BGR2RGB)
        from core.utils.image_process import read_image_and_convert_to_tensor
        x, h, w = read_image_and_convert_to_tensor(img, self.input_image_size, letterbox=self.letterbox_image, device=self.device)
        results = self.predict_tensor(model, x, h, w)
        out = cv2.cvtColor(img, cv2.COLOR_RGB2BGR)
        for x1, y1, x2, y2, cls, sc in (results[0] if len(results[0]) else []):
            cv2.rectangle(out, (int(x1), int(y1)), (int(x2), int(y2)), (0, 255, 0), 2)
            cv2.putText(out, f"{int(cls)}:{sc:.2f}", (int(x1), max(int(y1) - 3, 0)), cv2.FONT_HERSHEY_SIMPLEX, 0.5, (0, 255, 0), 1)
        return out
