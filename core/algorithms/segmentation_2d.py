"""DeepLabv3+ algorithm wrapper -- the duck-typed interface of the reference's ``DeeplabV3PlusA``
(core/algorithms/segmentation_2d.py:43-201): ``__init__(cfg, device)``, ``build_model() -> (nn.Module, name)``,
``build_loss()`` (FocalLoss / CrossEntropyLoss, :59-64, as the engine's fused ``SegLoss``), ``postprocess_seg2d`` (argmax ->
colour map), ``predict``.  The network runs on the MI355X engine (``computervision.pytorch_amd.deeplab``).
"""
import os

import numpy as np
import torch

from computervision.pytorch_amd import _lib as L
from computervision.pytorch_amd.deeplab import DeepLabV3PlusR101, SegLoss
from configs import DeeplabV3PlusConfig
from registry import model_registry


def voc_colormap(n: int = 21):
    """The PASCAL VOC palette (core/data/segmentation_dataset.py:14-36 lists its first 21 entries): class index bits spread over
    the high bits of R, G, B, three bits per round."""
    cmap = []
    for i in range(n):
        r = g = b = 0
        c = i
        for j in range(8):
            r |= ((c >> 0) & 1) << (7 - j)
            g |= ((c >> 1) & 1) << (7 - j)
            b |= ((c >> 2) & 1) << (7 - j)
            c >>= 3
        cmap.append((r, g, b))
    return cmap


def postprocess_seg2d(dataset_type, pred, device):
    """(B, C, H, W) logits -> (B, H, W, 3) colours (reference :20-30)."""
    if dataset_type.lower() not in ("voc", "sbd"):
        raise NotImplementedError(f"不支持{dataset_type}数据集")
    colormap = torch.tensor(voc_colormap(), device=device)
    return colormap[torch.argmax(pred, dim=1).long(), :]


@model_registry("deeplabv3plus")
class DeeplabV3PlusA:
    def __init__(self, cfg: DeeplabV3PlusConfig, device) -> None:
        self.cfg, self.device = cfg, device
        self.loss_type = cfg.loss.loss_type
        self.num_classes = cfg.dataset.num_classes
        self.input_image_size = cfg.arch.input_size
        self.batch_size = cfg.train.batch_size
        self.dataset_name = cfg.dataset.dataset_name

    def build_model(self):
        if self.cfg.arch.output_stride != 16 or self.cfg.arch.backbone_name != "resnet101":
            raise L.CvxError("the MI355X engine builds DeepLabv3+ with a ResNet-101 backbone at output stride 16 (the reference's configuration)")
        if self.cfg.arch.backbone_pretrained:
            raise L.CvxError("backbone_pretrained needs a torchvision download; load a checkpoint with load_state_dict instead")
        return DeepLabV3PlusR101(self.num_classes), "deeplabv3plus"

    def build_loss(self):
        """Reference :59-64: "ce" -> nn.CrossEntropyLoss(reduction="mean"), "focal" -> FocalLoss() (alpha 0.25, gamma 2,
        ignore_index -100, mean over all pixels; core/loss/focal_loss.py:6-22)."""
        if self.loss_type not in ("ce", "focal"):
            raise L.CvxError(f"loss_type {self.loss_type!r}: the reference knows 'ce' and 'focal'")
        return SegLoss(self.loss_type)

    def predict_tensor(self, model, images: torch.Tensor):
        """(B,3,H,W) normalised images on the device -> (B,H,W,3) class colours (the tensor part of ``predict``)."""
        model.eval()
        with torch.no_grad():
            return postprocess_seg2d(self.dataset_name, model(images), images.device)

    def predict(self, model, image_path, print_on, save_result):
        """Reference :80-113: read, resize to the network size (no letterbox), forward, colour, resize back, blend 50 % with the
        original image, RGB -> BGR.  Image I/O needs OpenCV, imported lazily as in the other algorithm classes."""
        import cv2
        original = cv2.cvtColor(cv2.imread(image_path), cv2.COLOR_BGR2RGB)
        h, w = original.shape[:2]
        size = self.cfg.arch.input_size[1:]
        img = cv2.resize(original, (size[1], size[0])).astype(np.float32) / 255.0
        x = torch.from_numpy(img).permute(2, 0, 1).unsqueeze(0).to(self.device)
        colours = self.predict_tensor(model, x)[0].to(torch.uint8).cpu().numpy()
        colours = cv2.resize(colours, (w, h), interpolation=cv2.INTER_NEAREST)
        result = cv2.addWeighted(original, 0.5, colours, 0.5, 0.0)[..., ::-1]
        if save_result:
            os.makedirs(self.cfg.decode.test_results, exist_ok=True)
            cv2.imwrite(os.path.join(self.cfg.decode.test_results, os.path.basename(image_path).split(".")[0] + "@cvx.jpg"), result)
            return None
        return result
