"""``Yolo7Trainer`` -- registered as ``trainer_yolo7`` like the reference's (core/trainer/yolo7_train.py), so that
``export_from_registry("yolo7")`` resolves.  The MI355X engine runs YOLOv7 for inference only this round; training raises."""
from computervision.pytorch_amd import _lib as L
from registry import trainer_registry


@trainer_registry("yolo7")
class Yolo7Trainer:
    def __init__(self, cfg, device):
        self.cfg, self.device = cfg, device

    def train(self):
        raise L.CvxError("YOLOv7 training is not built on the MI355X engine yet (inference + decode only); see DESIGN.md")
