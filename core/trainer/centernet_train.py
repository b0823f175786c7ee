"""``CenterNetTrainer`` -- registered as ``trainer_centernet`` like the reference's (core/trainer/centernet_train.py), so that
``export_from_registry("centernet")`` resolves.  The MI355X engine runs CenterNet for inference only this round; training raises."""
from computervision.pytorch_amd import _lib as L
from registry import trainer_registry


@trainer_registry("centernet")
class CenterNetTrainer:
    def __init__(self, cfg, device):
        self.cfg, self.device = cfg, device

    def train(self):
        raise L.CvxError("CenterNet training is not built on the MI355X engine yet (inference + decode only); see DESIGN.md")
