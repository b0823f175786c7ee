"""``SsdTrainer`` -- registered as ``trainer_ssd`` like the reference's (core/trainer/ssd_train.py), so that
``export_from_registry("ssd")`` resolves.  The MI355X engine runs SSD for inference only this round; training raises."""
from computervision.pytorch_amd import _lib as L
from registry import trainer_registry


@trainer_registry("ssd")
class SsdTrainer:
    def __init__(self, cfg, device):
        self.cfg, self.device = cfg, device

    def train(self):
        raise L.CvxError("SSD training is not built on the MI355X engine yet (inference + decode only); see DESIGN.md")
