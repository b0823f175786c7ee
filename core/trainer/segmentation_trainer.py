"""``DeeplabV3PlusTrainer`` -- registered as ``trainer_deeplabv3plus`` like the reference's (core/trainer/segmentation_trainer.py),
so that ``export_from_registry("deeplabv3plus")`` resolves.  The MI355X engine runs DeepLabv3+ for inference only this round."""
from computervision.pytorch_amd import _lib as L
from registry import trainer_registry


@trainer_registry("deeplabv3plus")
class DeeplabV3PlusTrainer:
    def __init__(self, cfg, device):
        self.cfg, self.device = cfg, device

    def train(self):
        raise L.CvxError("DeepLabv3+ training is not built on the MI355X engine yet (inference only); see DESIGN.md")
