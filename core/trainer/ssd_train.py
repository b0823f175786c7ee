"""``SsdTrainer`` -- registered as ``trainer_ssd`` like the reference's (core/trainer/ssd_train.py), so that
``export_from_registry("ssd")`` resolves.  The network's forward + backward run on the MI355X engine (``SSD300VGG`` in training mode); the loss
(MultiBoxLossV2) and target encoding, and therefore this trainer's loop, are not built: ``train()`` raises."""
from computervision.pytorch_amd import _lib as L
from registry import trainer_registry


@trainer_registry("ssd")
class SsdTrainer:
    def __init__(self, cfg, device):
        self.cfg, self.device = cfg, device

    def train(self):
        raise L.CvxError("the SSD training LOOP is not built (MultiBoxLossV2 / target encoding have no HIP kernels); the network's forward + backward are: see DESIGN.md 7d")
