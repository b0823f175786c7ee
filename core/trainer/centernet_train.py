"""``CenterNetTrainer`` -- registered as ``trainer_centernet`` like the reference's (core/trainer/centernet_train.py), so that
``export_from_registry("centernet")`` resolves.  The network's forward + backward run on the MI355X engine (``CenterNetDLA34`` in
training mode); the loss (CombinedLoss) and target generation, and therefore this trainer's loop, are not built: ``train()`` raises."""
from computervision.pytorch_amd import _lib as L
from registry import trainer_registry


@trainer_registry("centernet")
class CenterNetTrainer:
    def __init__(self, cfg, device):
        self.cfg, self.device = cfg, device

    def train(self):
        raise L.CvxError("the CenterNet training LOOP is not built (CombinedLoss / target drawing have no HIP kernels); the network's forward + backward are: see DESIGN.md 7")
