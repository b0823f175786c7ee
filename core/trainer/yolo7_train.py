"""``Yolo7Trainer`` -- registered as ``trainer_yolo7`` like the reference's (core/trainer/yolo7_train.py), so that
``export_from_registry("yolo7")`` resolves.  The network's forward + backward run on the MI355X engine (``Yolo7L`` in training mode);
the loss (Yolo7Loss, SimOTA) and therefore this trainer's loop are not built: ``train()`` raises."""
from computervision.pytorch_amd import _lib as L
from registry import trainer_registry


@trainer_registry("yolo7")
class Yolo7Trainer:
    def __init__(self, cfg, device):
        self.cfg, self.device = cfg, device

    def train(self):
        raise L.CvxError("the YOLOv7 training LOOP is not built (Yolo7Loss has no HIP kernel); the network's forward + backward are: see DESIGN.md 7c")
