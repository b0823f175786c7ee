"""Trainer classes; importing registers them as ``trainer_<name>`` (reference core/trainer/__init__.py)."""
from .yolo8_train import Yolo8Trainer  # noqa: F401
from .centernet_train import CenterNetTrainer  # noqa: F401
from .segmentation_trainer import DeeplabV3PlusTrainer  # noqa: F401
from .yolo7_train import Yolo7Trainer  # noqa: F401
from .ssd_train import SsdTrainer  # noqa: F401
