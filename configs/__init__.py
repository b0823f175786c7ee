"""Config classes; importing this package registers them (reference configs/__init__.py:1-5).

Every whitelisted model of the reference has an MI355X path (YOLOv8: training + inference; the others: inference).
"""
from .yolo8_det_cfg import Yolo8DetConfig  # noqa: F401
from .centernet_cfg import CenternetConfig  # noqa: F401
from .deeplabv3plus_cfg import DeeplabV3PlusConfig  # noqa: F401
from .yolo7_cfg import Yolo7Config  # noqa: F401
from .ssd_cfg import SsdConfig  # noqa: F401
