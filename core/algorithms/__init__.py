"""Algorithm classes; importing registers them as ``model_<name>`` (reference core/algorithms/__init__.py)."""
from .yolo_v8 import YOLOv8  # noqa: F401
from .centernet import CenterNetA  # noqa: F401
from .segmentation_2d import DeeplabV3PlusA  # noqa: F401
from .yolo_v7 import YOLOv7  # noqa: F401
from .ssd import Ssd  # noqa: F401
