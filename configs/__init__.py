"""Config classes; importing this package registers them (reference configs/__init__.py:1-5).

Only the configs whose MI355X path exists are imported here; the remaining reference models
(ssd) is a SURVEY.md section 8(f) "next" rows.
"""
from .yolo8_det_cfg import Yolo8DetConfig  # noqa: F401
from .centernet_cfg import CenternetConfig  # noqa: F401
from .deeplabv3plus_cfg import DeeplabV3PlusConfig  # noqa: F401
from .yolo7_cfg import Yolo7Config  # noqa: F401
