"""MI355X-native (gfx950) detection engine: the YOLOv8 hot path of ComputerVision.pytorch.

Importable as ``computervision.pytorch_amd`` (the repo-root ``computervision`` package maps the dotted
name onto this directory).  Everything compute-bearing lives in ``csrc/`` (hand-written HIP) behind the
C ABI of ``include/cvx_engine.h``; this package is the Python mirror of the reference's model /
loss / optimiser interface for that path.  There is no CPU or eager fallback.
"""
from . import _lib
from ._lib import CvxError, LIB_PATH

__all__ = ["CvxError", "LIB_PATH", "_lib"]
