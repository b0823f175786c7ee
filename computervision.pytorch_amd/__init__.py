"""MI355X-native (gfx950) detection engine: the YOLOv8 hot path of ComputerVision.pytorch.

Importable as ``computervision.pytorch_amd`` (the repo-root ``computervision`` package maps the dotted
name onto this directory).  Everything compute-bearing lives in ``csrc/`` (hand-written HIP) behind the
C ABI of ``include/cvx_engine.h``; this package is the Python mirror of the reference's model /
loss / optimiser interface for that path.  There is no CPU or eager fallback.
"""
import os as _os
import sys as _sys
import warnings as _warnings


def _hardware_queue_budget():
    """A process that puts more than four hardware queues to work pays 2.2-2.5x on every step on this runtime.  With
    GPU_MAX_HW_QUEUES=1 the runtime maps all streams of one priority onto one queue; the engine's three streams have three priorities, so
    it loses nothing and the process becomes immune to whatever streams torch or the caller add.  Set here -- for trainers and tests as
    for bench.py -- when the HIP runtime has not been initialised yet and the process is not one rank of several (there the collective's
    stream shares the main stream's priority: bench.py explains); an explicit setting in the environment always wins."""
    if "GPU_MAX_HW_QUEUES" in _os.environ or int(_os.environ.get("WORLD_SIZE", "1") or 1) > 1 or _os.environ.get("CVX_KEEP_HW_QUEUES"):
        return None        # (CVX_KEEP_HW_QUEUES=1: the opt-out for hosts that manage the variable themselves)
    torch = _sys.modules.get("torch")
    if torch is not None and getattr(torch, "cuda", None) is not None and torch.cuda.is_initialized():
        _warnings.warn("computervision.pytorch_amd imported after the HIP runtime was initialised: GPU_MAX_HW_QUEUES=1 cannot be applied any "
                       "more; export it before the process starts if other libraries create HIP streams (README: hardware queues)", stacklevel=2)
        return None
    _os.environ["GPU_MAX_HW_QUEUES"] = "1"
    return "1"


# what this import set GPU_MAX_HW_QUEUES to (None: left alone -- already set by the caller, a rank of a multi-GPU job, opted out, or too late)
HW_QUEUES_APPLIED = _hardware_queue_budget()

from . import _lib  # noqa: E402
from ._lib import CvxError, LIB_PATH  # noqa: E402

__all__ = ["CvxError", "LIB_PATH", "_lib"]
