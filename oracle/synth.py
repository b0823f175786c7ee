"""Seeded synthetic inputs: the generators live in the package (bench.py uses them without touching ``oracle/``);
this module re-exports them for the oracle-side scripts and the tests."""
from computervision.pytorch_amd.synth import images, nms_pred, nms_pred_borderline, targets  # noqa: F401
