"""CPU oracle (test infrastructure only; see yolov8_ref.py / nms_ref.py headers)."""
