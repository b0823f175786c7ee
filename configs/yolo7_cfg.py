"""YOLOv7 config -- the reference's attribute bag (configs/yolo7_cfg.py:5-96), field for field."""
from types import SimpleNamespace

from configs.dataset_cfg import VOC_CFG
from registry import config_registry


class _Group(SimpleNamespace):
    pass


@config_registry("yolo7")
class Yolo7Config:
    def __init__(self):
        self.arch = _Group(input_size=(3, 640, 640),
                           anchors=[12, 16, 19, 36, 40, 28, 36, 75, 76, 55, 72, 146, 142, 110, 192, 243, 459, 401],
                           anchors_mask=[[6, 7, 8], [3, 4, 5], [0, 1, 2]], phi="l")                       # (reference :17-27)
        self.dataset = _Group(dataset_name=VOC_CFG["name"], num_classes=VOC_CFG["num_classes"])             # (:29-34)
        self.train = _Group(resume_training="", last_epoch=-1, epoch=100, batch_size=4, initial_lr=1e-3, warmup_iters=0, milestones=[30, 60],
                            gamma=0.1, pretrained=True, pretrained_weights="saves/yolov7_weights.pth", save_interval=5, eval_interval=0,
                            save_path="saves", tensorboard_on=True, mixed_precision=True, num_workers=0, max_num_boxes=30)  # (:36-70)
        self.loss = _Group(ignore_threshold=0.5, label_smoothing=0)
        self.optimizer = _Group(name="Adam", scheduler_name="multi_step")
        self.log = _Group(root="log", print_interval=50)
        self.decode = _Group(test_results="result", letterbox_image=True, conf_threshold=0.5, nms_threshold=0.3)  # (:88-96)
