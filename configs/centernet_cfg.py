"""CenterNet config -- the reference's attribute bag (configs/centernet_cfg.py:5-95), field for field."""
from types import SimpleNamespace

from configs.dataset_cfg import VOC_CFG
from registry import config_registry


class _Group(SimpleNamespace):
    pass


@config_registry("centernet")
class CenternetConfig:
    def __init__(self):
        self.arch = _Group(input_size=(3, 384, 384), downsampling_ratio=4)                    # (reference :17-22)
        self.dataset = _Group(num_classes=VOC_CFG["num_classes"], dataset_name=VOC_CFG["name"])  # (:24-30)
        self.train = _Group(resume_training="", last_epoch=-1, epoch=100, batch_size=16, initial_lr=1e-3, warmup_iters=0, milestones=[],
                            gamma=0.1, pretrained=False, pretrained_weights="", save_interval=1, eval_interval=0, save_path="saves",
                            tensorboard_on=True, mixed_precision=True, num_workers=0, max_num_boxes=30)   # (:32-63)
        self.loss = _Group(hm_weight=1.0, wh_weight=0.1, off_weight=1.0)                       # (:65-70)
        self.optimizer = _Group(name="Adam")
        self.log = _Group(root="log", print_interval=50)
        self.decode = _Group(test_results="result", max_boxes_per_img=100, letterbox_image=True, score_threshold=0.1, use_nms=True,
                             nms_threshold=0.5)                                                  # (:85-95)
