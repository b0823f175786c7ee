"""SSD config -- the reference's attribute bag (configs/ssd_cfg.py:5-150), field for field (input size 300)."""
from types import SimpleNamespace

from configs.dataset_cfg import VOC_CFG
from registry import config_registry


class _Group(SimpleNamespace):
    pass


@config_registry("ssd")
class SsdConfig:
    def __init__(self):
        third = 1.0 / 3
        self.arch = _Group(backbone="vgg", input_size=(3, 300, 300),
                           aspect_ratios=[[1, 2, 0.5], [1, 2, 0.5, 3, third], [1, 2, 0.5, 3, third], [1, 2, 0.5, 3, third], [1, 2, 0.5], [1, 2, 0.5]],
                           feature_channels=[512, 1024, 512, 256, 256, 256], feature_shapes=[38, 19, 10, 5, 3, 1],
                           anchor_sizes=[30, 60, 111, 162, 213, 264, 315])                                # (reference :8-75)
        self.dataset = _Group(num_classes=VOC_CFG["num_classes"], dataset_name=VOC_CFG["name"])
        self.train = _Group(resume_training="", last_epoch=-1, epoch=100, batch_size=16, initial_lr=1e-3, warmup_iters=1000, milestones=[],
                            gamma=0.1, pretrained=False, pretrained_weights="", save_interval=1, eval_interval=0, save_path="saves",
                            tensorboard_on=True, mixed_precision=True, num_workers=0)
        self.loss = _Group(alpha=0.25, gamma=2.0, overlap_threshold=0.5, neg_pos=3, variance=[0.1, 0.2])
        self.optimizer = _Group(name="Adam")
        self.log = _Group(root="log", print_interval=50)
        self.decode = _Group(test_results="result", letterbox_image=True, nms_threshold=0.5, confidence_threshold=0.7)
