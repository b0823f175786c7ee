"""DeepLabv3+ config -- the reference's attribute bag (configs/deeplabv3plus_cfg.py:5-95), field for field."""
from types import SimpleNamespace

from configs.dataset_cfg import VOC_CFG
from registry import config_registry


class _Group(SimpleNamespace):
    pass


@config_registry("deeplabv3plus")
class DeeplabV3PlusConfig:
    def __init__(self):
        self.arch = _Group(backbone_name="resnet101", backbone_pretrained=False, input_size=(3, 513, 513), crop_size=(513, 513),
                           output_stride=16)                                                                  # (reference :17-25)
        self.dataset = _Group(num_classes=VOC_CFG["num_classes"] + 1, dataset_name=VOC_CFG["name"], root=VOC_CFG["root"])  # (:27-35)
        self.train = _Group(resume_training="", last_epoch=-1, epoch=100, batch_size=16, initial_lr=1e-3, warmup_iters=0, milestones=[],
                            gamma=0.1, pretrained=False, pretrained_weights="", save_interval=10, eval_interval=5, save_path="saves",
                            tensorboard_on=True, mixed_precision=True, num_workers=0)                          # (:37-68)
        self.loss = _Group(loss_type="focal")
        self.optimizer = _Group(name="Adam")
        self.log = _Group(root="log", print_interval=50)
        self.decode = _Group(test_results="result")
