"""YOLOv8 detection config -- the reference's "flag system" for this model.

Field-for-field the attribute bag of the reference (configs/yolo8_det_cfg.py:5-93): a plain
object with nested ``arch / dataset / train / loss / optimizer / log / decode`` groups, built with
no arguments by ``builder.export_from_registry``.  Extra, engine-only knobs live in
``cfg.engine`` and default to values that leave the reference's behaviour unchanged.
"""
from types import SimpleNamespace

from configs.dataset_cfg import COCO_CFG
from registry import config_registry


class _Group(SimpleNamespace):
    """Attribute bag; a class (not a dict) so ``cfg.train.batch_size`` style access works."""


@config_registry("yolo8_det")
class Yolo8DetConfig:
    def __init__(self):
        # model scale n/s/m/l/x and (C, H, W) network input        (reference :17-22)
        self.arch = _Group(model_type="n", input_size=(3, 640, 640))
        # dataset: class count and "voc"/"coco"                     (reference :24-30)
        self.dataset = _Group(num_classes=COCO_CFG["num_classes"], dataset_name=COCO_CFG["name"])
        # training schedule                                         (reference :32-63)
        self.train = _Group(
            resume_training="",      # checkpoint to resume / test from; "" = start at epoch 0
            last_epoch=-1,           # epoch of that checkpoint; -1 = fresh run
            epoch=100,
            batch_size=8,
            initial_lr=1e-3,
            warmup_iters=0,
            milestones=[],
            gamma=0.1,
            pretrained=False,
            pretrained_weights="",
            save_interval=10,        # epochs between checkpoints
            eval_interval=0,         # epochs between validation passes (0 = never)
            save_path="saves",
            tensorboard_on=True,
            mixed_precision=True,    # fp16 compute; the MI355X engine always computes in fp16/fp32-acc
            num_workers=0,
        )
        # loss gains                                                (reference :65-70)
        self.loss = _Group(box=7.5, cls=0.5, dfl=1.5)
        self.optimizer = _Group(name="Adam")                       # (reference :72-75)
        self.log = _Group(root="log", print_interval=50)           # (reference :77-83)
        # decode / NMS                                              (reference :85-92)
        self.decode = _Group(
            test_results="result",
            letterbox_image=True,
            conf_threshold=0.25,
            nms_threshold=0.7,
            max_det=300,
        )
        # --- MI355X engine knobs (new; not in the reference) ---------------------------------
        self.engine = _Group(
            loss_scale=1024.0,       # static fp16 gradient scale (bench.py, hipGraph replay)
            dynamic_loss_scale=True,  # Yolo8Trainer: GradScaler policy (skip non-finite steps, backoff/growth), see train.DynamicLossScale
            init_loss_scale=65536.0,  # GradScaler's initial scale
            graph_capture=False,     # Yolo8Trainer: replay the step as a hipGraph (needs a fixed target count per batch)
            allreduce_buckets=5,     # op ranges of the overlapped gradient exchange: RCCL all-reduces per step, issued while backward runs
        )
