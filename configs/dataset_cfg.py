"""Dataset descriptors read by the config classes (reference configs/dataset_cfg.py:1-117).

Same module-level names and dict keys as the reference: ``ALL_DATASETS_ROOT``, ``VOC_CFG`` and
``COCO_CFG`` with ``root / name / num_classes / classes``.  The class lists are the public
VOC-2012 and COCO-2017 category names in dataset order (multi-word names keep their spaces).
"""

ALL_DATASETS_ROOT = "../../Datasets/"


def _names(block: str):
    return [w.replace("_", " ") for w in block.split()]


_VOC_NAMES = _names("""
    person bird cat cow dog horse sheep aeroplane bicycle boat bus car motorbike train bottle chair
    diningtable pottedplant sofa tvmonitor
""")

_COCO_NAMES = _names("""
    person bicycle car motorcycle airplane bus train truck boat traffic_light fire_hydrant stop_sign
    parking_meter bench bird cat dog horse sheep cow elephant bear zebra giraffe backpack umbrella
    handbag tie suitcase frisbee skis snowboard sports_ball kite baseball_bat baseball_glove skateboard
    surfboard tennis_racket bottle wine_glass cup fork knife spoon bowl banana apple sandwich orange
    broccoli carrot hot_dog pizza donut cake chair couch potted_plant bed dining_table toilet tv laptop
    mouse remote keyboard cell_phone microwave oven toaster sink refrigerator book clock vase scissors
    teddy_bear hair_drier toothbrush
""")

VOC_CFG = {
    "root": ALL_DATASETS_ROOT + "VOCdevkit/VOC2012/",
    "name": "voc",
    "num_classes": len(_VOC_NAMES),
    "classes": _VOC_NAMES,
}

COCO_CFG = {
    "root": ALL_DATASETS_ROOT + "coco",
    "name": "coco",
    "num_classes": len(_COCO_NAMES),
    "classes": _COCO_NAMES,
}
