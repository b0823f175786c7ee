"""Study: what does storing the raw conv output in fp16 cost on the logits, layer by layer?

Runs the oracle's train-mode forward in fp32 (= the reference's CPU path) and with the engine's fp16 rounding points emulated
(O.FP16_STORAGE), with the raw conv output additionally rounded to fp16 for a chosen set of layers (O.FP16_RAW_LAYERS), and prints the
relative L2 error of the three Detect levels against the fp32 run.  Test infrastructure (CPU only):  python oracle/fp16_raw_study.py
"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import synth  # noqa: E402
from oracle import yolov8_ref as O  # noqa: E402


def bn_layers(sd):
    return [k[:-len(".bn.weight")] for k in sd if k.endswith(".bn.weight")]


BIG = lambda p: (p.startswith(("model.1", "model.2.", "model.3", "model.4.", "model.15.")) and not p.startswith(("model.10", "model.12", "model.16", "model.18", "model.19"))) \
    or p.startswith(("model.22.cv2.0.", "model.22.cv3.0."))


def rel(a, b):
    return float((a - b).norm() / b.norm())


def run(sd, x, raw):
    O.FP16_STORAGE[0] = True
    O.FP16_RAW_LAYERS = set(raw)
    out = O.forward(copy.deepcopy(sd), x, "n", 80, True)
    O.FP16_STORAGE[0] = False
    O.FP16_RAW_LAYERS = None
    return out


def main():
    torch.set_num_threads(8)
    sd = O.init_state_dict("n", 80, seed=0)
    layers = [p for p in bn_layers(sd) if p != "model.0"]
    big = [p for p in layers if BIG(p)]
    print("layers with >= 80x80 outputs at 640x640:", big)
    cases = [("128x128 b2", synth.images(2, 128, 128, seed=1)), ("96x160 b2", synth.images(2, 96, 160, seed=3)), ("320x320 b2", synth.images(2, 320, 320, seed=1)),
             ("640x640 b1", synth.images(1, 640, 640, seed=1))]
    neck = [p for p in layers if p.startswith(("model.12.", "model.15.", "model.16", "model.18.", "model.19", "model.21."))]
    head = [p for p in layers if p.startswith("model.22.")]
    groups = {"none (rounds 2-4)": [], "neck + head": neck + head, "head": head, "neck": neck, "neck+head+SPPF(9)": neck + head + [p for p in layers if p.startswith("model.9.")],
              "neck+head+8+9": neck + head + [p for p in layers if p.startswith(("model.9.", "model.8.", "model.7"))]}
    groups_old = {"large layers": big, "all layers": layers,
              "large w/o model.1": [p for p in big if p != "model.1"], "large w/o Detect": [p for p in big if not p.startswith("model.22")],
              "only model.1+model.2": [p for p in big if p.startswith(("model.1", "model.2."))], "only Detect P3": [p for p in big if p.startswith("model.22")],
              "only model.4": [p for p in big if p.startswith("model.4.")], "only model.15": [p for p in big if p.startswith("model.15.")]}
    for name, x in cases:
        with torch.no_grad():
            ref = O.forward(copy.deepcopy(sd), x, "n", 80, True)
            print(f"--- {name}")
            for g, raw in groups.items():
                out = run(sd, x, raw)
                print(f"  {g:24s} P3 {rel(out[0], ref[0]):.2e}  P4 {rel(out[1], ref[1]):.2e}  P5 {rel(out[2], ref[2]):.2e}", flush=True)


if __name__ == "__main__":
    main()
