"""TEST INFRASTRUCTURE (oracle side): loss curves of 50 oracle train steps (fp32, and with the engine's fp16 rounding points emulated) on one
repeated 128x128 synthetic batch from the seed-0 initialisation -> tests/golden/yolov8n_traj_128.npz.  The oracle's train_step is pinned
against the real reference in oracle/make_golden.py (loss, every gradient, two Adam steps); this script only runs it longer.
    python oracle/make_traj_fixture.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from computervision.pytorch_amd import synth  # noqa: E402
from oracle import yolov8_ref as O  # noqa: E402

STEPS = 50


def curve(emulate):
    x, batch = synth.images(2, 128, 128, seed=1), synth.targets(2, seed=2)
    sd, state = O.init_state_dict("n", 80, seed=0), {}
    O.FP16_STORAGE[0] = emulate
    try:
        return np.array([float(O.train_step(sd, x, batch, state)[0]) for _ in range(STEPS)])
    finally:
        O.FP16_STORAGE[0] = False


if __name__ == "__main__":
    np.savez(os.path.join(ROOT, "tests", "golden", "yolov8n_traj_128.npz"), fp32=curve(False), fp16_emulation=curve(True), steps=STEPS,
             note="oracle/make_traj_fixture.py: images seed 1, targets seed 2, batch 2, 128x128, Adam lr 1e-3")
    print("written")
