#!/bin/bash
# A/B of knobs on the YOLOv8-n train step (tuning library)
run() { echo "== $*"; env "$@" CVX_LIB=build/libcvx_tuning.so python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run CVX_X=0
run CVX_BN_KB=16
run CVX_BN_KB=64
run CVX_WGRAD_BATCH=2
run CVX_WGRAD_BATCH=4
run CVX_WH_BLOCKS=256
run CVX_SLAB_TAIL=3
run CVX_X=0
