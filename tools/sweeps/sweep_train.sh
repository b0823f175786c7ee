#!/bin/bash
# A/B of the weight-gradient dispatch knobs on a train step of the tuning library:  tools/sweeps/sweep_train.sh ssd_train|yolov7_train|centernet_train
W=${1:-ssd_train}
run() { echo "== $W $*"; env "$@" CVX_LIB=build/libcvx_tuning.so python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items()})"; }
run CVX_X=0
run CVX_TUNE_SKIP_WGRAD=1
run CVX_WH_MAX_C=4095
run CVX_WH_MAX_C=16383
run CVX_WH_MAX_C=32767
run CVX_WGRAD_WIDE_MIN=131072
run CVX_WGRAD_WIDE_MIN=65536
run CVX_WH_MAX_C=16383 CVX_WGRAD_WIDE_MIN=131072
run CVX_WH_BLOCKS=512
run CVX_WH_BLOCKS=2048
run CVX_WGRAD_BATCH=1
run CVX_WGRAD_BATCH=8
run CVX_X=0
