#!/bin/bash
# confining the weight-gradient stream to n CUs (CU-mask stream) against the default (lowest-priority stream on all CUs), every train step
run() { W=$1; shift; echo "== $W $*"; env "$@" CVX_LIB=build/libcvx_tuning.so python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items()})"; }
for W in ssd_train deeplab_train yolov7_train centernet_train yolov8_train; do
  run $W CVX_X=0
  run $W CVX_SIDE_CUS=64
  run $W CVX_SIDE_CUS=96
  run $W CVX_SIDE_CUS=128
  run $W CVX_SIDE_CUS=192
done
