#!/bin/bash
# the work-based split rule of the register-tile weight-gradient kernel (CVX_WH_BIG_GF) and the generic kernel's block target, every train step
run() { W=$1; shift; echo "== $W $*"; env "$@" CVX_LIB=build/libcvx_tuning.so python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items() if k in ('conv_wgrad','slab_reduce')})"; }
for W in yolov8_train ssd_train yolov7_train centernet_train deeplab_train; do
  run $W CVX_X=0
  run $W CVX_WH_BIG_GF=8
  run $W CVX_WH_BIG_GF=16
  run $W CVX_WH_BIG_GF=64
  run $W CVX_WGRAD_BLOCKS=4096
  run $W CVX_SLAB_MB=16
done
