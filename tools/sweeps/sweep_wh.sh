#!/bin/bash
# CVX_WH_BLOCKS / CVX_WH_SLAB_MB (split count of the register-tile weight-gradient kernel) on every train step (tuning library)
run() { W=$1; shift; echo "== $W $*"; env "$@" CVX_LIB=build/libcvx_tuning.so python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items() if k in ('conv_wgrad','slab_reduce')})"; }
for W in yolov8_train ssd_train yolov7_train centernet_train deeplab_train; do
  run $W CVX_X=0
  run $W CVX_WH_BLOCKS=256
  run $W CVX_WH_BLOCKS=512
  run $W CVX_WH_BLOCKS=1024
  run $W CVX_WH_BLOCKS=512 CVX_WH_SLAB_MB=64
  run $W CVX_WH_BLOCKS=1024 CVX_WH_SLAB_MB=64
done
