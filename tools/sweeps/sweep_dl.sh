#!/bin/bash
# A/B of the wgrad knobs on the DeepLab train step (tuning library)
run() { echo "== $*"; env "$@" CVX_LIB=build/libcvx_tuning.so python bench.py --workload deeplab_train --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items()})"; }
run CVX_SLAB_MB=64 CVX_WH_MAX_C=65535
run CVX_SLAB_MB=128 CVX_WH_MAX_C=65535
run CVX_SLAB_MB=64 CVX_WH_MAX_C=16383
run CVX_SLAB_MB=64 CVX_WH_MAX_C=65535 CVX_WH_SLAB_MB=64 CVX_WH_BLOCKS=256
run CVX_SLAB_MB=64 CVX_WH_MAX_C=65535 CVX_WGRAD_BLOCKS=1024
run CVX_SLAB_MB=64 CVX_WH_MAX_C=65535 CVX_NSPLIT_CAP=64
CVX_SLAB_MB=64 CVX_WH_MAX_C=65535 CVX_LIB=build/libcvx_tuning.so python tools/op_profile.py 3 deeplab > gpurun_out/dl_op_profile2.txt 2>&1
