#!/bin/bash
# A/B of conv dispatch knobs on the DeepLab train step (tuning library)
run() { echo "== $*"; env "$@" CVX_LIB=build/libcvx_tuning.so python bench.py --workload deeplab_train --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items()})"; }
run CVX_X=0
run CVX_NO_PW=1
run CVX_PW_WKB=32
run CVX_PW_WKB=128
run CVX_HEAVY_W=100000000
run CVX_T64_HEAVY=100000000
run CVX_PW_OCC=2
run CVX_BN_KB=64
