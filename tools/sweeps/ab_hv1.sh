#!/bin/bash
# A/B of CVX_HALO_HV1_MODE on the tuning build
export CVX_LIB=build/libcvx_tuning.so
mkdir -p gpurun_out/s13
for mode in 1 3 1 3; do
  export CVX_HALO_HV1_MODE=$mode
  for wl in yolov8_train yolov8_eval centernet yolov7 ssd deeplab; do
    extra="--steps 20 --warmup 3"
    timeout -k 10 150 python bench.py --workload $wl $extra --no-cpu-baseline 2>/dev/null | grep metric | python -c "
import sys,json
d=json.loads(sys.stdin.read()); fe=d.get('forward_eval',{}).get('ms_per_batch')
print('mode $mode', '$wl', d['ms_per_step'], fe if fe else '')" | tee -a gpurun_out/s13/ab2.txt
  done
done
