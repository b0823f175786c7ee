#!/bin/bash
# A/B of two builds on one box: the release library against build/libcvx_tuning.so, train steps (ms per step and the weight-gradient class)
mkdir -p gpurun_out/ab_lib
for rep in 1 2; do
  for lib in release tuning; do
    if [ $lib = tuning ]; then export CVX_LIB=build/libcvx_tuning.so; else unset CVX_LIB; fi
    for wl in yolov8_train centernet_train ssd_train yolov7_train; do
      timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '"metric"' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d.get('kernel_classes',{}).get('conv_wgrad',{})
print('$lib', '$wl', d['ms_per_step'], 'wgrad', k.get('ms_per_step'))" | tee -a gpurun_out/ab_lib/ab.txt
    done
  done
done
