# same-box A/B of two release builds over the train workloads:  bash tools/sweeps/lib_ab_all.sh build/libcvx_r04.so [workloads...]
ALT=$(pwd)/$1; shift
WLS=${@:-"yolov8_train yolov8_eval deeplab_train ssd_train yolov7_train centernet_train"}
for wl in $WLS; do
  for rep in 1 2; do
    for lib in new old; do
      if [ $lib = old ]; then export CVX_LIB=$ALT; else unset CVX_LIB; fi
      python bench.py --workload $wl --steps 15 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-16s %-4s %.4f ms' % ('$wl', '$lib', d['ms_per_step']), flush=True)"
    done
  done
done
