# same-box A/B of two builds of the library:   bash tools/sweeps/lib_ab.sh build/libcvx_old.so   (against the in-tree release library)
ALT=$(pwd)/$1
for rep in 1 2 3; do
  for lib in new old; do
    if [ $lib = old ]; then export CVX_LIB=$ALT; else unset CVX_LIB; fi
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib', d['ms_per_step'], {k: v['ms_per_step'] for k, v in d.get('kernel_classes', {}).items() if k.startswith('bn')})"
  done
done
for lib in new old; do
  if [ $lib = old ]; then export CVX_LIB=$ALT; else unset CVX_LIB; fi
  python bench.py --workload yolov8_eval --steps 50 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('eval $lib', d['ms_per_step'])"
done
