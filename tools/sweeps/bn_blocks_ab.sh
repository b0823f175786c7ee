# same-box A/B of the BN passes' block-count cap (tuning library, CVX_BN_BLOCKS; 0 = the fixed 32 KB blocks of round 3)
export CVX_LIB=$(pwd)/build/libcvx_tuning.so
for rep in 1 2; do
  for nb in 0 512 768 1024; do
    CVX_BN_BLOCKS=$nb python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('blocks $nb', d['ms_per_step'], {k: v['ms_per_step'] for k, v in d.get('kernel_classes', {}).items() if k.startswith('bn')})"
  done
done
for nb in 0 512; do CVX_BN_BLOCKS=$nb python bench.py --workload yolov8_eval --steps 50 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('eval blocks $nb', d['ms_per_step'])"; done
