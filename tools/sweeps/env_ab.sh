# same-box A/B of one tuning-library switch:   bash tools/sweeps/env_ab.sh CVX_NO_BS_FUSE   (runs with the variable unset, then =1, three times)
export CVX_LIB=$(pwd)/build/libcvx_tuning.so
VAR=$1
for rep in 1 2 3; do
  for val in "" 1; do
    if [ -z "$val" ]; then unset $VAR; else export $VAR=$val; fi
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$VAR=$val', d['ms_per_step'], {k: (v['ms_per_step'], v['launches_per_step']) for k, v in d.get('kernel_classes', {}).items() if k.startswith('bn') or k == 'conv_dgrad'})"
  done
done
