export CVX_LIB=$(pwd)/build/libcvx_tuning.so
for v in 0 1 2; do CVX_TUNE_SKIP_WGRAD=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('skip_wgrad $v', d['ms_per_step'], {k: v['ms_per_step'] for k, v in d.get('kernel_classes', {}).items()})"; done
