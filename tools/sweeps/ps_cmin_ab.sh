export CVX_LIB=$(pwd)/build/libcvx_tuning.so
for rep in 1 2 3; do for v in 64 128 1024; do CVX_PS_CMIN=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ps_cmin $v', d['ms_per_step'], d['kernel_classes']['conv_dgrad'])"; done; done
CVX_PS_CMIN=64 python tools/op_profile.py 5 2>&1 | grep "conv_dgrad.*k3s2" | cut -c1-120
