set -o pipefail
CVX_LIB=build/libcvx_tuning.so timeout -k 10 120 python tools/gemm_debug.py 2>&1 | grep -v amdgpu.ids | tail -4
