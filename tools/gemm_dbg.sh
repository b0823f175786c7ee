set -o pipefail
export CVX_LIB=build/libcvx_tuning.so
for m in 1 3; do echo "== GEMM_DEBUG_MODE=$m"; GEMM_DEBUG_MODE=$m timeout -k 10 120 python tools/gemm_debug.py 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-30,100-400 || exit 1; done
