set -o pipefail
export CVX_LIB=build/libcvx_tuning.so
for t in 4 6 2 7; do echo "== CVX_GEMM_TILE (variant) = $t"; CVX_GEMM_TILE=$t timeout -k 10 120 python tools/gemm_debug.py 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-60,100-400 || exit 1; done
