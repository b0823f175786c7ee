set -o pipefail
export CVX_LIB=build/libcvx_tuning.so
for a in "CVX_GEMM_GFMIN=1 CVX_GEMM_KMIN=128" "CVX_GEMM_GFMIN=0 CVX_GEMM_KMIN=128" "CVX_GEMM_GFMIN=0 CVX_GEMM_KMIN=64" "CVX_GEMM_GFMIN=1 CVX_GEMM_KMIN=64" "CVX_GEMM_GFMIN=0 CVX_GEMM_KMIN=128 CVX_GEMM_MMIN=8192"; do echo "== $a"; env $a timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["forward_eval"]["ms_per_batch"])' || exit 1; done
