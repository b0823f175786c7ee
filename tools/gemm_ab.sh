# A/B of the GEMM-shaped conv kernel's dispatch gates on whole models (tuning build): value / ms_per_step of bench.py per workload.
#   bash tools/gemm_ab.sh            all arms          bash tools/gemm_ab.sh quick     the shipped gates only
set -o pipefail
export CVX_LIB=build/libcvx_tuning.so
run() {  # label, env..., -- bench args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  out=$(env "${envs[@]}" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1)
  echo "$label | $* | $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["unit"], d["ms_per_step"], "ms")')"
}
for wl in "--workload ssd" "--workload deeplab_train" "--workload centernet" "--workload yolov8_train --model s" "--workload yolov8_train" "--workload deeplab" "--workload ssd_train" "--workload centernet_train" "--workload yolov7" "--workload yolov7_train"; do
  if [ "$1" != "quick" ]; then run "no gemm            " CVX_NO_GEMM=1 -- $wl || exit 1; fi
  run "gemm (shipped)     " CVX_GEMM_GFMIN=0 -- $wl || exit 1
  if [ "$1" != "quick" ]; then
    run "gemm K>=256, 2 GF  " CVX_GEMM_GFMIN=2 CVX_GEMM_KMIN=256 -- $wl || exit 1
    run "gemm c>=64         " CVX_GEMM_GFMIN=0 CVX_GEMM_KMIN=64 CVX_GEMM_CMIN=64 -- $wl || exit 1
  fi
done
