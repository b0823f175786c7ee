# A/B of the GEMM-shaped conv kernel's dispatch gates on whole models (tuning build): value / ms_per_step of bench.py per workload
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
for wl in "--workload ssd" "--workload deeplab_train" "--workload centernet" "--workload yolov8_train --model s" "--workload yolov8_train" "--workload yolov8_eval" "--workload deeplab" "--workload ssd_train"; do
  run "no gemm      " CVX_NO_GEMM=1 -- $wl || exit 1
  run "gemm c>=128  " CVX_GEMM_CMIN=128 -- $wl || exit 1
  run "gemm c>=64   " CVX_GEMM_CMIN=64 -- $wl || exit 1
done
