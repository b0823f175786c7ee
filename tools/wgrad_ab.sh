# A/B of the GEMM-shaped weight-gradient kernel on the train steps (tuning build)
set -o pipefail
export CVX_LIB=build/libcvx_tuning.so
run() {
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  out=$(env "${envs[@]}" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1)
  echo "$label | $* | $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["unit"], d["ms_per_step"], "ms")')"
}
for wl in "--workload deeplab_train" "--workload ssd_train" "--workload yolov8_train --model s" "--workload yolov8_train" "--workload centernet_train" "--workload yolov7_train"; do
  run "generic wgrad    " CVX_NO_WGRAD_GEMM=1 -- $wl || exit 1
  run "GEMM-shaped wgrad" CVX_WGG_CMIN=128 -- $wl || exit 1
  if [ "$1" != "quick" ]; then
    run "  ... from 64 ch " CVX_WGG_CMIN=64 -- $wl || exit 1
    run "  ... 128x128    " CVX_WGG_TILE=2 -- $wl || exit 1
  fi
done
