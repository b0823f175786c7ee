set -o pipefail
export CVX_LIB=build/libcvx_tuning.so
for wl in "--workload yolov8_train --model s" "--workload yolov7_train" "--workload centernet_train" "--workload deeplab_train" "--workload yolov8_train"; do
for a in "CVX_NO_PHASE_GEMM=1" "CVX_PHASE_GEMM=1"; do echo "$a $wl: $(env $a timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $wl 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])')"; done; done
