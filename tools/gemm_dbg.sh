set -o pipefail
export CVX_LIB=build/libcvx_tuning.so
for wl in "--workload deeplab_train" "--workload ssd_train" "--workload yolov8_train --model s" "--workload yolov7_train"; do
for a in "CVX_WGG_BLOCKS_BIG=256 CVX_WGG_BLOCKS=512" "CVX_WGG_BLOCKS_BIG=512 CVX_WGG_BLOCKS=512" "CVX_WGG_BLOCKS_BIG=256 CVX_WGG_BLOCKS=1024" "CVX_WGG_BLOCKS_BIG=128 CVX_WGG_BLOCKS=256" "CVX_WGG_BLOCKS_BIG=512 CVX_WGG_BLOCKS=1024"; do echo "$a $wl: $(env $a timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $wl 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"; done; done
