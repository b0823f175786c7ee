set -o pipefail
export CVX_LIB=build/libcvx_tuning.so
for wl in "--workload ssd" "--workload ssd_train"; do
for a in "CVX_NO_STEM7=1" "CVX_STEM7=1" "CVX_NO_STEM7=1" "CVX_STEM7=1"; do echo "$a $wl: $(env $a timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $wl 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])')"; done; done
