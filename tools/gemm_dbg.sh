set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CVX_LIB=build/libcvx_tuning.so
rm -rf gpurun_out/wg2
CVX_WH_MAX_C=16383 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/wg2 -- python3 tools/micro/wgrad_shapes.py > gpurun_out/wg2.log 2>&1
python - <<'PY'
import csv,glob,os
f=max(glob.glob('gpurun_out/wg2/*/*kernel_trace.csv'), key=os.path.getmtime)
rows=[r for r in csv.DictReader(open(f)) if 'wgrad' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
i=0
while i<len(rows):
    grp=rows[i:i+3]; best=min(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in grp)/1e3
    print(grp[0]['Kernel_Name'][:70], f"{best:8.1f} us")
    i+=3
PY
for wl in "--workload ssd_train" "--workload yolov8_train --model s" "--workload yolov8_train" "--workload centernet_train"; do
for a in "CVX_WH_MAX_C=65535" "CVX_WH_MAX_C=16383"; do echo "$a $wl: $(env $a timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $wl 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"; done; done
