set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/wg
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/wg -- python3 tools/micro/wgrad_shapes.py > gpurun_out/wg.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/wg/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'wgrad' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
i=0
while i<len(rows):
    grp=rows[i:i+3]; best=min(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in grp)/1e3
    print(grp[0]['Kernel_Name'][:70], f"{best:8.1f} us")
    i+=3
PY
