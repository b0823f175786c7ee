#!/bin/bash
# A/B sweep on the GPU box with the tuning build:  tools/sweep.sh VAR v1 v2 ...   (prints ms/step per value; same box, same run)
VAR=$1; shift
for v in "$@"; do
  r=$(env CVX_LIB=build/libcvx_tuning.so $VAR=$v timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['forward_eval']['ms_per_batch'])")
  echo "$VAR=$v ms/step,eval_ms: $r"
done
