# same-box A/B of environment settings on the TUNING library, interleaved, REPS times each:
#   bash tools/sweeps/ab.sh "base:CVX_NO_WGRAD_STREAM=1" "stream:" "big:CVX_WS_MIN_M=800000"
export CVX_LIB=$(pwd)/build/libcvx_tuning.so
REPS=${REPS:-2}
STEPS=${STEPS:-30}
for rep in $(seq $REPS); do
  for cfg in "$@"; do
    label=${cfg%%:*}
    envs=${cfg#*:}
    env $(echo $envs | tr ',' ' ') python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k=d.get('kernel_classes',{})
print('%-14s %.4f ms  eval %.4f  ' % ('$label', d['ms_per_step'], d['forward_eval']['ms_per_batch']) + '  '.join('%s %.3f' % (n, k[n]['ms_per_step']) for n in ('conv_wgrad','slab_reduce','bn_silu_fwd','bn_silu_bwd','conv_fwd','conv_dgrad') if n in k), flush=True)"
  done
done
