# same-box A/B of several builds of the library, interleaved:   REPS=3 bash tools/sweeps/libs_ab.sh build/libcvx_a.so build/libcvx_b.so ...
REPS=${REPS:-2}
STEPS=${STEPS:-30}
for rep in $(seq $REPS); do
  for lib in "$@"; do
    CVX_LIB=$(pwd)/$lib python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k=d.get('kernel_classes',{})
print('%-28s %.4f ms  eval %.4f  ' % ('$lib', d['ms_per_step'], d['forward_eval']['ms_per_batch']) + '  '.join('%s %.3f' % (n, k[n]['ms_per_step']) for n in ('conv_wgrad','slab_reduce','bn_silu_fwd','bn_silu_bwd','conv_fwd','conv_dgrad') if n in k), flush=True)"
  done
done
