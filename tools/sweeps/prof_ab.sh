set -e
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
export CVX_LIB=$ROOT/build/libcvx_tuning.so
export CVX_BN_FUSED=1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/bnf_prof_on -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/gpurun_out/bnf_prof_on.json 2> $ROOT/gpurun_out/bnf_prof_on.err
export CVX_BN_FUSED=0
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/bnf_prof_off -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/gpurun_out/bnf_prof_off.json 2> $ROOT/gpurun_out/bnf_prof_off.err
cd $ROOT
for d in on off; do cp $(ls gpurun_out/bnf_prof_$d/*/*_kernel_stats.csv | head -1) gpurun_out/bnf_stats_$d.csv; cp $(ls gpurun_out/bnf_prof_$d/*/*_kernel_trace.csv | head -1) gpurun_out/bnf_trace_$d.csv; rm -rf gpurun_out/bnf_prof_$d; done
