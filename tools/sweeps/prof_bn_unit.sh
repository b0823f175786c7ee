set -e
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
export CVX_LIB=$ROOT/build/libcvx_tuning.so
export CVX_BN_FUSED=1
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/bnu_on -- python $ROOT/tools/micro/bn_bwd_unit.py 10 > $ROOT/gpurun_out/bnu_on.txt 2>&1
export CVX_BN_FUSED=0
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/bnu_off -- python $ROOT/tools/micro/bn_bwd_unit.py 10 > $ROOT/gpurun_out/bnu_off.txt 2>&1
cd $ROOT
for d in on off; do cp $(ls gpurun_out/bnu_$d/*/*_kernel_trace.csv | head -1) gpurun_out/bnu_trace_$d.csv; rm -rf gpurun_out/bnu_$d; done
