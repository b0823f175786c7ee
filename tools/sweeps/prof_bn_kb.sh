# isolated BN-backward passes (tools/micro/bn_bwd_unit.py) under rocprofv3 for several block sizes (tuning library: CVX_BN_KB = KiB of xhat per block)
set -e
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
export CVX_LIB=$ROOT/build/libcvx_tuning.so
for kb in 16 32 64 128; do
  export CVX_BN_KB=$kb
  rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/bnkb_$kb -- python $ROOT/tools/micro/bn_bwd_unit.py 10 > $ROOT/gpurun_out/bnkb_$kb.txt 2>&1
  cp $(ls $ROOT/gpurun_out/bnkb_$kb/*/*_kernel_trace.csv | head -1) $ROOT/gpurun_out/bnkb_trace_$kb.csv; rm -rf $ROOT/gpurun_out/bnkb_$kb
done
