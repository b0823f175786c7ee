# Phase stamps (s_memtime / s_memrealtime inside the kernels, tuning library) of every <= 40x40 convolution launch of YOLOv8-n at batch 32:
# the round-4 dispatcher (row-band kernel where it applies) against the round-3 kernels (mode bit 0x4000), three epilogues.
#   gpurun -- bash tools/sweeps/conv_floor.sh     -> gpurun_out/r04_floor/*.txt   (tools/sweeps/conv_floor_table.py folds them into a table)
set -e
export CVX_LIB=$(pwd)/build/libcvx_tuning.so
export CONV_CLOCK_SET=r04
mkdir -p gpurun_out/r04_floor
for mode in 1 3 0; do
  CONV_CLOCK_MODE=$mode python tools/conv_clock.py > gpurun_out/r04_floor/new_mode$mode.txt 2>&1
  CONV_CLOCK_MODE=$mode CONV_CLOCK_FORCE=0x4000 python tools/conv_clock.py > gpurun_out/r04_floor/old_mode$mode.txt 2>&1
done
