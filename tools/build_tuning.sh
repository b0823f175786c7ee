#!/bin/bash
# Builds a SECOND copy of the library with the tuning knobs compiled in (-DCVX_TUNING: environment variables select tile
# thresholds etc.) for A/B sweeps on the GPU box:   CVX_LIB=build/libcvx_tuning.so CVX_BN_KB=64 python bench.py ...
# It also carries the tile-resident chain kernel (conv_chain.hip, -DCVX_WITH_CHAIN), which the release library dropped in round 4.
# The release library (computervision.pytorch_amd/lib/libcvx_engine.so) never reads the environment.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/obj_tuning
SRC=computervision.pytorch_amd/csrc
pids=()
for f in $SRC/*.hip; do
  o=build/obj_tuning/$(basename ${f%.hip}).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCVX_TUNING -DCVX_WITH_CHAIN -c $f -o $o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libcvx_tuning.so build/obj_tuning/*.o
echo built build/libcvx_tuning.so
