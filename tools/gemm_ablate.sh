#!/bin/bash
# Phase timeline and ablations of the GEMM-shaped conv kernel (tuning build; run on the GPU box from the repo root):
#   per variant (conv_gemm.hip kVariants 1..7): workgroups, prologue / K loop / epilogue per workgroup, end of the launch, clock held;
#   on SSD's conv4_x with the cost model's variant: the K loop without DMA (32), without MFMA (64), without both (96), without the barrier (128).
# The numbers quoted in DESIGN.md section 5b come from this script.     bash tools/build_tuning.sh && gpurun -- 'bash tools/gemm_ablate.sh'
set -o pipefail
export CVX_LIB=build/libcvx_tuning.so
for v in 1 2 3 4 5 6 7; do
  echo "== variant $v, folded BN + SiLU epilogue"
  CVX_GEMM_TILE=$v timeout -k 10 120 python tools/gemm_debug.py 2>&1 | grep -v amdgpu.ids | tail -5 || exit 1
done
echo "== training epilogue (raw fp32 + statistics), cost model's variants"
GEMM_DEBUG_MODE=3 timeout -k 10 120 python tools/gemm_debug.py 2>&1 | grep -v amdgpu.ids | tail -5 || exit 1
for d in 0 32 64 96 128; do
  echo "== conv4_x, 256 x 256, CVX_GEMM_DBG=$d"
  CVX_GEMM_TILE=1 CVX_GEMM_DBG=$d GEMM_DEBUG_FIRST=1 timeout -k 10 120 python tools/gemm_debug.py 2>&1 | grep -v amdgpu.ids | tail -1 || exit 1
done
