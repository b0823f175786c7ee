set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CVX_LIB=build/libcvx_tuning.so
GEMM_PROBE_FORCE=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gemm_trace_force -- python3 tools/gemm_probe.py > gpurun_out/gemm_trace_force.log 2>&1 && python tools/gemm_trace.py gpurun_out/gemm_trace_force > gpurun_out/gemm_trace_force.txt && cat gpurun_out/gemm_trace_force.txt
