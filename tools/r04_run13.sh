set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/chain_trace.py 32 20 2>&1 | grep -v amdgpu > gpurun_out/r04_chain_trace1.log; cat gpurun_out/r04_chain_trace1.log
