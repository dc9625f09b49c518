set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CHAIN_NMP=16384 timeout 300 python tools/chain_trace.py 32 5 2>&1 | grep -v amdgpu > gpurun_out/r04_chain_trace5.log; grep -v "^  *[0-9]* |" gpurun_out/r04_chain_trace5.log
