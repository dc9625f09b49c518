set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CHAIN_NMP=0 CHAIN_N64=32768,65536 timeout 300 python tools/chain_ab.py 32 8 16 2>&1 | grep -v "amdgpu\|alone" > gpurun_out/r04_chain_ab6.log; cat gpurun_out/r04_chain_ab6.log
