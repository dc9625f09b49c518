set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_cholinv.py -x -q -m gpu -k "one_launch" 2>&1 | tail -5
timeout 300 python tools/chain_trace.py 32 20 2>&1 | grep -v amdgpu > gpurun_out/r04_chain_trace2.log; cat gpurun_out/r04_chain_trace2.log
CHAIN_N64=32768 timeout 600 python tools/chain_ab.py 0 16 32 64 2>&1 | grep -v "amdgpu\|alone" > gpurun_out/r04_chain_ab2.log; cat gpurun_out/r04_chain_ab2.log
