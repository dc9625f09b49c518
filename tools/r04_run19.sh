set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cholinv.py -x -q -m gpu -k "one_launch or not_spd or matches_oracle or reference_dump or harder or knobs" 2>&1 | tail -5
timeout 300 python tools/chain_trace.py 32 20 2>&1 | grep -v amdgpu > gpurun_out/r04_chain_trace6.log; grep -v "^  *[0-9]* |" gpurun_out/r04_chain_trace6.log
CHAIN_N64=32768 timeout 600 python tools/chain_ab.py 0 32 2>&1 | grep -v "amdgpu\|alone" > gpurun_out/r04_chain_ab4.log; cat gpurun_out/r04_chain_ab4.log
