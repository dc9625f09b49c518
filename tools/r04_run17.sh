set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_cholinv.py -x -q -m gpu -k "one_launch or not_spd or matches_oracle" 2>&1 | tail -5
timeout 300 python tools/chain_trace.py 32 20 2>&1 | grep -v amdgpu > gpurun_out/r04_chain_trace4.log; grep -v "^  *[0-9]* |" gpurun_out/r04_chain_trace4.log
