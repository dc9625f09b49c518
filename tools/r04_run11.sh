set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dist.py -q -m gpu -k "2d_block_cyclic" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_mixed.py tests/test_gpu_cacqr.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/dist_p1_bench.py 32768 2>&1 | grep "2D plan\|strip=2 depth2=1\|single" > gpurun_out/r04_dist_p1c.log; cat gpurun_out/r04_dist_p1c.log
timeout 300 python tools/mp_kernel_ab.py 65536 8 2>&1 | grep -v amdgpu > gpurun_out/r04_mp_ab5.log; tail -12 gpurun_out/r04_mp_ab5.log
