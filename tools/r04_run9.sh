set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_dist.py -q -m gpu -k "2d_block_cyclic" 2>&1 | tail -12 > gpurun_out/r04_t6.log
tail -12 gpurun_out/r04_t6.log
timeout 400 python tools/dist_p1_bench.py 32768 > gpurun_out/r04_dist_p1.log 2>&1; grep -v amdgpu gpurun_out/r04_dist_p1.log
timeout 300 python -m pytest tests/test_gpu_mixed.py -x -q -m gpu 2>&1 | tail -3
MP_MIN_TILES=1024 timeout 300 python tools/mp_kernel_ab.py 65536 8 > gpurun_out/r04_mp_ab5.log 2>&1; tail -4 gpurun_out/r04_mp_ab5.log
