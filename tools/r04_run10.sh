set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_cacqr.py -x -q -m gpu 2>&1 | tail -4
CAP_CQR_PAIR=0 timeout 120 python tools/cqr_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r04_cqr_bench.log
CAP_CQR_PAIR=1 timeout 120 python tools/cqr_bench.py 2>&1 | grep -v amdgpu >> gpurun_out/r04_cqr_bench.log
CAP_CQR_PAIR=0 timeout 120 python tools/cqr_bench.py 2>&1 | grep -v amdgpu >> gpurun_out/r04_cqr_bench.log
CAP_CQR_PAIR=1 timeout 120 python tools/cqr_bench.py 2>&1 | grep -v amdgpu >> gpurun_out/r04_cqr_bench.log
cat gpurun_out/r04_cqr_bench.log
timeout 900 python -m pytest tests/test_dist.py -q -m gpu -k "2d_block_cyclic_schedule_options or (multirank_schedule_variants and ipc)" 2>&1 | tail -5
timeout 300 python tools/dist_p1_bench.py 32768 2>&1 | grep "2D plan\|strip=2 depth2=1" > gpurun_out/r04_dist_p1b.log; cat gpurun_out/r04_dist_p1b.log
