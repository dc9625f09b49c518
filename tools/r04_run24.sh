set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mixed.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/mp_kernel_ab.py 65536 8 2>&1 | grep -v amdgpu | grep "kernel=0" > gpurun_out/r04_mp_ab7.log; cat gpurun_out/r04_mp_ab7.log
timeout 300 python tools/mixed_bench.py 65536 8 2>&1 | grep -v amdgpu | tee gpurun_out/r04_mixed_bench1.log
timeout 300 python tools/cqr_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04_cqr_bench2.log
CAP_CHAIN_COOP=0 timeout 300 python tools/cqr_bench.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/r04_cqr_bench2.log
