set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_mixed.py -x -q -m gpu 2>&1 | tail -6
timeout 300 python tools/bf16_bench.py 49152 > gpurun_out/r04_bf16_bench3.log 2>&1
cat gpurun_out/r04_bf16_bench3.log
timeout 500 python tools/mp_kernel_ab.py 65536 8 > gpurun_out/r04_mp_ab2.log 2>&1; tail -14 gpurun_out/r04_mp_ab2.log
