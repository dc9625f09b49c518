set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_mixed.py -x -q -m gpu -k "bf16_update" 2>&1 | tail -25 > gpurun_out/r04_t4.log
cat gpurun_out/r04_t4.log | tail -12
timeout 300 python tools/bf16_bench.py 16384 32768 49152 > gpurun_out/r04_bf16_bench.log 2>&1
cat gpurun_out/r04_bf16_bench.log
timeout 400 python tools/mp_kernel_ab.py 65536 4 8 16 > gpurun_out/r04_mp_ab.log 2>&1; tail -12 gpurun_out/r04_mp_ab.log
timeout 900 python -m pytest tests/test_dist.py -q -m gpu -k "distributed_inverse or 8rank_reference" 2>&1 | tail -8 > gpurun_out/r04_t5.log
tail -8 gpurun_out/r04_t5.log
