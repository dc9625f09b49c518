set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_mixed.py -x -q -m gpu -k "bf16_update_kernels" 2>&1 | tail -5
timeout 300 python tools/bf16_bench.py 32768 49152 > gpurun_out/r04_bf16_bench2.log 2>&1
cat gpurun_out/r04_bf16_bench2.log
BF16_K=1024 BF16_PLAIN=1 timeout 200 python tools/bf16_bench.py 49152 >> gpurun_out/r04_bf16_bench2.log 2>&1
BF16_K=4096 BF16_PLAIN=1 timeout 200 python tools/bf16_bench.py 49152 >> gpurun_out/r04_bf16_bench2.log 2>&1
tail -8 gpurun_out/r04_bf16_bench2.log
