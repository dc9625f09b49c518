set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_mixed.py -x -q -m gpu 2>&1 | tail -4
export CAP_BF16_V2=0
bash tools/prof_mixed.sh 65536
cp gpurun_out/prof_mixed/trace/*kernel_stats.csv gpurun_out/r04_mixed_v1_solve3_kernel_stats.csv
cat gpurun_out/prof_mixed/trace.log | tail -3
