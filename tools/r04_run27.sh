set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/r04_full_gpu_suite.log; cat gpurun_out/r04_full_gpu_suite.log
