set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r04_full_gpu_suite.log; cat gpurun_out/r04_full_gpu_suite.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
