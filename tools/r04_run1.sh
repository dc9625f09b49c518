set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dist.py -x -q -m gpu -k "redistribution or reference_layout or mixed_precision" 2>&1 | tail -15 > gpurun_out/r04_t1.log
timeout 600 python -m pytest tests/test_gpu_cholinv.py -x -q -m gpu -k "rinv_is_validated or workspace_follows or knobs" 2>&1 | tail -15 > gpurun_out/r04_t2.log
timeout 900 python bench.py > gpurun_out/r04_bench1.json 2> gpurun_out/r04_bench1.err
tail -c 3000 gpurun_out/r04_t1.log gpurun_out/r04_t2.log
