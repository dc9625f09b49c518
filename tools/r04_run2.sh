set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_dist.py -q -m gpu -k "redistribution or reference_layout or summa_trmm or reference_recursion or distributed_inverse or 8rank_reference or summa_gemm" 2>&1 | tail -40 > gpurun_out/r04_t3.log
tail -c 4000 gpurun_out/r04_t3.log
