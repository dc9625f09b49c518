set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 300 python -m pytest tests/test_dist.py -q -m gpu -x -k "multirank_schedule_variants and 8-8192-512" 2>&1 | tail -2
done
for i in 1 2 3; do
  CAP_CHAIN_FENCE=1 timeout 300 python -m pytest tests/test_dist.py -q -m gpu -x -k "multirank_schedule_variants and 8-8192-512" 2>&1 | tail -2
done
for i in 1 2; do
  CAP_CHAIN_COOP=0 timeout 300 python -m pytest tests/test_dist.py -q -m gpu -x -k "multirank_schedule_variants and 8-8192-512" 2>&1 | tail -2
done
