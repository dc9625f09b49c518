set -x
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do
  timeout 200 python -m pytest tests/test_dist.py -q -m gpu -x -k "multirank_schedule_variants and 8-8192-512-extra8" 2>&1 | tail -1
done
