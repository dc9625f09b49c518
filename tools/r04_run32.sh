set -x
cd $GRAFT_REPO_ROOT
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 200 python -m pytest tests/test_gpu_cholinv.py -x -q -m gpu -k "one_launch_diagonal or not_spd" 2>&1 | tail -2
