#!/bin/bash
# launches / collectives per factor call of the 1 x 8 and the 2 x 4 plan (rank 0), 8 ranks sharing this GPU (host-staged collectives)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
N=${1:-8192}; NB=${2:-512}
export MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29911 $R/tests/dist_worker.py --mode gpu --size $N --nb $NB 2>/dev/null | grep "DIST-OK"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29912 $R/tests/dist_worker.py --mode gpu --size $N --nb $NB --ipc 1 2>/dev/null | grep "DIST-OK"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29913 $R/tests/dist_worker.py --mode gpu2d --size $N --nb $NB --pr 2 2>/dev/null | grep "DIST2D-OK"
