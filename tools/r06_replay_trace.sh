#!/bin/bash
# kernel trace of one replayed rank (tools/replay.py) + what runs beside its big bulk launches
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
RANK=${1:-6}; shift
OUT=$R/gpurun_out/prof_replay_r$RANK; rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o rp -- python $R/tools/replay.py --n 65536 --of 8 --ranks $RANK --steps 1 "$@" > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log | cut -c1-400
T=$(ls $OUT/trace/*/*kernel_trace.csv $OUT/trace/*kernel_trace.csv 2>/dev/null | head -1)
python $R/tools/r06_replay_trace.py $T > $R/gpurun_out/r06_replay_trace_r$RANK.txt 2>&1
head -70 $R/gpurun_out/r06_replay_trace_r$RANK.txt
