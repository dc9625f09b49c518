#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_cqr; rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o cqr -- python $R/tools/cqr_bench.py > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log
cat $OUT/trace/*kernel_stats.csv | cut -c1-230
