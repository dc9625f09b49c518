#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_mixed; rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o mx -- python $R/tools/mixed_bench.py ${1:-32768} 8 > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log
head -16 $OUT/trace/*kernel_stats.csv | cut -c1-200
