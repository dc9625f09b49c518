#!/bin/bash
# rocprofv3 kernel trace + stats of the full bench command; output under gpurun_out/prof_bench
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_bench; rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py ${BENCH_ARGS:---n 65536 --steps 2 --warmup 1 --no-cpu-baseline} > $OUT/trace.log 2>&1
grep '^{' $OUT/trace.log
head -30 $OUT/trace/bench_kernel_stats.csv
