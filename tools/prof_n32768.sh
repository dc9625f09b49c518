#!/bin/bash
# kernel trace + stats of the N = 32768 bench (configs[1]: the diagonal-block chain under contention)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_n32768; rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o bench -- python $R/bench.py --n 32768 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | cut -c1-200
cut -c1-200 $OUT/t/bench_kernel_stats.csv | head -14
