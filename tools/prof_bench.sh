#!/bin/bash
# rocprofv3 of the bench command: kernel trace + stats, then separate PMC passes (FETCH_SIZE / WRITE_SIZE).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_bench; rm -rf $OUT; mkdir -p $OUT
cd /tmp
ARGS="${BENCH_ARGS:---steps 2 --warmup 1 --no-cpu-baseline}"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py $ARGS > $OUT/trace.log 2>&1
grep '^{' $OUT/trace.log
head -8 $OUT/trace/bench_kernel_stats.csv | cut -c1-200
if [ -n "$PMC" ]; then
  PARGS="${PMC_ARGS:---steps 1 --warmup 0 --no-cpu-baseline}"
  timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python $R/bench.py $PARGS > $OUT/pmc_fetch.log 2>&1
  timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python $R/bench.py $PARGS > $OUT/pmc_write.log 2>&1
  ls -la $OUT/pmc_fetch $OUT/pmc_write
fi
