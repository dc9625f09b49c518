#!/bin/bash
# round profile set: kernel trace + stats of the default bench, N=32768, CholeskyQR2; PMC traffic of the dominant kernel
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_round; rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b65536 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $OUT/b65536.log 2>&1
grep '^{' $OUT/b65536.log | cut -c1-400
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b32768 -o bench -- python $R/bench.py --n 32768 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/b32768.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cqr -o bench -- python $R/bench.py --workload cacqr --steps 3 --warmup 1 --no-cpu-baseline > $OUT/cqr.log 2>&1
grep '^{' $OUT/cqr.log | cut -c1-300
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-include-regex "dgemm_tn_dma_kernel<1" --output-format csv -d $OUT/pmc_$c -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-check > $OUT/pmc_$c.log 2>&1
  echo "$c rc=$?"
done
find $OUT -name "*.csv" | head -20
