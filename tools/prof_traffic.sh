#!/bin/bash
# PMC passes restricted to the trailing-update kernel of the bench command (FETCH_SIZE / WRITE_SIZE in separate runs).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_traffic; rm -rf $OUT; mkdir -p $OUT
cd /tmp
ARGS="${BENCH_ARGS:---steps 1 --warmup 0 --no-cpu-baseline}"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout ${PMC_TIMEOUT:-170} rocprofv3 --pmc $c --kernel-include-regex "dgemm_tn_dma_kernel<1" --output-format csv -d $OUT/$c -o bench -- python $R/bench.py $ARGS > $OUT/$c.log 2>&1
  echo "$c rc=$?"; ls $OUT/$c 2>/dev/null | head -3
done
