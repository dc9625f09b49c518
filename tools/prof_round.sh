#!/bin/bash
# round profile set: kernel trace + stats of the default bench, N=32768, CholeskyQR2, mixed; PMC traffic of the dominant kernel.
# Copies the summaries into gpurun_out/prof_round/summary/ under the names they are committed with (profiles/<tag>_*).
export TMPDIR=/tmp
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_round; rm -rf $OUT; mkdir -p $OUT/summary
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b65536 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $OUT/b65536.log 2>&1
grep '^{' $OUT/b65536.log | cut -c1-400
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b32768 -o bench -- python $R/bench.py --n 32768 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/b32768.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b32768ci0 -o bench -- python $R/bench.py --n 32768 --complete-inv 0 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/b32768ci0.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cqr -o bench -- python $R/bench.py --workload cacqr --steps 3 --warmup 1 --no-cpu-baseline > $OUT/cqr.log 2>&1
grep '^{' $OUT/cqr.log | cut -c1-300
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mixed -o bench -- python $R/bench.py --workload mixed --steps 2 --warmup 1 --no-cpu-baseline > $OUT/mixed.log 2>&1
grep '^{' $OUT/mixed.log | cut -c1-300
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-include-regex "dgemm_tn_dma_kernel<1" --output-format csv -d $OUT/pmc_$c -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-check > $OUT/pmc_$c.log 2>&1
  echo "$c rc=$?"
done
# named regions (roctx ranges CI::factor_diag / CI::trsm / CI::tmu / CI::inverse_*, CQR::*): marker trace of a small factorization
timeout 200 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $OUT/markers -o bench -- python $R/bench.py --n 16384 --complete-inv 1 --steps 1 --warmup 1 --no-cpu-baseline --no-extra > $OUT/markers.log 2>&1
for f in $(find $OUT/markers -name "*marker*stats*.csv" -o -name "*marker_api_trace.csv" | head -3); do cp $f $OUT/summary/${TAG}_markers_$(basename $f); done
for t in b65536:bench_n65536 b32768:bench_n32768 b32768ci0:bench_n32768_complete_inv0 cqr:bench_cacqr_2p21x256 mixed:bench_mixed_n65536; do
  d=${t%%:*}; n=${t##*:}
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/summary/${TAG}_${n}_kernel_stats.csv
  grep '^{' $OUT/$d.log > $OUT/summary/${TAG}_${n}_profiled_stdout.log
done
for c in FETCH_SIZE WRITE_SIZE; do
  f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/summary/${TAG}_bench_n65536_pmc_${c}_trailing_kernel.csv
done
ls -la $OUT/summary
