#!/bin/bash
# round-4 evidence set: the default bench line, kernel trace + stats of the headline / N = 32768 / CholeskyQR2 / mixed runs, PMC passes on both
# generations of the bf16 update.  Summaries land in gpurun_out/prof_r04/summary under the names they are committed with (profiles/r04_*).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_r04; rm -rf $OUT; mkdir -p $OUT/summary
cd $R
timeout 600 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
grep '^{' $OUT/bench_default.log > $OUT/summary/r04_bench_default_stdout.log; cut -c1-600 $OUT/summary/r04_bench_default_stdout.log
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b65536 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $OUT/b65536.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b32768 -o bench -- python $R/bench.py --n 32768 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/b32768.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cqr -o bench -- python $R/bench.py --workload cacqr --steps 3 --warmup 1 --no-cpu-baseline > $OUT/cqr.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mixed -o bench -- python $R/bench.py --workload mixed --steps 2 --warmup 1 --no-cpu-baseline > $OUT/mixed.log 2>&1
for t in b65536:bench_n65536 b32768:bench_n32768 cqr:bench_cacqr_2p21x256 mixed:bench_mixed_n65536; do
  d=${t%%:*}; n=${t##*:}
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/summary/r04_${n}_kernel_stats.csv
  grep '^{' $OUT/$d.log > $OUT/summary/r04_${n}_profiled_stdout.log
  rm -rf $OUT/$d
done
cd $R
{ echo "== bf16_tn_kernel (v1, default) inside the mixed-precision factorization, N = 32768 (tools/prof_bf16.sh)"; MP_N=32768 SUFFIX=_v1 bash tools/prof_bf16.sh;
  echo "== bf16_tn_v2_kernel (CAP_BF16_V2=1, every launch: CAP_BF16_V2_MIN=0)"; CAP_BF16_V2=1 CAP_BF16_V2_MIN=0 MIN_GRID=300000 MP_N=32768 SUFFIX=_v2 bash tools/prof_bf16.sh; } > $OUT/summary/r04_bf16_update_pmc.txt 2>&1
cat $OUT/summary/r04_bf16_update_pmc.txt
rm -rf $R/gpurun_out/prof_bf16_v1 $R/gpurun_out/prof_bf16_v2
ls -la $OUT/summary
