#!/bin/bash
# round-6 evidence set: the default bench line, kernel trace + stats of the headline / N = 32768 / CholeskyQR2 / mixed runs, the PMC passes of the
# dominant kernel's HBM traffic (separate --pmc runs, never combined with a trace domain), the emulated `bench.py --gpus 2` line started
# WITHOUT a launcher.  Summaries land in gpurun_out/prof_r06/summary under the names they are committed with (profiles/r06_*).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_r06; rm -rf $OUT; mkdir -p $OUT/summary
cd $R
timeout 700 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
grep '^{' $OUT/bench_default.log > $OUT/summary/r06_bench_default_stdout.log; cut -c1-700 $OUT/summary/r06_bench_default_stdout.log
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b65536 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $OUT/b65536.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b32768 -o bench -- python $R/bench.py --n 32768 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/b32768.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cqr -o bench -- python $R/bench.py --workload cacqr --steps 3 --warmup 1 --no-cpu-baseline > $OUT/cqr.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mixed -o bench -- python $R/bench.py --workload mixed --steps 2 --warmup 1 --no-cpu-baseline > $OUT/mixed.log 2>&1
for t in b65536:bench_n65536 b32768:bench_n32768 cqr:bench_cacqr_2p21x256 mixed:bench_mixed_n65536; do
  d=${t%%:*}; n=${t##*:}
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/summary/r06_${n}_kernel_stats.csv
  grep '^{' $OUT/$d.log > $OUT/summary/r06_${n}_profiled_stdout.log
  rm -rf $OUT/$d
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-include-regex "dgemm_tn_dma_kernel<1" --output-format csv -d $OUT/pmc_$c -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-check > $OUT/pmc_$c.log 2>&1
  echo "$c rc=$?"
  f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/summary/r06_bench_n65536_pmc_${c}_trailing_kernel.csv
done
cd $R
python tools/make_traffic_json.py $OUT r06 > $OUT/summary/r06_traffic_summary.txt 2>&1; cat $OUT/summary/r06_traffic_summary.txt
cp profiles/r06_traffic_bench_n65536.json $OUT/summary/ 2>/dev/null
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
# the P-GPU projection (single-GPU replay of every rank of the 1 x 8 schedule) and the SUMMA line
timeout 600 python bench.py --replay-rank -1 --of 8 --steps 3 > $OUT/replay8.log 2> $OUT/replay8.err
grep '^{' $OUT/replay8.log > $OUT/summary/r06_replay_bench_p8_final.json
timeout 300 python bench.py --workload summa --no-cpu-baseline > $OUT/summa.log 2> $OUT/summa.err
grep '^{' $OUT/summa.log > $OUT/summary/r06_bench_summa_stdout.log; cut -c1-300 $OUT/summary/r06_bench_summa_stdout.log
# the N > 1 entry point as the driver uses it (no launcher around it), two ranks sharing this GPU
env -u RANK -u WORLD_SIZE -u LOCAL_RANK CAPITAL_BENCH_EMULATE=1 timeout 600 python bench.py --gpus 2 --size 8192 --steps 1 --warmup 1 --cpu-n 2048 --cpu-budget-s 40 > $OUT/emu2.log 2> $OUT/emu2.err
grep '^{' $OUT/emu2.log > $OUT/summary/r06_bench_gpus2_emulated_stdout.log; cut -c1-400 $OUT/summary/r06_bench_gpus2_emulated_stdout.log
ls -la $OUT/summary
