set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
OUT=$R/gpurun_out/prof_chain_t; rm -rf $OUT; mkdir -p $OUT
timeout 170 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o mx -- python $R/tools/mp_factor_only.py 65536 2 > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log
F=$(ls $OUT/trace/*kernel_trace.csv | head -1)
python $R/tools/kernel_sums.py $F > $R/gpurun_out/r04_chain_sums_32b.log
head -30 $R/gpurun_out/r04_chain_sums_32b.log
python $R/tools/strip_timeline.py $F 90 2 > $R/gpurun_out/r04_strip_timeline_a.log
python $R/tools/strip_timeline.py $F 110 2 > $R/gpurun_out/r04_strip_timeline_b.log
python $R/tools/strip_timeline.py $F 74 2 > $R/gpurun_out/r04_strip_timeline_c.log
rm -rf $OUT
