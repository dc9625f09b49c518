set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
for G in 0 32; do
  OUT=$R/gpurun_out/prof_chain_$G; rm -rf $OUT; mkdir -p $OUT
  CAP_CHAIN_COOP=$G timeout 170 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o mx -- python $R/tools/mp_factor_only.py 65536 2 > $OUT/trace.log 2>&1
  tail -1 $OUT/trace.log
  python $R/tools/kernel_sums.py $(ls $OUT/trace/*kernel_trace.csv | head -1) > $R/gpurun_out/r04_chain_sums_$G.log
  cat $R/gpurun_out/r04_chain_sums_$G.log
  rm -rf $OUT
done
