set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_cholinv.py -x -q -m gpu -k "one_launch or not_spd or matches_oracle or reference_dump" 2>&1 | tail -3
cd /tmp
OUT=$R/gpurun_out/prof_chain_t; rm -rf $OUT; mkdir -p $OUT
timeout 170 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o mx -- python $R/tools/mp_factor_only.py 65536 1 > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log
F=$(ls $OUT/trace/*kernel_trace.csv | head -1)
python $R/tools/bulk_gaps.py $F > $R/gpurun_out/r04_bulk_gaps.log
cat $R/gpurun_out/r04_bulk_gaps.log
rm -rf $OUT
cd $R
timeout 300 python tools/chain_trace.py 32 20 2>&1 | grep -v amdgpu > gpurun_out/r04_chain_trace8.log; grep -v "^  *[0-9]* |" gpurun_out/r04_chain_trace8.log | head -8
