#!/bin/bash
# kernel trace of the mixed-precision factorization alone (N = 65536, options from MP_OPTIONS) + the bulk-launch analysis of tools/r06_mp_trace.py
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
TAG=${1:-v3}
OUT=$R/gpurun_out/prof_mp_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o mx -- python $R/tools/mp_factor_only.py ${2:-65536} 2 > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log
T=$(ls $OUT/trace/*/*kernel_trace.csv $OUT/trace/*kernel_trace.csv 2>/dev/null | head -1)
python $R/tools/r06_mp_trace.py $T > $R/gpurun_out/r06_mp_trace_$TAG.txt 2>&1
tail -60 $R/gpurun_out/r06_mp_trace_$TAG.txt
