#!/bin/bash
# PMC passes on the standalone GEMM/SYRK bench: MFMA busy, waits, effective clock
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_gemm2; rm -rf $OUT; mkdir -p $OUT
cd /tmp
ARGS="${GEMM_ARGS:-8192 8192 8192 0 2}"
run() { name=$1; shift; timeout 120 rocprofv3 "$@" --kernel-include-regex "dgemm_tn_dma" --output-format csv -d $OUT/$name -o gemm -- $R/tools/gemm_bench.bin $ARGS > $OUT/$name.log 2>&1; tail -n 1 $OUT/$name.log; }
run trace --kernel-trace --stats
run pmc1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES
run pmc2 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS
python3 - <<PY
import csv, glob, collections
for name in ("pmc1", "pmc2"):
    fs = glob.glob("$OUT/%s/*counter_collection.csv" % name)
    if not fs: print(name, "no csv"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print(name, k, "per dispatch avg %.4g over %d" % (sum(v)/len(v), len(v)))
fs = glob.glob("$OUT/trace/*kernel_stats.csv")
if fs: print(open(fs[0]).read())
PY
