#!/bin/bash
# where does the diagonal-block chain lose time under a concurrent bulk update?  PMC on the chain kernels inside bench N=32768
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_chain; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $OUT/sq_counters.txt
run() { name=$1; shift; timeout 170 rocprofv3 "$@" --kernel-include-regex "leaf_cholinv|panel64|dgemm_small" --output-format csv -d $OUT/$name -o b -- python $R/bench.py --n 32768 --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-check > $OUT/$name.log 2>&1; tail -n 1 $OUT/$name.log | cut -c1-200; }
run pmc1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_WAVES
run pmc2 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VALU
python3 - <<PY
import csv, glob, collections
for name in ("pmc1", "pmc2"):
    fs = glob.glob("$OUT/%s/*counter_collection.csv" % name)
    if not fs: print(name, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(name, k, {c: "%.4g" % (sum(v)/len(v)) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
