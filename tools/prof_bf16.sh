#!/bin/bash
# PMC passes on the bf16 trailing update inside the mixed-precision factorization (tools/mp_bench.bin): MFMA busy, waits, LDS conflicts, L2 hit rate
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_bf16${SUFFIX}; rm -rf $OUT; mkdir -p $OUT
cd /tmp
run() { name=$1; shift; timeout 150 rocprofv3 "$@" --kernel-include-regex "bf16_tn" --output-format csv -d $OUT/$name -o mp -- $R/tools/mp_bench.bin ${MP_N:-32768} 1 > $OUT/$name.log 2>&1; tail -n 1 $OUT/$name.log; }
run pmc1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run pmc2 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
python3 - <<PY
import csv, glob, collections
for name in ("pmc1", "pmc2"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True)
    if not fs: print(name, "no csv"); continue
    acc = collections.defaultdict(float); n = collections.Counter(); grid = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        if int(r["Grid_Size"]) < int("${MIN_GRID:-5120000}"): continue           # the big updates only (MIN_GRID threads)
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k, v in acc.items(): print(name, k, "sum over %d big launches %.4g" % (n[k], v))
PY
