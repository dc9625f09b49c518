#!/bin/bash
# round-5 counter passes (separate --pmc runs, no trace domain next to them):
#   1. the two CholeskyQR kernels (gram256 / qrapply256) inside tools/cqr_bench.py: matrix-pipe busy, waits, LDS waits, L2 hits
#   2. the bf16 trailing update inside the mixed factorization at N = 65536 with the paired far update (tools/mp_bench.bin): K = 2048 and K = 4096 launches
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_r05_pmc; rm -rf $OUT; mkdir -p $OUT/summary
cd /tmp
run_cqr() { name=$1; shift; timeout 170 rocprofv3 "$@" --kernel-include-regex "qrapply256|gram256_kernel" --output-format csv -d $OUT/$name -o cqr -- python $R/tools/cqr_bench.py > $OUT/$name.log 2>&1; tail -n 1 $OUT/$name.log; }
run_cqr cqr1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run_cqr cqr2 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64
run_mp() { name=$1; shift; timeout 170 rocprofv3 "$@" --kernel-include-regex "bf16_tn" --output-format csv -d $OUT/$name -o mp -- $R/tools/mp_bench.bin 65536 1 > $OUT/$name.log 2>&1; tail -n 1 $OUT/$name.log; }
run_mp mp1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run_mp mp2 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
python3 - <<PY > $OUT/summary/r05_pmc_cqr_bf16.txt
import csv, glob, collections
def rows(name):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True)
    return list(csv.DictReader(open(fs[0]))) if fs else []
print("== CholeskyQR2 2^21 x 256 (tools/cqr_bench.py): per kernel, averages per dispatch")
for name in ("cqr1", "cqr2"):
    acc = collections.defaultdict(list)
    for r in rows(name):
        k = "qrapply256" if "qrapply256" in r["Kernel_Name"] else "gram256"
        acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()): print(name, k, c, "avg %.4g over %d dispatches" % (sum(v) / len(v), len(v)))
print("== bf16_tn_kernel inside the mixed factorization, N = 65536, paired far update (tools/mp_bench.bin 65536 1): launches by grid size")
for name in ("mp1", "mp2"):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in rows(name):
        g = int(r["Grid_Size"])
        if g < 5120000: continue                       # the big updates only
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for c, v in sorted(acc.items()): print(name, c, "sum over %d big launches %.4g" % (n[c], v))
PY
cat $OUT/summary/r05_pmc_cqr_bf16.txt
for f in cqr1 mp1; do tail -n 2 $OUT/$f.log; done
rm -rf $OUT/cqr1 $OUT/cqr2 $OUT/mp1 $OUT/mp2
