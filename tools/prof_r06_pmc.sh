#!/bin/bash
# round-6 counter passes (separate --pmc runs, no trace domain next to them): the third-generation bf16 update (bf16_tn3x_kernel) inside the
# mixed-precision factorization at N = 65536 (python tools/mp_factor_only.py) and the two CholeskyQR kernels (tools/cqr_bench.py)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_r06_pmc; rm -rf $OUT; mkdir -p $OUT/summary
cd /tmp
run_mp() { name=$1; shift; timeout 250 rocprofv3 "$@" --kernel-include-regex "bf16_tn3" --output-format csv -d $OUT/$name -o mp -- python $R/tools/mp_factor_only.py 65536 1 > $OUT/$name.log 2>&1; tail -n 1 $OUT/$name.log; }
run_mp mp1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run_mp mp2 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
run_cqr() { name=$1; shift; timeout 170 rocprofv3 "$@" --kernel-include-regex "qrapply256|gram256_kernel" --output-format csv -d $OUT/$name -o cqr -- python $R/tools/cqr_bench.py > $OUT/$name.log 2>&1; tail -n 1 $OUT/$name.log; }
run_cqr cqr1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
python3 - <<PY > $OUT/summary/r06_bf16_update_pmc.txt
import csv, glob, collections
def rows(name):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True)
    return list(csv.DictReader(open(fs[0]))) if fs else []
print("== bf16_tn3x_kernel (third-generation bf16 update) inside the mixed-precision factorization, N = 65536 (python tools/mp_factor_only.py 65536 1: 2 factor calls)")
print("   launches of >= 1024 workgroups (the bulk + big head updates); separate --pmc passes; SQ_* summed over the chip, GRBM over the 8 XCDs")
for name in ("mp1", "mp2"):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in rows(name):
        if int(r["Grid_Size"]) < 1024 * 512: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for c, v in sorted(acc.items()): print(name, c, "sum over %d launches %.4g" % (n[c], v))
    if name == "mp1" and acc.get("SQ_BUSY_CYCLES"):
        # matrix pipe busy: SQ_VALU_MFMA_BUSY_CYCLES (per SIMD cycles, 1024 SIMDs) against the launches' shader-engine busy cycles
        print("   waves waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) = %.3f ; issue stalls (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) = %.3f" % (acc["SQ_WAIT_ANY"] / acc["SQ_WAVE_CYCLES"], acc["SQ_WAIT_INST_ANY"] / acc["SQ_WAVE_CYCLES"]))
    if name == "mp2" and acc.get("GRBM_GUI_ACTIVE"):
        print("   TCC hit rate = %.3f ; GRBM_GUI_ACTIVE per XCD = %.4g cycles" % (acc["TCC_HIT_sum"] / (acc["TCC_HIT_sum"] + acc["TCC_MISS_sum"]), acc["GRBM_GUI_ACTIVE"] / 8))
a1 = collections.defaultdict(float)
for r in rows("mp1"):
    if int(r["Grid_Size"]) >= 1024 * 512: a1[r["Counter_Name"]] += float(r["Counter_Value"])
a2 = collections.defaultdict(float)
for r in rows("mp2"):
    if int(r["Grid_Size"]) >= 1024 * 512: a2[r["Counter_Name"]] += float(r["Counter_Value"])
if a1.get("SQ_VALU_MFMA_BUSY_CYCLES") and a2.get("GRBM_GUI_ACTIVE"):
    print("   matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8) = %.3f" % (a1["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (a2["GRBM_GUI_ACTIVE"] / 8)))
print("== CholeskyQR2 2^21 x 256 (tools/cqr_bench.py): per kernel, averages per dispatch (qrapply256: row-sliced waves, hand-over work inside the MFMA stream)")
acc = collections.defaultdict(list)
for r in rows("cqr1"):
    k = "qrapply256" if "qrapply256" in r["Kernel_Name"] else "gram256"
    acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()): print("cqr1", k, c, "avg %.4g over %d dispatches" % (sum(v) / len(v), len(v)))
PY
cat $OUT/summary/r06_bf16_update_pmc.txt
rm -rf $OUT/mp1 $OUT/mp2 $OUT/cqr1
