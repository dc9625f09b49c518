#!/bin/bash
# NOTE: the kernel variant this script switches on (CAP_KPHASE / CAP_KSTAG in tn_dma_tile, gemm.hip) was measured from a working-tree patch and
# removed again because it made things worse (profiles/r03_experiments.log section 8 describes it); the script is kept for the record.
# k-phase alignment of the bulk update (CAP_KPHASE): timing stand-alone and in the factorization, L2-miss traffic (FETCH_SIZE)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/exp_kphase; rm -rf $OUT; mkdir -p $OUT
cd $R
[ -x tools/gemm_bench.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/gemm_bench.bin tools/gemm_bench.cpp -Lcapital_amd/lib -lcapital_amd -Wl,-rpath,$R/capital_amd/lib
for kp in 0 2; do
  echo "== CAP_KPHASE=$kp"
  CAP_KPHASE=$kp tools/gemm_bench.bin 32768 32768 1024 1 5
  CAP_KPHASE=$kp tools/gemm_bench.bin 24576 24576 1024 1 5
  CAP_KPHASE=$kp tools/gemm_bench.bin 8192 8192 8192 0 3
done
cd /tmp
for kp in 0 2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    CAP_KPHASE=$kp timeout 120 rocprofv3 --pmc $c --kernel-include-regex "dgemm_tn_dma" --output-format csv -d $OUT/pmc_${kp}_$c -o g -- $R/tools/gemm_bench.bin 24576 24576 1024 1 3 > $OUT/pmc_${kp}_$c.log 2>&1
  done
done
python3 - <<PY
import csv, glob
for kp in (0, 2):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob("$OUT/pmc_%d_%s/**/*counter_collection.csv" % (kp, c), recursive=True)
        if not fs: print(kp, c, "no csv"); continue
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == c]
        print("CAP_KPHASE=%d %s per launch: %.3f GB reported (x2 for FETCH on gfx950), %d launches" % (kp, c, sum(v) / len(v) * 1024 / 1e9, len(v)))
PY
cd $R
for kp in 0 1; do
  echo "== factorization CAP_KPHASE=$kp"
  CAP_KPHASE=$kp timeout 200 tools/opt_bench.bin 32768 -1 3
  CAP_KPHASE=$kp timeout 200 tools/opt_bench.bin 65536 -1 2
done
