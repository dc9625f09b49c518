#!/bin/bash
# NOTE: the kernel variant this script switches on (CAP_KPHASE / CAP_KSTAG in tn_dma_tile, gemm.hip) was measured from a working-tree patch and
# removed again because it made things worse (profiles/r03_experiments.log section 8 describes it); the script is kept for the record.
# k stagger of the tiles sharing a panel slice (CAP_KSTAG = K tiles per step): time and fabric read requests (x 128 B)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/exp_kstag; rm -rf $OUT; mkdir -p $OUT
cd $R
for s in ${STAGS:-0 1 2 4 8}; do
  echo "== CAP_KSTAG=$s"
  CAP_KSTAG=$s tools/gemm_bench.bin 32768 32768 1024 1 5
  CAP_KSTAG=$s tools/gemm_bench.bin 24576 24576 1024 1 5
  CAP_KSTAG=$s tools/gemm_bench.bin 8192 8192 8192 0 3
done
cd /tmp
for s in ${STAGS:-0 1 2 4 8}; do
  CAP_KSTAG=$s timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "dgemm_tn_dma" --output-format csv -d $OUT/s$s -o g -- $R/tools/gemm_bench.bin 24576 24576 1024 1 3 > $OUT/s$s.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*/")):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    a = {k: sum(v) / len(v) for k, v in acc.items()}
    print(d.split("/")[-2], "RDREQ x 128 B = %.2f GB per launch | hit rate %.3f" % (a.get("TCC_EA0_RDREQ_sum", 0) * 128 / 1e9, a.get("TCC_HIT_sum", 0) / max(1.0, a.get("TCC_HIT_sum", 0) + a.get("TCC_MISS_sum", 0))))
PY
cd $R
for s in ${FSTAGS:-0 1 2}; do
  echo "== factorization CAP_KSTAG=$s"
  CAP_KSTAG=$s timeout 200 tools/opt_bench.bin 32768 -1 3
  CAP_KSTAG=$s timeout 200 tools/opt_bench.bin 65536 -1 2
done
