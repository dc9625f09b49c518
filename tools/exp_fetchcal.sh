#!/bin/bash
# what does FETCH_SIZE count for this kernel's LDS-DMA reads?  One tile row (A slice: 1 MiB, L2-resident), 1024 tile columns, beta = 0:
# every B slice is read by exactly one workgroup exactly once -> the true fetch is 131072 x 1024 x 8 B = 1.074 GB (+ A per XCD).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/exp_fetchcal; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_RD[A-Z0-9_]*\|TCC_EA0_WR[A-Z0-9_]*\|TCC_BUBBLE[A-Z0-9_]*\|TCC_MISS[A-Z0-9_]*\|TCC_HIT[A-Z0-9_]*\|TCC_REQ[A-Z0-9_]*\|TCC_READ[A-Z0-9_]*" | sort -u | tr '\n' ' '; echo
ARGS="128 131072 1024 0 3"
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '+')
  BETA=0 timeout 120 rocprofv3 --pmc $c --kernel-include-regex "dgemm_tn_dma" --output-format csv -d $OUT/cal_$n -o g -- $R/tools/gemm_bench.bin $ARGS > $OUT/cal_$n.log 2>&1
done
ARGS="24576 24576 1024 1 3"
for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '+')
  timeout 120 rocprofv3 --pmc $c --kernel-include-regex "dgemm_tn_dma" --output-format csv -d $OUT/syrk_$n -o g -- $R/tools/gemm_bench.bin $ARGS > $OUT/syrk_$n.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*/")):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d.split("/")[-2], {k: "%.4g" % (sum(v) / len(v)) for k, v in acc.items()})
PY
tail -n 3 $OUT/*.log | grep -i "error\|fail" | head
