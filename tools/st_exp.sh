#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd /tmp
for st in 8 4 16 2; do
  export CAP_ST=$st
  echo "== ST=$st"; $R/tools/gemm_bench.bin 32768 32768 1024 1 3
  rm -rf /tmp/pf; timeout 100 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "dgemm_tn_dma" --output-format csv -d /tmp/pf -o g -- $R/tools/gemm_bench.bin 32768 32768 1024 1 2 > /dev/null 2>&1
  python3 - <<'PY'
import csv,glob
rows=list(csv.DictReader(open(glob.glob('/tmp/pf/*counter_collection.csv')[0])))
v=[float(r['Counter_Value']) for r in rows if r['Counter_Name']=='FETCH_SIZE']
print("   FETCH_SIZE reported per launch: %.2f GB (x2 corrected %.2f GB)" % (sum(v)/len(v)/1e6*1.024, 2*sum(v)/len(v)/1e6*1.024))
PY
done
