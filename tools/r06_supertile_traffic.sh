#!/bin/bash
# experiment build: the bulk SYRK shape alone (tools/syrk_alone.py, first shape only) with supertile edges CAP_ST = 4 / 8 / 16: time and HBM-side bytes
# (separate --pmc passes, FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE, KiB units -> GB)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); cd /tmp
for st in ${STS:-4 8 16}; do
  echo "== CAP_ST=$st"
  CAP_ST=$st SYRK_ONLY=1 python $R/tools/syrk_alone.py 2>&1 | grep SYRK
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm; CAP_ST=$st SYRK_ONLY=1 rocprofv3 --pmc $c --kernel-include-regex "dgemm_tn_dma_kernel" --output-format csv -d /tmp/pm -o p -- python $R/tools/syrk_alone.py > /dev/null 2>&1
    python3 - "$c" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == sys.argv[1]]
if v: print("   %s per launch: %.2f GB%s (%d launches)" % (sys.argv[1], sum(v) / len(v) * 1024 / 1e9 * (2 if sys.argv[1] == "FETCH_SIZE" else 1), " (x2 corrected)" if sys.argv[1] == "FETCH_SIZE" else "", len(v)))
PY
  done
done
