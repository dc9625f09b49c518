#!/bin/bash
# NOTE: the strip-buffer padding (CAP_SB_PAD) part needs a removed working-tree patch; the LDPAD / gemm_bench part runs as is.
# leading dimension of the K-contiguous panel operands (row stride 8 KiB at ld = 1024: every row of a K tile in the same L2 channel?)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/exp_ldpad; rm -rf $OUT; mkdir -p $OUT
cd $R
for pad in 0 16 32 48 272; do
  LDPAD=$pad tools/gemm_bench.bin 32768 32768 1024 1 5
  LDPAD=$pad tools/gemm_bench.bin 24576 24576 1024 1 5
done
cd /tmp
pmc() { # name, env..., args
  name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    env "$@" timeout 120 rocprofv3 --pmc $c --kernel-include-regex "dgemm_tn_dma" --output-format csv -d $OUT/${name}_$c -o g -- $R/tools/gemm_bench.bin $ARGS > $OUT/${name}_$c.log 2>&1
  done
}
ARGS="24576 24576 1024 1 3"
pmc pad0 LDPAD=0; pmc pad16 LDPAD=16; pmc pad32 LDPAD=32; pmc pad272 LDPAD=272
# calibration: 8 tile rows x 512 tile columns, K = 1024: B is needed once (0.537 GB), A once (8 MiB), C once (0.537 GB read + write)
ARGS="1024 65536 1024 0 3"
pmc cal_atomic LDPAD=0; pmc cal_loadstore LDPAD=0 CAP_ATOMIC_C=0; pmc cal_atomic_pad16 LDPAD=16
python3 - <<PY
import csv, glob
for name in ("pad0", "pad16", "pad32", "pad272", "cal_atomic", "cal_loadstore", "cal_atomic_pad16"):
    o = []
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob("$OUT/%s_%s/**/*counter_collection.csv" % (name, c), recursive=True)
        if not fs: o.append("%s: no csv" % c); continue
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == c]
        o.append("%s %.3f GB reported (%d launches)" % (c, sum(v) / len(v) * 1024 / 1e9, len(v)))
    print(name, " | ".join(o))
PY
cd $R
for pad in 0 16; do
  echo "== factorization CAP_SB_PAD=$pad"
  CAP_SB_PAD=$pad timeout 200 tools/opt_bench.bin 32768 -1 3
  CAP_SB_PAD=$pad timeout 200 tools/opt_bench.bin 65536 -1 2
done
