#!/bin/bash
# kernel trace of ONE warm N = 65536 fp64 factorization; prints the launches between two consecutive big paired updates (one strip pair)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); cd /tmp
rm -rf /tmp/hl; rocprofv3 --kernel-trace --output-format csv -d /tmp/hl -o t -- python $R/tools/launch_curve.py 65536 > /tmp/hl.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/hl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in rows]
big = [i for i, k in enumerate(ks) if k[3].startswith("dgemm_tn_dma_kernel<1") and (k[1] - k[0]) > 60e6]
# second factorization: take the big launches of the last call
i0 = big[len(big) // 2 + 2]; i1 = big[len(big) // 2 + 3]
t0 = ks[i0][1]
print("between the end of one big paired update and the start of the next (%.2f ms):" % ((ks[i1][0] - t0) / 1e6))
for s, e, q, n, g in ks:
    if e >= t0 - 2e5 and s <= ks[i1][0] + 2e5 and (e - s) > 20e3:
        print("%9.3f ms +%8.3f ms q%-2d %-44s %6d wgs" % ((s - t0) / 1e6, (e - s) / 1e6, q, n, g))
PY
