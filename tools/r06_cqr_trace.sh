export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/cq -o t -- python $GRAFT_REPO_ROOT/tools/cqr_bench.py 2>&1 | tail -1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/cq/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows:
    n = r["Kernel_Name"]
    if "fill_random" in n: continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s
    short = n.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:40]
    print("%9.3f ms  +%8.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, short))
PY
