"""Reads a rocprofv3 kernel trace of tools/mp_factor_only.py and prints, for the LAST factorization in it: per-kernel totals, and for every
big bf16 bulk launch its duration, its grid, and what ran beside it (per other kernel: summed duration inside the launch's window weighted by
min(1, workgroups / 256) - the share of the chip it can have taken).   python tools/r06_mp_trace.py trace.csv"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:40]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"]),
             int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1) // max(1, int(r["Workgroup_Size_X"]))) for r in rows)
# factorizations start with the fp32 import
imp = [i for i, x in enumerate(ks) if x[3].startswith("f64_to_f32_upper")]
i0 = imp[-1]
ks = ks[i0:]
t0, t1 = ks[0][0], max(x[1] for x in ks)
print("last factorization: %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(ks)))
tot = defaultdict(lambda: [0, 0.0])
for s, e, q, n, g in ks:
    tot[n][0] += 1; tot[n][1] += (e - s) / 1e6
for n, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-42s %5d launches %8.2f ms" % (n, c, ms))
bulk = [x for x in ks if x[3].startswith("bf16_tn") and x[4] >= 1024]
print("bulk launches (>= 1024 workgroups): %d, %.2f ms" % (len(bulk), sum(e - s for s, e, *_ in bulk) / 1e6))
for s, e, q, n, g in bulk:
    beside = defaultdict(float)
    for s2, e2, q2, n2, g2 in ks:
        if q2 == q or e2 <= s or s2 >= e: continue
        beside[n2] += (min(e, e2) - max(s, s2)) / 1e3 * min(1.0, g2 / 256.0)
    top = sorted(beside.items(), key=lambda kv: -kv[1])[:4]
    print("  +%7.2f ms %-18s %6d wgs %8.3f ms | chip-share beside it %5.1f %% : %s" % (
        (s - t0) / 1e6, n, g, (e - s) / 1e6, 100.0 * sum(beside.values()) / ((e - s) / 1e3), ", ".join("%s %.0f us" % (a[:22], b) for a, b in top)))
