"""Kernel trace of tools/replay.py (one rank, --steps 1): the LAST factor call's per-kernel totals and, for its biggest bulk launches, what ran
beside them (summed duration inside the launch's window x min(1, workgroups / 256)).   python tools/r06_replay_trace.py trace.csv"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"]),
             int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1) // max(1, int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1))) for r in rows)
# a factor call starts with the A -> R copy (copy_rect) that follows an info / export kernel: take the last 1/3 of the trace that starts at a copy_rect_v2 of full height
bulk = [i for i, x in enumerate(ks) if x[3].startswith("dgemm_tn_dma_kernel<1") and x[4] >= 4096]
# the last factor call = the last maximal run of bulk launches with decreasing size: find the last launch that is larger than its predecessor
starts = [bulk[j] for j in range(len(bulk)) if j == 0 or ks[bulk[j]][4] > 4 * ks[bulk[j - 1]][4]]
i0 = starts[-1]
ks = ks[max(0, i0 - 40):]
t0, t1 = ks[0][0], max(x[1] for x in ks)
print("last factor call: about %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(ks)))
tot = defaultdict(lambda: [0, 0.0])
for s, e, q, n, g in ks:
    tot[n][0] += 1; tot[n][1] += (e - s) / 1e6
for n, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-46s %5d launches %8.2f ms" % (n, c, ms))
big = [x for x in ks if x[3].startswith("dgemm_tn_dma_kernel<1") and x[4] >= 2048]
for s, e, q, n, g in big[:40]:
    beside = defaultdict(float)
    for s2, e2, q2, n2, g2 in ks:
        if (s2, e2, q2) == (s, e, q) or e2 <= s or s2 >= e: continue
        beside[n2] += (min(e, e2) - max(s, s2)) / 1e3 * min(1.0, g2 / 256.0)
    top = sorted(beside.items(), key=lambda kv: -kv[1])[:5]
    print("  +%7.2f ms %6d wgs %8.3f ms | beside it %5.1f %% of a chip: %s" % ((s - t0) / 1e6, g, (e - s) / 1e6, 100.0 * sum(beside.values()) / ((e - s) / 1e3),
                                                                                ", ".join("%s %.0f us" % (a[:26], b) for a, b in top)))
