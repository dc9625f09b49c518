"""Per (queue, kernel) totals of a rocprofv3 kernel trace csv: launches, summed duration, mean; and per queue the union of busy time.
    python tools/kernel_sums.py trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
agg = collections.defaultdict(lambda: [0, 0.0]); spans = collections.defaultdict(list)
for r in rows:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"])
    k = (q, short(r["Kernel_Name"])); agg[k][0] += 1; agg[k][1] += (e - s) / 1e6; spans[q].append((s, e))
for q, sp in sorted(spans.items()):
    sp.sort(); busy = 0; cs, ce = sp[0]
    for s, e in sp[1:]:
        if s > ce: busy += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    busy += ce - cs
    print("queue %d: %d launches, busy (union) %.2f ms, span %.2f ms" % (q, len(sp), busy / 1e6, (sp[-1][1] - sp[0][0]) / 1e6))
for (q, n), (c, ms) in sorted(agg.items(), key=lambda x: -x[1][1])[:40]:
    print("  queue %d %-70s %6d launches %9.2f ms  mean %8.1f us" % (q, n, c, ms, ms / c * 1e3))
