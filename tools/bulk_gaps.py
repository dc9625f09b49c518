"""The big launches of one queue of a rocprofv3 kernel trace (default: kernels longer than 300 us): start, duration, gap since the end of
the previous one, and what the other queues ran inside each gap.   python tools/bulk_gaps.py trace.csv [min_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:40]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"])) for r in rows)
big = [x for x in ks if x[3].startswith("bf16_tn_kernel") and (x[1] - x[0]) / 1e3 >= min_us]
qbig = max(set(x[2] for x in big), key=lambda q: sum(x[1] - x[0] for x in big if x[2] == q))
big = [x for x in big if x[2] == qbig]
t0 = big[0][0]; prev = None
for s, e, q, n in big:
    line = "%9.1f us  %-20s %8.1f us" % ((s - t0) / 1e3, n, (e - s) / 1e3)
    if prev is not None:
        gap = (s - prev) / 1e3
        inside = [x for x in ks if x[2] != qbig and x[0] < s and x[1] > prev]
        line += "   gap %7.1f us" % gap
        if gap > 30:
            line += "   in it: " + ", ".join("q%d %s %.0f+%.0f" % (x[2], x[3][:18], (x[0] - prev) / 1e3, (x[1] - x[0]) / 1e3) for x in inside[:8])
    print(line); prev = e
