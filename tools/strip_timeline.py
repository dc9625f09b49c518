"""All kernels between the k-th and the (k + span)-th diagonal-block chain of a rocprofv3 kernel trace: queue, name, start offset,
duration, workgroups - what the streams of the mixed-precision factorization do during one strip.
    python tools/strip_timeline.py trace.csv [k] [span]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
span = int(sys.argv[3]) if len(sys.argv) > 3 else 2
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"]),
             int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) // max(1, int(r["Workgroup_Size_X"]))) for r in rows)
chains = [i for i, x in enumerate(ks) if x[3].startswith(("chain64_coop", "leaf_cholinv"))]
i0, i1 = chains[k], chains[k + span]
t0 = ks[i0][0]
qs = sorted({x[2] for x in ks[i0:i1]})
print("window: %.1f us, queues %s" % ((ks[i1][0] - t0) / 1e3, qs))
for s, e, q, n, g in ks[i0:i1]:
    print("  %8.1f us  q%d %s%-44s %8.1f us  %6d wgs" % ((s - t0) / 1e3, q, "    " * qs.index(q), n, (e - s) / 1e3, g))
