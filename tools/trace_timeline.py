"""Timeline of one factor call out of a rocprofv3 kernel trace (bench_kernel_trace.csv): per queue busy time, and per bulk launch
(dgemm_tn_dma_kernel<1...>) its start / duration / gap to the previous bulk launch - shows where the bulk stream waits for the chain."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["LDS_Block_Size"])) for r in rows]
ks.sort()
# factor calls are delimited by copy_window (A -> R)
starts = [i for i, k in enumerate(ks) if k[3].startswith("copy_window")]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
i0 = starts[which]; i1 = starts[which + 1] if which + 1 < len(starts) and which != -1 else len(ks)
# drop the residual kernels after the factor (sumsq, Cijk)
seg = [k for k in ks[i0:i1] if not k[3].startswith(("Cijk", "sumsq", "at::", "fill_", "__amd"))]
t0 = seg[0][0]; tend = max(k[1] for k in seg)
print("factor: %.2f ms, %d kernels" % ((tend - t0) / 1e6, len(seg)))
busy = collections.defaultdict(float); cnt = collections.Counter()
for s, e, q, n, g, l in seg: busy[(q, n)] += (e - s) / 1e6; cnt[(q, n)] += 1
for (q, n), b in sorted(busy.items(), key=lambda x: -x[1]): print("  queue %d %-60s %5d launches %8.2f ms" % (q, n, cnt[(q, n)], b))
prev = None
print("bulk launches (queue of dgemm<1>): start ms | dur ms | gap since previous bulk end | wgs")
for s, e, q, n, g, l in seg:
    if n.startswith("dgemm_tn_dma_kernel<1"):
        print("  %8.2f %7.3f %7.3f %6d lds %d" % ((s - t0) / 1e6, (e - s) / 1e6, (s - prev) / 1e6 if prev else 0, g, l))
        prev = e
