"""profiles/r03_traffic_bench_n65536.json (argv[2] overrides the round tag) from the PMC passes of tools/prof_round.sh (FETCH_SIZE / WRITE_SIZE of the trailing-update
kernel inside `python bench.py`), stamped with the hash of the kernel source it was measured on."""
import csv, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_round")
def avg(c):
    import glob
    cands = glob.glob(os.path.join(src, "pmc_%s" % c, "**", "*counter_collection.csv"), recursive=True) + \
        glob.glob(os.path.join(src, "summary", "*pmc_%s_trailing_kernel.csv" % c))
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(cands[0])) if r["Counter_Name"] == c]
    return sum(v) / len(v), len(v)
def algorithmic_bytes_per_launch(n=65536, nb=512, pair=True):
    """The plan's own schedule (cholinv.hip right_looking: strips of NB = 2 nb rows, nb-wide ones once n / 8 columns are left, look-ahead depth 2,
    round 5's paired far update) replayed for the launches of the trailing-update kernel: per launch 16 B (fp64 read + write) per element of the
    updated upper staircase + the strip's K x N rows once.  Reproduces the bench's launch count (139) and average flops per launch."""
    NB, tail = 2 * nb, n // 8
    bnd = [0]
    while bnd[-1] < n:
        bnd.append(min(n, bnd[-1] + (NB if n - bnd[-1] > tail else nb)))
    ns, L, deferred = len(bnd) - 1, [], False
    for k in range(ns):
        rows, m = bnd[k + 1] - bnd[k], n - bnd[k + 1]
        if m <= 0:
            break
        rows1 = bnd[k + 2] - bnd[k + 1]; m2 = m - rows1
        if m2 <= 0:
            continue
        rows2 = (bnd[k + 3] - bnd[k + 2]) if k + 3 <= ns else m2
        if rows2 < m2:
            L.append((rows2, m2, rows)); m3 = m2 - rows2
            if deferred:
                L.append((m3, m3, 2 * NB)); deferred = False
            elif pair and k % 2 == 0 and k + 4 < ns and rows == NB and rows1 == NB:
                L.append((bnd[k + 4] - bnd[k + 3], m3, rows)); deferred = True
            else:
                L.append((m3, m3, rows))
        else:
            L.append((m2, m2, rows))
    by = [16.0 * (M * N - M * (M - 1) / 2.0) + 8.0 * K * N for (M, N, K) in L]
    return sum(by) / len(L), len(L)


f, nf = avg("FETCH_SIZE"); w, nw = avg("WRITE_SIZE")
alg, nlaunch = algorithmic_bytes_per_launch(pair=os.environ.get("CAP_PAIR_REST", "1") != "0")
out = {"command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-include-regex 'dgemm_tn_dma_kernel<1' -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-check (tools/prof_round.sh, separate passes)",
       "config": {"n": 65536, "complete_inv": -1}, "kernel": "dgemm_tn_dma_kernel<1, false, 0, true, false>", "dispatches": nf,
       "FETCH_SIZE_KB_per_launch_reported": f, "WRITE_SIZE_KB_per_launch": w,
       "fetch_correction": "x2 (gfx950 rocprofv3 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section; calibrated on this kernel's own LDS-DMA pattern with tools/exp_fetchcal.sh: a product whose B operand can only be fetched once reads 0.571 GB reported for 1.074 GB true, TCC_EA0_RDREQ x 128 B = 1.141 GB); WRITE_SIZE is used as reported (it matches the algorithmic C-tile bytes 1:1)",
       "traffic_bytes_per_launch": (2 * f + w) * 1024.0, "algorithmic_bytes_per_launch": alg, "launches_per_factor_modelled": nlaunch,
       "kernel_src_sha16": hashlib.sha256(b"".join(open(os.path.join(ROOT, "capital_amd", "csrc", f), "rb").read()
                                                    for f in ("gemm.hip", "tile_dma.h"))).hexdigest()[:16]}
sys.path.insert(0, ROOT)
from capital_amd import build as _build      # the machine code of gemm.hip's kernels: what the counters were really collected on
out["kernel_text_md5"] = _build.device_text_md5("gemm.hip")
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_traffic_bench_n65536.json" % tag), "w"), indent=1)
print(out["traffic_bytes_per_launch"] / 1e9, "GB per launch,", out["traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"], "x algorithmic")
