"""bf16 trailing update by itself: C32[m x m] -= S^T S (upper), S: K x m bf16 - both kernel generations, interleaved rounds in ONE process
(cdna_hip_programming.md rule 24), random data.   python tools/bf16_bench.py [m ...]   env: BF16_K (2048), BF16_ROUNDS (5)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capital_amd import _lib  # noqa: E402

L = _lib.lib()
K = int(os.environ.get("BF16_K", "2048"))
rounds = int(os.environ.get("BF16_ROUNDS", "5"))
sizes = [int(x) for x in sys.argv[1:]] or [16384, 32768, 49152]
for m in sizes:
    a16 = torch.randn(m, K, device="cuda").to(torch.bfloat16)
    c = torch.zeros(m, m, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    variants = [(0, 0, "v1 128x128"), (1, 8, "v2 tpw=8"), (3, 8, "v3 nst=3 st=8"), (4, 8, "v3 nst=4 st=8"), (3, 4, "v3 nst=3 st=4"), (5, 8, "v3w st=8"), (5, 4, "v3w st=4"), (6, 4, "v3x st=4"), (116, 8, "v2 pfC/2"), (117, 8, "v2 pfC/4"), (118, 8, "v2 pfC/8"), (104, 8, "v2 -mfma"),
                (120, 8, "v2 pfC/4 -mfma"), (101, 8, "v2 -atomics"),
                (301, 8, "v3 -dma"), (302, 8, "v3 -reads"), (303, 8, "v3 -dma -reads"), (304, 8, "v3 -mfma"), (308, 8, "v3 -epilogue"), (309, 8, "v3 -dma -epi"),
                (311, 8, "v3 mfma only"), (312, 8, "v3 dma+reads only"),
                (501, 4, "v3w -dma"), (502, 4, "v3w -reads"), (508, 4, "v3w -epilogue"), (509, 4, "v3w -dma -epi"), (511, 4, "v3w mfma only"), (512, 4, "v3w dma+reads only"),
                (608, 4, "v3x -epilogue"), (609, 4, "v3x -dma -epi"), (611, 4, "v3x mfma only"), (612, 4, "v3x dma+reads only")]
    if os.environ.get("BF16_V3_ONLY"):
        variants = [v for v in variants if v[0] in (0, 3, 5, 6) or v[0] >= 300]
    if os.environ.get("BF16_PLAIN"):
        variants = [v for v in variants if v[0] < 100]
    times = {v[2]: [] for v in variants}
    for r in range(rounds + 1):
        for (var, tpw, name) in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            st = L.cap_bf16_update(var, m, m, K, -1.0, a16.data_ptr(), K, a16.data_ptr(), K, c.data_ptr(), m, 1, tpw, s)
            e1.record(); torch.cuda.synchronize()
            if st != 0:
                continue
            if r:
                times[name].append(e0.elapsed_time(e1))
    fl = 2.0 * K * (m * (m + 1) / 2)
    for name, ts in times.items():
        if not ts:
            continue
        ts.sort()
        print("m=%d K=%d %-12s median %.3f ms = %.0f TF (%.3f of 2.5 PF)  min %.3f ms = %.0f TF" % (
            m, K, name, ts[len(ts) // 2], fl / ts[len(ts) // 2] / 1e9, fl / ts[len(ts) // 2] / 1e9 / 2500, ts[0], fl / ts[0] / 1e9), flush=True)
    del a16, c
    torch.cuda.empty_cache()
