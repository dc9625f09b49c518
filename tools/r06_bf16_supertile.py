import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from capital_amd import _lib, mixed
from capital_amd.matrix import matrix
L = _lib.lib()
m, K = 32768, 4096
a16 = torch.randn(m, K, device="cuda").to(torch.bfloat16); c = torch.zeros(m, m, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    for st in (1, 2, 3, 4, 8):
        ts = []
        for r in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rc = L.cap_bf16_update(6, m, m, K, -1.0, a16.data_ptr(), K, a16.data_ptr(), K, c.data_ptr(), m, 1, st, s); e1.record(); torch.cuda.synchronize()
            assert rc == 0
            if r: ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[1]
        print("alone m=%d K=%d st=%d: %.3f ms = %.0f TF" % (m, K, st, t, 2.0 * K * (m * (m + 1) / 2) / t / 1e9), flush=True)
del a16, c
n = 65536
A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
p = mixed.plan(n, 8)
def t(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for rep in range(2):
    for st in (2, 3, 4, 8):
        p.set_option("update_v3_st", st)
        print("factor N=%d update_v3_st=%d: %.1f ms" % (n, st, t(lambda: p.factor(A)) * 1e3), flush=True)
