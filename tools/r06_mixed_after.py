"""Does the mixed-precision factorization slow down after other workloads ran in the same process (bench.py's default run: 156 ms against 134 alone)?"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import cholinv, cacqr, mixed
from capital_amd.matrix import matrix
def time_mixed(tag):
    n = 65536
    A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
    p = mixed.plan(n, 8)
    p.factor(A); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2): p.factor(A)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 2 * 1e3
    print("%-46s mixed factor %.1f ms | torch reserved %.1f GB" % (tag, ms, torch.cuda.memory_reserved() / 1e9), flush=True)
    del p, A; gc.collect(); torch.cuda.empty_cache()
def f64(n, ci):
    A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
    pack = cholinv.info(ci, 1, -5, 'U')
    for _ in range(2): cholinv.factor(A, pack, None)
    torch.cuda.synchronize(); del A, pack; gc.collect(); torch.cuda.empty_cache()
def cq():
    m, k = 1 << 21, 256
    Q = matrix(k, m, 1, 1); Q.distribute_random(0, 0, 1, 1, 0)
    qp = cacqr.info(2, cholinv.info(1, 1, 0, 'U'))
    for _ in range(3): cacqr.factor(Q, qp, None)
    torch.cuda.synchronize(); del Q, qp; gc.collect(); torch.cuda.empty_cache()
time_mixed("fresh process")
time_mixed("again")
f64(65536, -1); time_mixed("after fp64 N=65536 mode -1")
f64(32768, 0); time_mixed("after fp64 N=32768 complete_inv=0")
f64(32768, 1); time_mixed("after fp64 N=32768 complete_inv=1")
f64(65536, 0); time_mixed("after fp64 N=65536 complete_inv=0")
cq(); time_mixed("after CholeskyQR2")
