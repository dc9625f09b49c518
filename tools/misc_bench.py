"""Round-1 measurements beyond the headline: reference-semantics cholinv modes and CholeskyQR2 (1 GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import cholinv, cacqr, validate
from capital_amd.matrix import matrix
def timeit(f, reps=3):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3
for n in (16384, 32768):
    A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
    for ci in (0, 1):
        pack = cholinv.info(ci, 1, -3, 'U')
        t = timeit(lambda: cholinv.factor(A, pack, None))
        print("cholinv N=%d complete_inv=%d (upstream semantics, R and R^-1): %.1f ms  %.2f TF (N^3/3)  residual %.2e" % (n, ci, t * 1e3, n ** 3 / 3 / t / 1e12, validate.cholesky.residual(A, pack)))
        del pack
    del A
    torch.cuda.empty_cache()
for (m, n) in [(1 << 21, 256), (1 << 20, 512), (1 << 22, 128)]:
    A = matrix(n, m, 1, 1); A.distribute_random(0, 0, 1, 1, 0)
    pack = cacqr.info(2, cholinv.info(1, 1, 0, 'U'))
    t = timeit(lambda: cacqr.factor(A, pack, None))
    fl = 4.0 * m * n * n
    by = 6.0 * 8 * m * n
    print("CholeskyQR2 %dx%d: %.2f ms  %.2f TF (4mn^2)  %.0f GB/s algorithmic (6*8*m*n)  residual %.2e orth %.2e" % (m, n, t * 1e3, fl / t / 1e12, by / t / 1e9, validate.qr.residual(A, pack), validate.qr.orthogonality(A, pack)))
    del A, pack
    torch.cuda.empty_cache()
