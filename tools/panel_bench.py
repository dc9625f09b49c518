"""Time the panel chain in isolation: cholinv (mode 1) of small N (the diagonal-block factorization) and N=2048 mode -1."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import cholinv, validate
from capital_amd.matrix import matrix
for (n, ci) in [(64, 1), (128, 1), (256, 1), (512, 1), (1024, 1), (2048, -1), (4096, -1), (8192, -1)]:
    A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
    pack = cholinv.info(ci, 1, -2, 'U'); cholinv.factor(A, pack, None); torch.cuda.synchronize()
    reps = 20
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): cholinv.factor(A, pack, None)
    e.record(); torch.cuda.synchronize()
    print("cholinv n=%5d complete_inv=%2d: %8.1f us   residual %.2e" % (n, ci, s.elapsed_time(e) / reps * 1e3, validate.cholesky.residual(A, pack)))
