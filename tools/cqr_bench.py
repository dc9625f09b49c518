import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import cholinv, cacqr, validate
from capital_amd.matrix import matrix
m, n = (1 << 21), 256
A = matrix(n, m, 1, 1); A.distribute_random(0, 0, 1, 1, 0)
pack = cacqr.info(2, cholinv.info(1, 1, 0, 'U'))
cacqr.factor(A, pack, None); torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(3): cacqr.factor(A, pack, None)
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) / 3 * 1e-3
print("CholeskyQR2 %dx%d: %.2f ms  %.2f TF (4mn^2)  %.0f GB/s (6*8*m*n)" % (m, n, t * 1e3, 4.0 * m * n * n / t / 1e12, 48.0 * m * n / t / 1e9))
