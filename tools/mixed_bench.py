"""Mixed-precision Cholesky solve vs the fp64 path at one size: factor time, solve time, sweeps, accuracy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import cholinv, mixed
from capital_amd.matrix import matrix
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nrhs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
B = matrix(nrhs, n, 1, 1); B.distribute_random(0, 0, 1, 1, 7)
def t(f, reps=2):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
p = mixed.plan(n, nrhs)
tf = t(lambda: p.factor(A))
info = p.last_info()
res = {}
def solve(): res["x"] = p.solve(A, B, max_iter=30, tol=1e-15)
ts = t(solve, 1)
X, it, rr = res["x"]
pack = cholinv.info(-1, 1, -5, 'U')
t64 = t(lambda: cholinv.factor(A, pack, None))
print("N=%d nrhs=%d mixed: factor %.1f ms (%.1f TF on N^3/3; bf16 MFMA + fp64 panels), solve %.1f ms (%d refinement sweeps, relres %.2e), info %d | fp64 factor %.1f ms (%.1f TF) | factor speed-up %.2fx, factor+solve vs fp64 factor %.2fx"
      % (n, nrhs, tf * 1e3, n ** 3 / 3 / tf / 1e12, ts * 1e3, it, rr, info, t64 * 1e3, n ** 3 / 3 / t64 / 1e12, t64 / tf, t64 / (tf + ts)), flush=True)
