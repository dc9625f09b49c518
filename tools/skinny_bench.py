"""The streaming kernels behind the refinement's substitutions (C[m x 8] -= op(A) B, gemm.hip dgemm_{tn,nn}_skinny_kernel) by themselves:
sweep shapes (K = 1024 rows of T) against divide-and-conquer shapes (K = m).   python tools/skinny_bench.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import _lib
L = _lib.lib()
N = 65536
T = torch.randn(N // 2, N, dtype=torch.float64, device="cuda")      # column-major N x N/2 window of a big operand, ld = N
B = torch.randn(8, N, dtype=torch.float64, device="cuda")           # N x 8, ld = N
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for ta, name in ((1, "TN (forward substitution)"), (0, "NN (backward substitution)")):
    for m, k in ((32768, 1024), (16384, 1024), (4096, 1024), (1024, 1024), (32768, 32768), (16384, 16384), (8192, 8192), (4096, 4096), (2048, 2048), (65536, 32768)):
        if k > N // 2 and ta == 0: continue
        # TN: A is k x m (ld N): rows of T = k <= N, columns m <= N/2.  NN: A is m x k: rows m <= N, columns k <= N/2
        if ta == 1 and m > N // 2: continue
        if ta == 0 and k > N // 2: continue
        f = lambda: L.cap_dgemm(ta, 0, m, 8, k, -1.0, T.data_ptr(), N, B.data_ptr(), N, 1.0, B.data_ptr() + 8 * (N // 2 if m <= N // 2 and k <= N // 2 else 0), N, None)
        if m + k > N: continue
        ms = t(f)
        print("%-28s m=%6d K=%6d: %8.3f ms  %7.1f GB/s" % (name, m, k, ms, 8.0 * m * k / ms / 1e6), flush=True)
