"""Round 5: mixed-precision factorization with the far part of the bf16 bulk update taking two strips at a time (option pair_rest:
K = 4096 instead of 2048), interleaved in one process; factor time and the live profile of the bulk launches.   python tools/r05_mixed_pair.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import mixed
from capital_amd.matrix import matrix
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
p = mixed.plan(n, 8)
def t(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for rnd in range(3):
    for pr in (0, 1):
        p.set_option("pair_rest", pr)
        tf = t(lambda: p.factor(A))
        nl, ms, fl, by = p.profile_update(A)
        print("N=%d pair_rest=%d: factor %.1f ms = %.1f TF-eq | bulk launches: %d, %.1f ms, %.0f TF (%.3f of 2.5 PF), %.0f GB/s | info %d"
              % (n, pr, tf * 1e3, n ** 3 / 3 / tf / 1e12, nl, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500, by / ms / 1e6, p.last_info()), flush=True)
p.set_option("pair_rest", 1)
