"""Round 6: mixed-precision factorization with the far bf16 bulk updates on the third-generation kernel (option update_kernel = 3 / 4,
csrc/bf16_tn3.hip) against the 128-tile kernel (0), interleaved in one process; factor time and the live profile of the bulk launches.
python tools/r06_mixed_v3.py [n] [min_tiles ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import mixed
from capital_amd.matrix import matrix
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
mins = [int(x) for x in sys.argv[2:]] or [256]
A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
p = mixed.plan(n, 8)
def t(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
ref = None
for rnd in range(2):
    for uk, mt, st in [(0, 0, 8), (5, 64, 1032), (6, 64, 1032)] + [(6, 64, 1000 + m) for m in mins]:
        p.set_option("update_kernel", uk)
        p.set_option("update_v3_head_min_tiles", st - 1000 if st >= 1000 else -1)
        if uk: p.set_option("update_v3_min_tiles", mt); p.set_option("update_v3_st", 4)
        tf = t(lambda: p.factor(A))
        nl, ms, fl, by = p.profile_update(A)
        print("N=%d update_kernel=%d min_tiles=%d st=%d: factor %.1f ms = %.1f TF-eq | bulk launches: %d, %.1f ms, %.0f TF (%.3f of 2.5 PF), %.0f GB/s | info %d"
              % (n, uk, mt, st, tf * 1e3, n ** 3 / 3 / tf / 1e12, nl, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500, by / ms / 1e6, p.last_info()), flush=True)
p.set_option("update_kernel", 0)
