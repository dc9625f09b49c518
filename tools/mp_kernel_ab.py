"""Mixed-precision factorization with either bf16 update kernel (option update_kernel 0 | 1), interleaved in one process:
factor time and the live profile of the big updates.   python tools/mp_kernel_ab.py [n] [tpw ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import mixed
from capital_amd.matrix import matrix
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
tpws = [int(x) for x in sys.argv[2:]] or [8]
A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
p = mixed.plan(n, 8)
def t(f, reps=2):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
reserves = [int(x) for x in os.environ.get("MP_RESERVE", "0").split(",")]
mins = [int(x) for x in os.environ.get("MP_MIN_TILES", "1024").split(",")]
for rnd in range(2):
    for kern, tpw, s3, res, mt in [(k, w, s3, r, mt) for r in reserves for k in (0, 1) for w in (tpws if k else [8]) for mt in (mins if k else [1024])
                                    for s3 in ((0, 1) if r == 0 and os.environ.get("MP_S3AB") else (1,))]:
        p.set_option("update_kernel", kern); p.set_option("update_tpw", tpw); p.set_option("solve3", s3); p.set_option("reserve", res)
        p.set_option("update_min_tiles", mt)
        tf = t(lambda: p.factor(A))
        nl, ms, fl, by = p.profile_update(A)
        print("N=%d kernel=%d tpw=%d solve3=%d reserve=%d min_tiles=%d: factor %.1f ms = %.1f TF-eq | big updates: %d launches %.1f ms %.0f TF (%.3f of 2.5 PF) %.0f GB/s | info %d"
              % (n, kern, tpw, s3, res, mt, tf * 1e3, n ** 3 / 3 / tf / 1e12, nl, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500, by / ms / 1e6, p.last_info()), flush=True)
p.set_option("update_kernel", 0); p.set_option("update_tpw", 8); p.set_option("reserve", 0)
