"""N factorizations of the mixed-precision plan and nothing else (kernel traces): python tools/mp_factor_only.py [n] [reps]; options from
the environment (CAP_CHAIN_COOP ...)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import mixed
from capital_amd.matrix import matrix
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
p = mixed.plan(n, 8)
for kv in os.environ.get("MP_OPTIONS", "").split(","):      # e.g. MP_OPTIONS=update_kernel=3,pair_rest=1
    if "=" in kv: p.set_option(kv.split("=")[0], int(kv.split("=")[1]))
p.factor(A); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): p.factor(A)
torch.cuda.synchronize()
print("N=%d mixed factor %.1f ms, info %d" % (n, (time.perf_counter() - t0) / reps * 1e3, p.last_info()))
