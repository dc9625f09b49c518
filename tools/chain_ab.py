"""One-launch diagonal-block chain (option "chain_coop" = resident workgroups, 0 = one launch per step) against the stepwise chain,
interleaved in one process: the chain alone on an idle GPU, the mixed-precision factorization, the fp64 factorization.
    python tools/chain_ab.py [G ...]          env CHAIN_N64 (fp64 sizes, default "32768"), CHAIN_NMP (mixed size, default 65536)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from capital_amd import cholinv, mixed
from capital_amd.matrix import matrix
Gs = [int(x) for x in sys.argv[1:]] or [0, 16, 32, 64]
def t(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
# 1. the chain alone: one 1024 / 512 block, factor + inverse
for nb in (512, 1024):
    A = matrix(nb, nb, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
    pack = cholinv.info(1, 1, -2, 'U'); pack.set_option("nb", nb)
    base = None
    for G in Gs:
        pack.set_option("chain_coop", G)
        dt = t(lambda: cholinv.factor(A, pack, None), 20)
        R = cholinv.construct_R(pack).to_numpy()
        if base is None: base = R
        print("alone nb=%d G=%d: %.1f us per cholinv, bitwise equal to G=%d: %s" % (nb, G, dt * 1e6, Gs[0], np.array_equal(R, base)), flush=True)
    pack._release()
# 2. mixed precision
n = int(os.environ.get("CHAIN_NMP", "65536"))
if n > 0:
    A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
    p = mixed.plan(n, 8)
    for rnd in range(2):
        for G in Gs:
            p.set_option("chain_coop", G)
            tf = t(lambda: p.factor(A), 2)
            nl, ms, fl, by = p.profile_update(A)
            print("mixed N=%d G=%d: factor %.1f ms = %.1f TF-eq | big updates %d launches %.1f ms %.0f TF (%.3f) | info %d"
                  % (n, G, tf * 1e3, n ** 3 / 3 / tf / 1e12, nl, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500, p.last_info()), flush=True)
    p.set_option("chain_coop", Gs[0]); p.close(); del A, p
    torch.cuda.empty_cache()
# 3. fp64
for n in [int(x) for x in os.environ.get("CHAIN_N64", "32768").split(",") if x]:
    A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
    pack = cholinv.info(-1, 1, -2, 'U')
    cholinv.factor(A, pack, None)
    for rnd in range(2):
        for G in Gs:
            pack.set_option("chain_coop", G)
            tf = t(lambda: cholinv.factor(A, pack, None), 3)
            print("fp64 N=%d G=%d: %.1f ms = %.2f TF | info %d" % (n, G, tf * 1e3, n ** 3 / 3 / tf / 1e12, pack.last_info()), flush=True)
    pack.set_option("chain_coop", Gs[0]); pack._release(); del A
    torch.cuda.empty_cache()
