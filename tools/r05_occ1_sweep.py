"""Round 5: where should the bulk update switch to one workgroup per CU (option occ1_m)?  The chain trace (profiles/r05_chain_fp64_*.log)
shows the one-launch chain at its isolated speed (0.31 ms per 512-block) only where it has CUs to itself; next to fp64-MFMA bulk waves
every fp64 VALU operation of its leaf waits for the matrix pipe (3 ms per block).   python tools/r05_occ1_sweep.py n [occ1_m ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import cholinv
from capital_amd.matrix import matrix
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
vals = [int(x) for x in sys.argv[2:]] or [16384, 20480, 24576, 32768]
A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
pack = cholinv.info(-1, 1, -5, 'U')
cholinv.factor(A, pack, None); torch.cuda.synchronize()
def t(reps):
    cholinv.factor(A, pack, None); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): cholinv.factor(A, pack, None)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for rnd in range(2):
    for v in vals:
        pack.set_option("occ1_m", v)
        tf = t(3 if n <= 32768 else 2)
        print("N=%d occ1_m=%d: %.2f ms = %.2f TF | info %d" % (n, v, tf * 1e3, n ** 3 / 3 / tf / 1e12, pack.last_info()), flush=True)
