"""The multi-GPU schedule (csrc/dist.hip) run by ONE rank on one GPU, next to the single-GPU plan: same kernels, so the
difference is schedule quality (strips, look-ahead, copies) - the part of the multi-GPU path a 1-GPU box can measure."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import cholinv, dist_cholesky as dc, validate
from capital_amd.matrix import matrix

def timeit(f, reps=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

sizes = [int(x) for x in (sys.argv[1:] or ["16384", "32768"])]
force = os.environ.get("CAP_P1_RCCL") == "1"
for n in sizes:
    A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
    pack = cholinv.info(-1, 1, -5, 'U')
    t = timeit(lambda: cholinv.factor(A, pack, None))
    print("single-GPU plan      N=%d: %.1f ms  %.2f TF" % (n, t * 1e3, n ** 3 / 3 / t / 1e12), flush=True)
    del A, pack; torch.cuda.empty_cache()
    comm = dc.RcclComm(force_rccl=force)
    for (strip, d2) in ((2, 1), (2, 0), (1, 1), (1, 0)):
        ctx = dc.Context(n, 512, comm); ctx.fill_symmetric(True)
        ctx.set_option("strip", strip); ctx.set_option("depth2", d2)
        t = timeit(ctx.factor)
        info = ctx.last_info()
        print("dist plan P=1 %s N=%d strip=%d depth2=%d: %.1f ms  %.2f TF  info=%d" % ("rccl" if force else "self", n, strip, d2, t * 1e3, n ** 3 / 3 / t / 1e12, info), flush=True)
        ctx.close(); torch.cuda.empty_cache()
    comm.close()
