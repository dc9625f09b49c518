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
    # R and R^-1 through the distributed plan (all-gather of R = a copy at P = 1, then the local block substitution)
    for ci in (1, 0):
        ctx = dc.Context(n, 512, comm); ctx.fill_symmetric(True)
        ctx.set_option("complete_inv", ci)
        t = timeit(ctx.factor)
        fl = n ** 3 / 3.0 + (n ** 3 / 3.0 if ci == 1 else n ** 3 / 12.0)
        print("dist plan P=1 N=%d complete_inv=%d (factor + distributed-style inverse): %.1f ms  %.2f TF on N^3/3, %.2f TF true" % (n, ci, t * 1e3, n ** 3 / 3 / t / 1e12, fl / t / 1e12), flush=True)
        ctx.close(); torch.cuda.empty_cache()
    # the Pr x Pc plan on a 1 x 1 grid: the round-3 schedule (strip = 1 block row, look-ahead depth 1) and the round-4 one
    for (strip, d2) in ((1, 0), (2, 0), (2, 1)):
        c2 = dc.Context2D(n, 512, comm, 1); c2.fill_symmetric(True)
        c2.set_option("strip", strip); c2.set_option("depth2", d2)
        t = timeit(c2.factor)
        print("2D plan 1x1 N=%d strip=%d depth2=%d: %.1f ms  %.2f TF  info=%d launches=%s" % (n, strip, d2, t * 1e3, n ** 3 / 3 / t / 1e12, c2.last_info(), c2.launch_counts()), flush=True)
        c2.close(); torch.cuda.empty_cache()
    for ci in (1, 0):
        c2 = dc.Context2D(n, 512, comm, 1); c2.fill_symmetric(True)
        c2.set_option("complete_inv", ci)
        t = timeit(c2.factor)
        fl = n ** 3 / 3.0 + (n ** 3 / 3.0 if ci == 1 else n ** 3 / 12.0)
        print("2D plan 1x1 N=%d complete_inv=%d: %.1f ms  %.2f TF on N^3/3, %.2f TF true" % (n, ci, t * 1e3, n ** 3 / 3 / t / 1e12, fl / t / 1e12), flush=True)
        c2.close(); torch.cuda.empty_cache()
    comm.close()
