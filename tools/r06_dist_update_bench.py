"""The distributed trailing update (cap_dist_update_launch: staircase + gathered-A form of dgemm_tn_dma_kernel, csrc/gemm.hip) ALONE, in the
shapes rank p of a 1 x P plan gives it at N = 65536 (strip t: rows below strip t + 2, my block columns), one and two workgroups per CU, next to
the plain SYRK of the same flops.  Internal launcher, C++-mangled: a tool, not API.   python tools/r06_dist_update_bench.py [P] [p] [strips ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import _lib
L = _lib.lib()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = int(sys.argv[2]) if len(sys.argv) > 2 else 6
strips = [int(x) for x in sys.argv[3:]] or [0, 8, 16, 24, 32, 40, 48]
n, nb, q = 65536, 512, 2
nblk = n // nb
f = getattr(L, "_Z22cap_dist_update_launchlllPKdlPKiS0_PdliiiiiP12ihipStream_tiiii")
f.restype = C.c_int
f.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
              C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
lbfirst = lambda r, k: (k - r) // P + 1 if k >= r else 0
nblocks_of = lambda r: (nblk - 1 - r) // P + 1 if r < nblk else 0
st = torch.cuda.current_stream().cuda_stream
lc = nblocks_of(p) * nb
Cm = torch.randn(lc, n, dtype=torch.float64, device="cuda")          # my block columns, ld = n
ldS = q * nb
for t in strips:
    a = t * q; b = a + q - 1; e = b + 1; e2 = e + q; e3 = e2 + q
    if e3 >= nblk: break
    nmax = max((nblocks_of(r) - lbfirst(r, b)) * nb for r in range(P))
    piece = ldS * nmax
    G = torch.randn(P * piece, dtype=torch.float64, device="cuda")
    gstart = (C.c_int * 8)(*[lbfirst(r, b) if r < P else 0 for r in range(8)])
    lbe = lbfirst(p, b); lbe3 = lbfirst(p, e3 - 1)
    m = n - e3 * nb; ncols = (nblocks_of(p) - lbe3) * nb
    Bp = G.data_ptr() + 8 * (p * piece + (lbe3 - lbe) * nb * ldS)
    Cp = Cm.data_ptr() + 8 * (e3 * nb + lbe3 * nb * n)
    # algorithmic flops: 2 K per element of my part of the upper staircase
    elems = 0.0
    for lb in range(lbe3, nblocks_of(p)):
        J = lb * P + p
        rows_above = min(m, (J - e3) * nb)
        elems += rows_above * nb + (0.5 * nb * (nb + 1) if (J - e3) * nb < m else 0)
    fl = 2.0 * ldS * elems
    out = []
    for occ in (0, -1):
        def run():
            rc = f(m, ncols, ldS, G.data_ptr(), piece, gstart, Bp, Cp, n, P, p, nb, e3, lbe3, st, occ, 1, 0, 0)
            assert rc == 0, rc
        run(); torch.cuda.synchronize()
        s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(5): run()
        s1.record(); torch.cuda.synchronize()
        ms = s0.elapsed_time(s1) / 5
        out.append("%s %.3f ms %.1f TF" % ("2 wg/CU" if occ == 0 else "1 wg/CU", ms, fl / ms / 1e9))
    print("strip %2d: m = %5d rows x %4d local columns, K = %d, %.1f GFLOP | %s" % (t, m, ncols, ldS, fl / 1e9, " | ".join(out)), flush=True)
    del G
