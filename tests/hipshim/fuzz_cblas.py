"""TEST INFRASTRUCTURE: random CBLAS / LAPACKE calls twice - through MKL (the BLAS the reference is linked with; /opt/conda/lib/libmkl_rt.so,
one thread) and through libcapital_amd_cblas.so over the CPU stand-in (compute mode) - on the same host arrays: ragged shapes, leading
dimensions larger than the windows, every transpose / side, alpha / beta incl. 0 and 1, junk (NaN) wherever BLAS promises not to look, and
now and then an illegal argument (both must leave the output as it was).  python tests/hipshim/fuzz_cblas.py SEED COUNT"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
MKL = "/opt/conda/lib/libmkl_rt.so"
COL, NT, TR, UP, LO, NONUNIT, LEFT, RIGHT = 102, 111, 112, 121, 122, 131, 141, 142


def main(seed, count):
    os.environ.setdefault("MKL_NUM_THREADS", "1")
    sys.path.insert(0, HERE)
    import build_shim
    dst = build_shim.build_cblas()
    shim = C.CDLL(os.path.join(build_shim.OUT, "libhipshim.so"), mode=C.RTLD_GLOBAL)
    C.CDLL(os.path.join(build_shim.OUT, "libcapital_amd_shim.so"), mode=C.RTLD_GLOBAL)
    ours, mkl = C.CDLL(dst), C.CDLL(MKL)
    shim.shim_set_compute(1)
    for L in (ours, mkl):
        L.LAPACKE_dpotrf.restype = C.c_int; L.LAPACKE_dtrtri.restype = C.c_int
    rng = np.random.default_rng(seed)
    d = C.c_double
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    f = np.asfortranarray
    dim = lambda: int(rng.choice([rng.integers(1, 9), rng.integers(9, 70), rng.integers(70, 300)]))
    scal = lambda: float(rng.choice([0.0, 1.0, -1.0, 0.75, -2.5]))
    bad = 0
    for i in range(count):
        kind = rng.choice(["gemm", "gemm", "syrk", "trmm", "potrf", "trtri"])
        m, n, k = dim(), dim(), dim()
        pad = lambda: int(rng.integers(0, 4))
        outs = []
        if kind == "gemm":
            ta, tb = (int(rng.choice([NT, TR])) for _ in range(2))
            ar, ac = (m, k) if ta == NT else (k, m); br, bc = (k, n) if tb == NT else (n, k)
            a = f(np.full((ar + pad(), ac), np.nan)); a[:ar] = rng.standard_normal((ar, ac))
            b = f(np.full((br + pad(), bc), np.nan)); b[:br] = rng.standard_normal((br, bc))
            alpha, beta = scal(), scal()
            c0 = f(np.full((m + pad(), n), 5.5)); c0[:m] = np.nan if beta == 0.0 else rng.standard_normal((m, n))
            lda = a.shape[0] - (1 + ar if rng.random() < 0.04 else 0)             # now and then an illegal leading dimension
            for L in (ours, mkl):
                c = c0.copy(order="F")
                L.cblas_dgemm(COL, ta, tb, m, n, k, d(alpha), p(a), lda, p(b), b.shape[0], d(beta), p(c), c.shape[0]); outs.append(c)
            what = "dgemm %d %d m=%d n=%d k=%d alpha=%g beta=%g lda=%d" % (ta, tb, m, n, k, alpha, beta, lda)
        elif kind == "syrk":
            tr, uplo = int(rng.choice([NT, TR])), int(rng.choice([UP, LO]))
            ar, ac = (n, k) if tr == NT else (k, n)
            a = f(np.full((ar + pad(), ac), np.nan)); a[:ar] = rng.standard_normal((ar, ac))
            alpha, beta = scal(), scal()
            c0 = f(np.full((n + pad(), n), 5.5)); c0[:n] = rng.standard_normal((n, n))
            if beta == 0.0:
                c0[:n][np.triu_indices(n) if uplo == UP else np.tril_indices(n)] = np.nan
            for L in (ours, mkl):
                c = c0.copy(order="F")
                L.cblas_dsyrk(COL, uplo, tr, n, k, d(alpha), p(a), a.shape[0], d(beta), p(c), c.shape[0]); outs.append(c)
            what = "dsyrk uplo=%d trans=%d n=%d k=%d alpha=%g beta=%g" % (uplo, tr, n, k, alpha, beta)
        elif kind == "trmm":
            side, tr = int(rng.choice([LEFT, RIGHT])), int(rng.choice([NT, TR]))
            t = m if side == LEFT else n
            tm = f(np.full((t + pad(), t), np.nan)); tm[:t] = np.triu(rng.standard_normal((t, t))) + np.tril(np.full((t, t), np.nan), -1)
            alpha = scal()
            b0 = f(np.full((m + pad(), n), 5.5)); b0[:m] = rng.standard_normal((m, n))
            uplo = UP                                                                # (Lower is legal BLAS the library does not implement: it ABORTS with the reason instead of
                                                                                     #  returning the window untouched - tests/test_reference_offload.py checks that in a child process)
            for L in (ours, mkl):
                b = b0.copy(order="F")
                L.cblas_dtrmm(COL, side, uplo, tr, NONUNIT, m, n, d(alpha), p(tm), tm.shape[0], p(b), b.shape[0]); outs.append(b)
            what = "dtrmm side=%d uplo=%d trans=%d m=%d n=%d alpha=%g" % (side, uplo, tr, m, n, alpha)
        else:
            g = rng.standard_normal((n, n + 5)); s = g @ g.T / n + 0.2 * np.eye(n)
            if kind == "potrf" and rng.random() < 0.15:
                s[int(rng.integers(0, n)), :] *= -1.0; s = 0.5 * (s + s.T)           # not positive definite: same info, same partial factor?
            a0 = f(np.full((n + pad(), n), 5.5)); a0[:n] = np.triu(s) + np.tril(np.full((n, n), np.nan), -1)
            if kind == "trtri":
                a0[:n] = np.triu(np.linalg.cholesky(g @ g.T / n + 0.2 * np.eye(n)).T) + np.tril(np.full((n, n), np.nan), -1)
            infos = []
            for L in (ours, mkl):
                a = a0.copy(order="F")
                infos.append(L.LAPACKE_dpotrf(COL, C.c_char(b"U"), n, p(a), a.shape[0]) if kind == "potrf" else
                             L.LAPACKE_dtrtri(COL, C.c_char(b"U"), C.c_char(b"N"), n, p(a), a.shape[0]))
                outs.append(a)
            what = "%s n=%d info=%s" % (kind, n, infos)
            if infos[0] != infos[1]:
                bad += 1; print("BAD  %s: info differs" % what, flush=True); continue
            if infos[0] > 0:
                j = infos[0] - 1                                                    # LAPACK: the leading j x j block is factored; compare that
                outs = [o[:j, :j] for o in outs]
        x, y = outs
        same_nan = np.array_equal(np.isnan(x), np.isnan(y))
        num = np.linalg.norm(np.nan_to_num(x) - np.nan_to_num(y)); den = max(np.linalg.norm(np.nan_to_num(y)), 1e-300)
        if not same_nan or num / den > 1e-12:
            bad += 1; print("BAD  %s: %s, off by %.2e" % (what, "NaN pattern differs" if not same_nan else "values differ", num / den), flush=True)
        else:
            print("ok   %s  %.1e" % (what, num / den), flush=True)
    print("%d calls, %d with findings" % (count, bad))
    return bad


if __name__ == "__main__":
    if not os.path.exists(MKL):
        print("libmkl_rt.so is not here"); sys.exit(0)
    sys.exit(1 if main(int(sys.argv[1]), int(sys.argv[2])) else 0)
