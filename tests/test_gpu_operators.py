"""-m gpu: operator seam (blas::engine / lapack::engine replacements) through the C ABI vs the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import capital_oracle as orc  # noqa: E402  (the checker)
from tests.gpu_util import DEV, relerr, to_dev, to_host  # noqa: E402


def _mods():
    from capital_amd import blas, lapack
    return blas, lapack


@pytest.mark.parametrize("ta,tb", [(1, 0), (0, 0), (1, 1), (0, 1)])
@pytest.mark.parametrize("m,n,k,pad", [(128, 128, 16, 0), (256, 384, 64, 0), (100, 37, 23, 3), (129, 257, 130, 1),
                                       (1, 1, 1, 0), (64, 200, 7, 0), (384, 128, 512, 0),
                                       # a few right-hand sides against a big operand: the skinny streaming kernels (ta = 1: MFMA reduction, ta = 0: K split in LDS)
                                       (2048, 8, 1024, 0), (1088, 3, 4104, 2), (4160, 5, 1000, 0), (1024, 1, 64, 0)])
def test_gemm_matches_oracle(ta, tb, m, n, k, pad):
    blas, _ = _mods()
    rng = np.random.default_rng(m * 1000 + n * 10 + k + ta * 2 + tb)
    a = rng.standard_normal((k, m) if ta else (m, k)); b = rng.standard_normal((n, k) if tb else (k, n))
    c = rng.standard_normal((m, n))
    A, _ = to_dev(a, a.shape[0] + pad); B, _ = to_dev(b, b.shape[0] + pad); C, Cv = to_dev(c, m + pad)
    pack = blas.ArgPack_gemm(blas.Order.AblasColumnMajor, blas.Transpose(ta), blas.Transpose(tb), -1.0, 1.0)
    blas.engine._gemm(A, B, C, m, n, k, a.shape[0] + pad, b.shape[0] + pad, m + pad, pack)
    ref = orc.gemm(a, b, c, ta, tb, -1.0, 1.0)
    assert relerr(to_host(Cv), ref) < 1e-14
    # beta = 0 must not propagate NaNs from an uninitialised C (BLAS semantics)
    C2 = torch.full_like(C, float("nan")); pack0 = blas.ArgPack_gemm(blas.Order.AblasColumnMajor, blas.Transpose(ta), blas.Transpose(tb), 2.0, 0.0)
    blas.engine._gemm(A, B, C2, m, n, k, a.shape[0] + pad, b.shape[0] + pad, m + pad, pack0)
    assert relerr(to_host(C2[:, :m].t()), orc.gemm(a, b, c, ta, tb, 2.0, 0.0)) < 1e-14


def test_gemm_empty_and_bad_args():
    blas, _ = _mods()
    from capital_amd import _lib
    A, _ = to_dev(np.ones((4, 4))); C, Cv = to_dev(np.ones((4, 4)))
    pack = blas.ArgPack_gemm(blas.Order.AblasColumnMajor, 0, 0, 1.0, 1.0)
    blas.engine._gemm(A, A, C, 0, 4, 4, 4, 4, 4, pack)       # m == 0: no-op
    blas.engine._gemm(A, A, C, 4, 4, 0, 4, 4, 4, pack)       # k == 0, beta == 1: C unchanged
    assert np.array_equal(to_host(Cv), np.ones((4, 4)))
    with pytest.raises(_lib.CapitalError):
        blas.engine._gemm(A, A, C, 4, 4, 4, 2, 4, 4, pack)   # lda < m
    with pytest.raises(_lib.CapitalError):
        blas.engine._gemm(torch.ones(4, 4, dtype=torch.float64), A, C, 4, 4, 4, 4, 4, 4, pack)  # CPU tensor: no CPU path


@pytest.mark.parametrize("n,k", [(256, 128), (1000, 77), (130, 512), (64, 4096), (256, 20000)])
def test_syrk_upper_trans(n, k):
    blas, _ = _mods()
    rng = np.random.default_rng(n + k)
    a = rng.standard_normal((k, n)); c = rng.standard_normal((n, n))
    A, _ = to_dev(a); C, Cv = to_dev(c)
    pack = blas.ArgPack_syrk(blas.Order.AblasColumnMajor, blas.UpLo.AblasUpper, blas.Transpose.AblasTrans, -1.0, 1.0)
    blas.engine._syrk(A, C, n, k, k, n, pack)
    out = to_host(Cv); ref = orc.syrk(a, c, True, True, -1.0, 1.0)
    assert relerr(np.triu(out), np.triu(ref)) < 1e-14
    assert np.array_equal(np.tril(out, -1), np.tril(c, -1)), "strictly lower triangle must not be touched"


def _spd(n, seed, cond_hard=False):
    rng = np.random.default_rng(seed)
    b = rng.standard_normal((n, n))
    return b.T @ b + (1e-6 if cond_hard else n) * np.eye(n)


@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 31, 33, 48, 63, 64, 65, 100, 128, 200, 256, 300, 513, 1024])
def test_potrf_upper(n):
    _, lapack = _mods()
    a = _spd(n, n)
    junk = a.copy(); junk[np.tril_indices(n, -1)] = 777.0       # lower triangle must be ignored and preserved
    A, Av = to_dev(junk, n + (n & 1))
    info = lapack.engine._potrf(A, n, n + (n & 1), lapack.ArgPack_potrf(lapack.Order.AlapackColumnMajor, lapack.UpLo.AlapackUpper))
    out = to_host(Av)
    ref, _ = orc.potrf_upper(junk)
    assert info == 0
    assert relerr(np.triu(out), np.triu(ref)) < 1e-13
    assert np.array_equal(np.tril(out, -1), np.tril(junk, -1))
    assert orc.cholesky_residual(a, np.triu(out)) < 1e-15


def test_potrf_ill_conditioned_and_not_spd():
    _, lapack = _mods()
    n = 96
    a = _spd(n, 5, cond_hard=True)                               # A = B^T B + 1e-6 I: kappa ~ 1e8
    A, Av = to_dev(a)
    pk = lapack.ArgPack_potrf(lapack.Order.AlapackColumnMajor, lapack.UpLo.AlapackUpper)
    assert lapack.engine._potrf(A, n, n, pk) == 0
    assert orc.cholesky_residual(a, np.triu(to_host(Av))) < 1e-14
    bad = _spd(n, 6); bad[40, 40] = -1.0                         # first failing pivot is 41 (1-based)
    B, _ = to_dev(bad)
    assert lapack.engine._potrf(B, n, n, pk) == 41
    assert orc.potrf_upper(bad)[1] != 0


@pytest.mark.parametrize("n", [1, 7, 16, 33, 64, 65, 129, 256, 500])
def test_trtri_upper(n):
    _, lapack = _mods()
    r = np.triu(np.random.default_rng(n).standard_normal((n, n))) + n * np.eye(n)
    junk = r.copy(); junk[np.tril_indices(n, -1)] = -5.0
    A, Av = to_dev(junk)
    lapack.engine._trtri(A, n, n, lapack.ArgPack_trtri(lapack.Order.AlapackColumnMajor, lapack.UpLo.AlapackUpper, lapack.Diag.AlapackNonUnit))
    out = to_host(Av); ref = orc.trtri_upper(junk)
    assert relerr(np.triu(out), np.triu(ref)) < 1e-13
    assert np.array_equal(np.tril(out, -1), np.tril(junk, -1))


# the three TRMM forms upstream issues (cholinv.hpp:118-121, :150-154; cacqr.hpp:24-25) + a real TRSM
@pytest.mark.parametrize("side,trans,alpha", [(0, 1, 1.0), (0, 0, 1.0), (1, 0, -1.0)])
@pytest.mark.parametrize("m,n", [(64, 64), (128, 200), (100, 37), (257, 129)])
def test_trmm_and_trsm(side, trans, alpha, m, n):
    blas, _ = _mods()
    td = m if side == 0 else n
    rng = np.random.default_rng(m + n + side + trans)
    t = rng.standard_normal((td, td)) + td * np.eye(td); t[np.tril_indices(td, -1)] = 9.0   # garbage below the diagonal
    b = rng.standard_normal((m, n))
    T, _ = to_dev(t); B, Bv = to_dev(b)
    pack = blas.ArgPack_trmm(blas.Order.AblasColumnMajor, blas.Side(side), blas.UpLo.AblasUpper, blas.Transpose(trans), blas.Diag.AblasNonUnit, alpha)
    blas.engine._trmm(T, B, m, n, td, m, pack)
    assert relerr(to_host(Bv), orc.trmm(t, b, side == 0, True, bool(trans), alpha)) < 1e-14
    B2, B2v = to_dev(b)
    blas.engine._trsm(T, B2, m, n, td, m, pack)
    assert relerr(to_host(B2v), orc.trsm(t, b, side == 0, True, bool(trans), alpha)) < 1e-12


@pytest.mark.parametrize("side,trans", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("td,other,alpha", [(700, 300, 1.0), (1536, 520, -0.5), (2304, 130, 2.0)])
def test_trsm_blocked_substitution_all_forms(side, trans, td, other, alpha):
    """cap_dtrsm = blocked substitution (diagonal-block inverses + GEMM updates): all four side/trans forms on
    several-block triangles with ragged last blocks, against scipy.linalg.solve_triangular."""
    import scipy.linalg as sla
    blas, _ = _mods()
    rng = np.random.default_rng(td + other + 2 * side + trans)
    t = np.triu(rng.standard_normal((td, td))) / np.sqrt(td) + 2.0 * np.eye(td)
    tj = t.copy(); tj[np.tril_indices(td, -1)] = -7.0            # the other triangle is not referenced
    m, n = (td, other) if side == 0 else (other, td)
    b = rng.standard_normal((m, n))
    if side == 0:
        ref = sla.solve_triangular(t, alpha * b, trans=trans, lower=False)
    else:     # X op(T) = alpha B  <=>  op(T)^T X^T = alpha B^T
        ref = sla.solve_triangular(t, alpha * b.T, trans=1 - trans, lower=False).T
    T, _ = to_dev(tj); B, Bv = to_dev(b)
    pack = blas.ArgPack_trmm(blas.Order.AblasColumnMajor, blas.Side(side), blas.UpLo.AblasUpper, blas.Transpose(trans), blas.Diag.AblasNonUnit, alpha)
    blas.engine._trsm(T, B, m, n, td, m, pack)
    x = to_host(Bv)
    assert relerr(x, ref) < 1e-12
    # backward error of the solve itself
    r = (t.T if trans else t) @ x - alpha * b if side == 0 else x @ (t.T if trans else t) - alpha * b
    assert np.linalg.norm(r) / (np.linalg.norm(t) * np.linalg.norm(x)) < 1e-14


@pytest.mark.parametrize("kappa", [1e4, 1e8])
def test_trsm_ill_conditioned_triangle(kappa):
    """kappa(T) ~ 1e4 / 1e8 (graded singular values): the blocked substitution must stay backward stable - the
    residual ||T x - b|| / (||T|| ||x||) at the 1e-14 level and the forward error within kappa * eps of scipy's."""
    import scipy.linalg as sla
    blas, _ = _mods()
    n, nrhs = 1024, 64
    rng = np.random.default_rng(int(np.log10(kappa)))
    q1, _ = np.linalg.qr(rng.standard_normal((n, n))); q2, _ = np.linalg.qr(rng.standard_normal((n, n)))
    a = q1 @ np.diag(np.logspace(0, -np.log10(kappa), n)) @ q2
    t = np.triu(np.linalg.qr(a)[1])
    t = t * np.sign(np.diag(t))[:, None]
    assert 0.1 * kappa < np.linalg.cond(t) < 10 * kappa
    b = rng.standard_normal((n, nrhs))
    for trans in (0, 1):
        ref = sla.solve_triangular(t, b, trans=trans, lower=False)
        T, _ = to_dev(t); B, Bv = to_dev(b)
        pack = blas.ArgPack_trmm(blas.Order.AblasColumnMajor, blas.Side(0), blas.UpLo.AblasUpper, blas.Transpose(trans), blas.Diag.AblasNonUnit, 1.0)
        blas.engine._trsm(T, B, n, nrhs, n, n, pack)
        x = to_host(Bv)
        op = t.T if trans else t
        assert np.linalg.norm(op @ x - b) / (np.linalg.norm(t) * np.linalg.norm(x)) < 5e-14
        assert relerr(x, ref) < 50 * kappa * np.finfo(float).eps


def test_trmm_hinted_paths_large():
    """public _trmm on operands big enough for the MFMA tile kernel with the triangular K-range hints (tags 8/16/32)."""
    blas, _ = _mods()
    rng = np.random.default_rng(5)
    for side, trans, m, n in ((0, 0, 1024, 640), (0, 1, 1152, 512), (1, 0, 768, 1280)):
        td = m if side == 0 else n
        t = rng.standard_normal((td, td)); t[np.tril_indices(td, -1)] = 3.0
        b = rng.standard_normal((m, n))
        T, _ = to_dev(t); B, Bv = to_dev(b)
        pack = blas.ArgPack_trmm(blas.Order.AblasColumnMajor, blas.Side(side), blas.UpLo.AblasUpper, blas.Transpose(trans), blas.Diag.AblasNonUnit, 1.5)
        blas.engine._trmm(T, B, m, n, td, m, pack)
        assert relerr(to_host(Bv), orc.trmm(t, b, side == 0, True, bool(trans), 1.5)) < 1e-14
