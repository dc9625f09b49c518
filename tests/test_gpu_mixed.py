"""-m gpu: mixed-precision Cholesky solve (bf16 MFMA factor + fp64 refinement) against the fp64 path / NumPy."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import capital_oracle as orc  # noqa: E402
from tests.gpu_util import relerr  # noqa: E402


def _spd(n, kind, seed=0):
    if kind == "reference":                      # upstream's distribute_symmetric(diagonallyDominant): kappa ~ 1
        return orc.symmetric_global(n, True)
    rng = np.random.default_rng(seed)
    b = rng.standard_normal((n, n))
    return b @ b.T / n + (0.5 if kind == "gram" else 0.02) * np.eye(n)      # kappa ~ 10 / ~ 200


@pytest.mark.parametrize("n,kind,nrhs", [(128, "reference", 1), (1024, "reference", 3), (2048, "gram", 130), (1152, "gram", 7), (4096, "reference", 16),
                                         # n % 128 == 0 but not a power of two below 1024: panel width = the power of two below n
                                         (384, "reference", 2), (640, "gram", 5), (896, "reference", 1)])
def test_solve_reaches_fp64_accuracy(n, kind, nrhs):
    from capital_amd import mixed
    from capital_amd.matrix import matrix
    a = _spd(n, kind)
    rng = np.random.default_rng(n + nrhs)
    b = rng.standard_normal((n, nrhs))
    A = matrix(n, n, 1, 1).from_numpy(a); B = matrix(nrhs, n, 1, 1).from_numpy(b)
    p = mixed.plan(n, nrhs)
    p.factor(A)
    assert p.last_info() == 0
    # the low-precision factor: bf16 products, fp32 accumulation -> a few 1e-3 of the fp64 factor, upper triangular
    r32 = p.R32().cpu().numpy().astype(np.float64)
    ref = np.linalg.cholesky(a).T
    assert np.array_equal(np.tril(r32, -1), np.zeros_like(r32))
    assert 1e-9 < relerr(r32, ref) < 2e-2
    X, iters, rr = p.solve(A, B, max_iter=30, tol=1e-15)
    x = X.to_numpy(); xref = np.linalg.solve(a, b)
    assert rr <= 1e-14 and 1 <= iters <= 25, (rr, iters)
    assert np.linalg.norm(a @ x - b) / np.linalg.norm(b) < 1e-14
    assert relerr(x, xref) < 1e-12 * np.linalg.cond(a)
    # plan reuse: same factor, new right-hand side
    b2 = rng.standard_normal((n, nrhs)); B2 = matrix(nrhs, n, 1, 1).from_numpy(b2)
    X2, _, rr2 = p.solve(A, B2)
    assert rr2 <= 1e-14 and relerr(X2.to_numpy(), np.linalg.solve(a, b2)) < 1e-12 * np.linalg.cond(a)
    p.close()


@pytest.mark.parametrize("n", [3072, 4096, 5248])
def test_strips_of_two_panels_against_single_panels(n):
    """"strip" = 2 (default): the bf16 update contracts two panels (K = 2048) at once.  Odd panel counts (3072: the last strip is one
    panel), even ones, a ragged last panel (5248 = 5 x 1024 + 128): both forms give a usable factor and the same refined solution;
    "split" = 1 (default) runs the same updates on three streams and must not change a bit."""
    from capital_amd import mixed
    from capital_amd.matrix import matrix
    a = _spd(n, "gram", seed=n); b = np.random.default_rng(n).standard_normal((n, 4))
    A = matrix(n, n, 1, 1).from_numpy(a); B = matrix(4, n, 1, 1).from_numpy(b)
    xs, fs = [], []
    for strip, split in ((1, 0), (2, 0), (2, 1)):
        p = mixed.plan(n, 4); p.set_option("strip", strip); p.set_option("split", split)
        p.factor(A)
        assert p.last_info() == 0
        fs.append(p.R32().cpu().numpy().astype(np.float64))
        X, iters, rr = p.solve(A, B)
        assert rr <= 1e-14 and iters <= 25, (strip, rr, iters)
        xs.append(X.to_numpy())
        p.close()
    ref = np.linalg.cholesky(a).T
    assert relerr(fs[0], ref) < 2e-2 and relerr(fs[1], ref) < 2e-2
    assert relerr(fs[0], fs[1]) < 1e-2                                  # same algorithm, different bf16 accumulation grouping
    assert relerr(xs[0], xs[1]) < 1e-12 * np.linalg.cond(a)
    assert np.linalg.norm(a @ xs[1] - b) / np.linalg.norm(b) < 1e-14
    # column-split schedule (near / far columns on two streams): every element still receives the same updates in the same order
    assert np.array_equal(fs[1], fs[2]) and np.array_equal(xs[1], xs[2])


def test_matches_the_fp64_path_and_reports_failures():
    from capital_amd import blas, cholinv, mixed
    from capital_amd.matrix import matrix
    from tests.gpu_util import to_dev, to_host
    n, nrhs = 2048, 8
    a = _spd(n, "reference"); b = np.random.default_rng(1).standard_normal((n, nrhs))
    A = matrix(n, n, 1, 1).from_numpy(a); B = matrix(nrhs, n, 1, 1).from_numpy(b)
    # oracle: the fp64 factorization + two blocked TRSMs of this library
    pack = cholinv.info(-1, 1, -2, 'U'); cholinv.factor(A, pack, None)
    R = cholinv.construct_R(pack)
    Bd, Bv = to_dev(b)
    for trans in (1, 0):
        blas.engine._trsm(R.data(), Bd, n, nrhs, R.ld(), n, blas.ArgPack_trmm(blas.Order.AblasColumnMajor, blas.Side.AblasLeft, blas.UpLo.AblasUpper,
                                                                              blas.Transpose(trans), blas.Diag.AblasNonUnit, 1.0))
    x64 = to_host(Bv)
    p = mixed.plan(n, nrhs); p.factor(A)
    X, iters, rr = p.solve(A, B)
    assert relerr(X.to_numpy(), x64) < 1e-13 and rr < 1e-14
    # not positive definite: info reports the pivot like the fp64 path
    bad = a.copy(); bad[700, 700] = -1.0
    Ab = matrix(n, n, 1, 1).from_numpy(bad)
    p.factor(Ab)
    assert 0 < p.last_info() <= n
    # too ill-conditioned for a bf16 factor: refinement stalls and says so (relres stays large, no exception, no NaN claim of success)
    rng = np.random.default_rng(2)
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    hard = (q * np.logspace(0, -6, n)) @ q.T; hard = (hard + hard.T) / 2
    Ah = matrix(n, n, 1, 1).from_numpy(hard)
    p.factor(Ah)
    if p.last_info() == 0:
        _, it, rr = p.solve(Ah, B, max_iter=5)
        assert it == 5 and not (rr <= 1e-14)
    p.close()

