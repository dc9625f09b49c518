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
    # kappa from the factor (2-norm condition of SPD a = (largest / smallest singular value of R)^2; an SVD of `a` itself cost 12 s at n = 4096)
    sv = np.linalg.svd(ref, compute_uv=False) if n <= 1536 else None
    kappa = (sv[0] / sv[-1]) ** 2 if sv is not None else {"reference": 4.0, "gram": 40.0}[kind]      # (measured bounds of the two generators)
    assert relerr(x, xref) < 1e-12 * kappa
    # plan reuse: same factor, new right-hand side
    b2 = rng.standard_normal((n, nrhs)); B2 = matrix(nrhs, n, 1, 1).from_numpy(b2)
    X2, _, rr2 = p.solve(A, B2)
    assert rr2 <= 1e-14 and relerr(X2.to_numpy(), np.linalg.solve(a, b2)) < 1e-12 * kappa
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
        p.set_option("pair_rest", 0)                                    # (the paired far update regroups sums: its own test below)
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
    assert relerr(xs[0], xs[1]) < 1e-10                                 # (kappa ~ 10 for this input: an SVD of the 5248^2 matrix just for the bound cost 10 s)
    assert np.linalg.norm(a @ xs[1] - b) / np.linalg.norm(b) < 1e-14
    # column-split schedule (near / far columns on two streams): every element still receives the same updates in the same order
    assert np.array_equal(fs[1], fs[2]) and np.array_equal(xs[1], xs[2])


def test_paired_far_update_of_the_split_schedule():
    """Option "pair_rest" (round 5, split schedule): the region below strip t + 2 takes the updates of the strips t and t + 1 in ONE bf16
    product with K = 4096 (the two strips are the halves of a pair buffer) instead of two with K = 2048.  n = 9216 = 9 panels: strips 0 and 1
    pair up (one more head on strip 2's rows, then the K = 4096 launch on the 3072 rows below), the strips behind them run unpaired, the last
    one is a single panel.  Same updates in another grouping: the factors agree at bf16-accumulation level, both are usable factors and
    both refine to the same fp64 solution - a region that missed an update would leave the refinement diverging."""
    from capital_amd import mixed
    from capital_amd.matrix import matrix
    n = 9216
    a = _spd(n, "gram", seed=7); b = np.random.default_rng(3).standard_normal((n, 4))      # kappa ~ 10
    A = matrix(n, n, 1, 1).from_numpy(a); B = matrix(4, n, 1, 1).from_numpy(b)
    fs, xs = [], []
    for pr in (0, 1):
        p = mixed.plan(n, 4); p.set_option("pair_rest", pr)
        for rep in range(2):                                            # plan reuse: the pair buffers are recycled
            p.factor(A)
        assert p.last_info() == 0
        fs.append(p.R32().cpu().numpy().astype(np.float64))
        X, iters, rr = p.solve(A, B)
        assert rr <= 1e-14 and iters <= 25, (pr, rr, iters)
        xs.append(X.to_numpy())
        p.close()
    ref = np.linalg.cholesky(a).T
    assert relerr(fs[0], ref) < 2e-2 and relerr(fs[1], ref) < 2e-2
    assert 0 < relerr(fs[0], fs[1]) < 1e-2                              # regrouped sums: close, and not identical (else the paired launch never ran)
    assert relerr(xs[0], xs[1]) < 1e-10
    assert np.linalg.norm(a @ xs[1] - b) / np.linalg.norm(b) < 1e-14


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



def _bf16_update(variant, a16, b16, c32, alpha, tri, tpw=0):
    """C32 += alpha A^T B through the C ABI (cap_bf16_update); a16: [m, k], b16: [n, k] bf16 (K-contiguous), c32: [n, m] fp32 buffer of C^T."""
    from capital_amd import _lib
    m, k = a16.shape; n = b16.shape[0]
    st = _lib.lib().cap_bf16_update(variant, m, n, k, float(alpha), a16.data_ptr(), a16.stride(0), b16.data_ptr(), b16.stride(0), c32.data_ptr(),
                                    c32.stride(0), int(tri), int(tpw), torch.cuda.current_stream().cuda_stream)
    return st


@pytest.mark.parametrize("m,n,k,tri,tpw", [(256, 128, 64, 0, 0), (512, 256, 128, 0, 1), (1024, 1024, 192, 1, 0), (2048, 2048, 512, 1, 3), (2304, 2304, 64, 1, 2),
                                           (2048, 1152, 1024, 0, 8), (4096, 4096, 2048, 1, 8), (3072, 3072, 256, 1, 1), (1280, 4224, 320, 0, 5)])
def test_bf16_update_kernels_against_torch(m, n, k, tri, tpw):
    """Both generations of the bf16 trailing-update kernel (csrc/mixed.hip) against torch's fp32 matmul of the same bf16 operands:
    every tile shape of the walk (supertile edges, diagonal tiles of the symmetric update, chunks of 1 .. 8 steps per workgroup,
    K tiles 1 .. 32 so that the three-deep LDS ring wraps, fills and drains), untouched strictly-lower part, padded leading dimensions.
    Not symmetric data: a transposed fragment or a swapped operand shows."""
    g = torch.Generator(device="cuda"); g.manual_seed(m + n + k)
    lda = k + 64                                                  # padded K stride (multiple of 8)
    abuf = torch.randn(m, lda, device="cuda", generator=g).to(torch.bfloat16); a16 = abuf[:, :k]
    if tri:
        b16 = a16
    else:
        bbuf = torch.randn(n, lda, device="cuda", generator=g).to(torch.bfloat16); b16 = bbuf[:, :k]
    c0 = torch.randn(n, m + 32, device="cuda", generator=g)      # C^T buffer: [col, row], ldc = m + 32
    ref = c0[:, :m].t().double() - 0.5 * (a16.double() @ b16.double().t())        # [m, n]
    mask = torch.triu(torch.ones(m, n, dtype=torch.bool, device="cuda")) if tri else torch.ones(m, n, dtype=torch.bool, device="cuda")
    scale = float(ref.abs().max())
    outs = []
    for variant in (0, 1):
        c = c0.clone()
        st = _bf16_update(variant, a16, b16, c, -0.5, tri, tpw)
        if variant == 0 and st != 0:
            assert m % 128 or n % 128 or k % 64, st           # the 128-tile kernel only refuses shapes outside its granularity
            continue
        assert st == 0, st
        torch.cuda.synchronize()
        got = c[:, :m].t().double()
        err = float((got - ref)[mask].abs().max()) / scale
        # fp32 accumulation of exact bf16 products: k / 16 chained MFMA accumulations of partial sums that grow like sqrt(k), against a
        # scale that grows the same way -> the relative error grows like sqrt(k / 16) ulp.  Model: 6e-8 * sqrt(k / 16) / 2 = 6e-8 at
        # k = 64, 3.4e-7 at k = 2048 for the typical element; the bound keeps that k-scaling with a factor ~ 60 for the maximum over
        # up to 1.7e7 elements and the order of the final read-modify-write (round 4 had replaced it by a flat 2e-5: 10 x looser at small k)
        tol = 4e-6 * (k / 64.0) ** 0.5
        assert err < tol, (variant, err, tol)
        if tri:
            assert torch.equal(c[:, :m].t()[~mask], c0[:, :m].t()[~mask]), "entries below the diagonal must not be touched"
        assert torch.equal(c[:, m:], c0[:, m:]), "padding of the leading dimension must not be touched"
        outs.append(got)
    if len(outs) == 2:
        assert float((outs[0] - outs[1])[mask].abs().max()) / scale < 8e-6 * (k / 64.0) ** 0.5


@pytest.mark.parametrize("m,n,k,tri,st", [(256, 256, 32, 0, 8), (256, 256, 64, 1, 8), (512, 768, 96, 0, 1), (1024, 1024, 160, 1, 2), (2304, 2304, 64, 1, 8),
                                          (2048, 2048, 512, 1, 8), (1280, 4352, 320, 0, 3), (4096, 4096, 2048, 1, 8), (3072, 3072, 4096, 1, 4), (768, 2048, 224, 1, 8), (512, 512, 128, 1, 8), (1024, 768, 192, 0, 2)])
def test_bf16_update_third_generation(m, n, k, tri, st):
    """The C-stationary 256 x 256 kernel (csrc/bf16_tn3.hip; LDS rings of 3 and 4 stages) against torch's fp32 matmul of the same bf16
    operands and BIT FOR BIT against the 128 x 128 kernel where both apply (same k order per element, one final add into C): 1 .. 128
    stages (ring fills, wraps and drains; the tail's vmcnt counts), supertile edges 1 .. 8, diagonal tiles, strips of a triangular
    update (m < n), untouched strictly-lower part and leading-dimension padding."""
    g = torch.Generator(device="cuda"); g.manual_seed(3 * m + n + k)
    lda = k + 64
    abuf = torch.randn(max(m, n) if tri else m, lda, device="cuda", generator=g).to(torch.bfloat16)
    if tri:
        a16 = abuf[:m, :k]; b16 = abuf[:n, :k]                    # C = rows [0, m) x columns [0, n) of a symmetric update
    else:
        a16 = abuf[:, :k]
        bbuf = torch.randn(n, lda, device="cuda", generator=g).to(torch.bfloat16); b16 = bbuf[:, :k]
    c0 = torch.randn(n, m + 32, device="cuda", generator=g)
    ref = c0[:, :m].t().double() - 0.5 * (a16.double() @ b16.double().t())
    mask = torch.triu(torch.ones(m, n, dtype=torch.bool, device="cuda")) if tri else torch.ones(m, n, dtype=torch.bool, device="cuda")
    scale = float(ref.abs().max())
    outs = {}
    for variant in (0, 3, 4, 5, 6):
        c = c0.clone()
        rc = _bf16_update(variant, a16, b16, c, -0.5, tri, st if variant else 0)
        if variant in (0, 5, 6) and rc != 0:
            assert k % 64, rc                                  # the 128-tile kernel and the wide staging take K in steps of 64
            continue
        assert rc == 0, (variant, rc)
        torch.cuda.synchronize()
        got = c[:, :m].t()
        err = float((got.double() - ref)[mask].abs().max()) / scale
        assert err < 4e-6 * (k / 64.0) ** 0.5 + 1e-7, (variant, err)
        if tri:
            assert torch.equal(got[~mask], c0[:, :m].t()[~mask]), "entries below the diagonal must not be touched"
        assert torch.equal(c[:, m:], c0[:, m:]), "padding of the leading dimension must not be touched"
        outs[variant] = got.clone()
    assert torch.equal(outs[3][mask], outs[4][mask])
    if 0 in outs:
        assert torch.equal(outs[0][mask], outs[3][mask]), "same k order, one final add: the generations must agree bit for bit"
        assert torch.equal(outs[5][mask], outs[3][mask]) and torch.equal(outs[6][mask], outs[3][mask])


def test_bf16_update_third_generation_refusals():
    a16 = torch.zeros(384, 64, device="cuda", dtype=torch.bfloat16); c = torch.zeros(384, 384, device="cuda")
    assert _bf16_update(3, a16, a16, c, 1.0, 1) != 0             # 384 is no multiple of 256
    a16 = torch.zeros(256, 48, device="cuda", dtype=torch.bfloat16); c = torch.zeros(256, 256, device="cuda")
    assert _bf16_update(3, a16, a16, c, 1.0, 1) != 0             # K = 48 is no multiple of 32


def test_bf16_update_dispatcher_and_refusals():
    """variant 1 refuses shapes outside whole 256 x 128 x 64 tiles (the dispatcher then takes the 128-tile kernel); the factorization
    gives the same factor (to fp32 rounding of a different summation order) with either kernel forced for every big update."""
    from capital_amd import mixed
    from capital_amd.matrix import matrix
    a16 = torch.randn(384, 64, device="cuda").to(torch.bfloat16)
    c = torch.zeros(384, 384, device="cuda")
    assert _bf16_update(1, a16, a16, c, 1.0, 1) != 0             # m = 384 is no multiple of 256
    assert _bf16_update(-1, a16, a16, c, 1.0, 1) == 0
    torch.cuda.synchronize()
    want = torch.triu(a16.float() @ a16.float().t())
    assert float((torch.triu(c.t()) - want).abs().max()) < 1e-4
    n = 6144
    a = _spd(n, "gram", seed=3)
    A = matrix(n, n, 1, 1).from_numpy(a)
    fs = []
    for kern in (0, 1):
        p = mixed.plan(n, 4)
        p.set_option("update_kernel", kern); p.set_option("update_min_tiles", 0)
        p.factor(A)
        assert p.last_info() == 0
        fs.append(p.R32().cpu().numpy().astype(np.float64))
        p.set_option("update_kernel", 0); p.set_option("update_min_tiles", 1024)
        p.close()
    ref = np.linalg.cholesky(a).T
    assert relerr(fs[0], ref) < 2e-2 and relerr(fs[1], ref) < 2e-2
    assert relerr(fs[1], fs[0]) < 1e-3            # same bf16 panels up to rare rounding flips; fp32 sums in a different order
