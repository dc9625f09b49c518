"""-m gpu: CholeskyQR / CholeskyQR2 1D path through the C ABI vs reference dumps and the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import capital_oracle as orc  # noqa: E402
from tests.gpu_util import relerr  # noqa: E402


def _run(m, n, variant, a=None):
    from capital_amd import cacqr, cholinv
    from capital_amd.matrix import matrix
    A = matrix(n, m, 1, 1)
    if a is None:
        A.distribute_random(0, 0, 1, 1, 0)
    else:
        A.from_numpy(a)
    pack = cacqr.info(variant, cholinv.info(1, 1, 0, 'U'))
    cacqr.factor(A, pack, None)
    return A, pack


@pytest.mark.parametrize("name", ["cacqr1_m192_n12.npz", "cacqr2_m256_n16.npz"])
def test_matches_reference_dump(golden_dir, name):
    from capital_amd import cacqr, validate
    g = np.load(os.path.join(golden_dir, name))
    m, n, variant = int(g["m"]), int(g["n"]), int(g["variant"])
    A, pack = _run(m, n, variant)
    assert np.array_equal(A.to_numpy(), g["A"])                       # upstream's distribute_random, bit exact
    Q = cacqr.construct_Q(pack).to_numpy(); R = cacqr.construct_R(pack).to_numpy()
    assert pack.last_info() == 0
    assert relerr(Q, g["Q"]) < 1e-12
    assert relerr(R, np.triu(g["R"])) < 1e-13
    assert validate.qr.residual(A, pack) < 1e-13                     # SURVEY App. A bars
    assert validate.qr.orthogonality(A, pack) < max(1e-15, 10 * float(g["ref_orthogonality"]))


@pytest.mark.parametrize("name", ["cacqr2_p8_c1_m256_n16.npz", "cacqr2_p8_c2_m256_n16.npz", "cacqr1_p8_c2_m200_n12.npz"])
def test_matches_multirank_reference_dump(golden_dir, name):
    """Q and R of the REAL reference run on 8 MPI ranks (1D path and sweep_3d on 2 x 2 x 2; gathered from the ranks' cyclic
    pieces): the GPU plan on the same global matrix gives the same factorization."""
    from capital_amd import cacqr, validate
    g = np.load(os.path.join(golden_dir, name))
    m, n, variant = int(g["m"]), int(g["n"]), int(g["variant"])
    A, pack = _run(m, n, variant, a=np.ascontiguousarray(g["A"]))
    Q = cacqr.construct_Q(pack).to_numpy(); R = cacqr.construct_R(pack).to_numpy()
    assert pack.last_info() == 0
    assert relerr(Q, g["Q"]) < 1e-12
    assert relerr(R, np.triu(g["R"])) < 1e-13
    assert validate.qr.residual(A, pack) < 1e-13
    assert validate.qr.orthogonality(A, pack) < max(1e-15, 10 * float(g["ref_orthogonality"]))


# (n = 256, m % 128 == 0 runs the dedicated gram256 / qrapply256 kernels: 8192 = one row tile per workgroup, 33024 = 258 row tiles (two per
#  workgroup on 129 workgroups), 70272 = 549 row tiles (three per workgroup, the last workgroup short) - the persistent K loop across row tiles)
@pytest.mark.parametrize("m,n,variant", [(4096, 64, 2), (5000, 37, 2), (8192, 256, 1), (8192, 256, 2), (33024, 256, 1), (70272, 256, 2), (100000, 128, 2), (130, 130, 2)])
def test_matches_oracle(m, n, variant):
    from capital_amd import cacqr, validate
    A, pack = _run(m, n, variant)
    a = A.to_numpy()
    assert np.array_equal(a, orc.random_local(m, n, 0, 0, 1, 1, 0))
    q_ref, r_ref = orc.cacqr_1d([a], variant)
    Q = cacqr.construct_Q(pack).to_numpy(); R = cacqr.construct_R(pack).to_numpy()
    assert pack.last_info() == 0
    assert relerr(R, r_ref) < 1e-11
    assert relerr(Q, q_ref[0]) < 1e-10
    res, orth = validate.qr.residual(A, pack), validate.qr.orthogonality(A, pack)
    assert abs(res - orc.qr_residual(a, Q, R)) < 1e-15
    assert abs(orth - orc.qr_orthogonality(Q)) < 1e-15
    assert res < 1e-13
    if variant == 2:
        assert orth < 1e-15
    assert np.array_equal(np.tril(R, -1), np.zeros_like(R))


def test_tall_skinny_properties_at_scale():
    """2^21 x 256 per GPU (the per-rank shape of BASELINE config 4): properties only."""
    from capital_amd import validate
    A, pack = _run(1 << 21, 256, 2)
    assert pack.last_info() == 0
    assert validate.qr.residual(A, pack) < 1e-13
    assert validate.qr.orthogonality(A, pack) < 1e-15
