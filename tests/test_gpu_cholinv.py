"""-m gpu: cholinv (factor / construct_R / construct_Rinv) through the C ABI vs the real reference's
dumps (tests/golden) and the oracle, plus size-independent properties at BASELINE sizes."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import capital_oracle as orc  # noqa: E402
from tests.gpu_util import relerr  # noqa: E402

RES_TOL = 1e-14      # ||A - R^T R||_F / ||A||_F (upper), fp64: SURVEY App. A suggested bar
GOLD = sorted(os.path.basename(p) for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "cholinv_n*.npz")))


def _factor(n, ci, split=1, bc=-2, a=None, opts=None):
    from capital_amd import cholinv
    from capital_amd.matrix import matrix
    A = matrix(n, n, 1, 1)
    if a is None:
        A.distribute_symmetric(0, 0, 1, 1, 0, True)
    else:
        A.from_numpy(a)
    pack = cholinv.info(ci, split, bc, 'U')
    for k, v in (opts or {}).items():
        pack.set_option(k, v)
    cholinv.factor(A, pack, None)
    return A, pack


@pytest.mark.parametrize("name", GOLD)
def test_matches_reference_dump(golden_dir, name):
    """Same input, same knobs as the REAL reference run -> same R and R^-1 (incl. which Rinv block stays empty)."""
    from capital_amd import cholinv
    g = np.load(os.path.join(golden_dir, name))
    n, ci, split, bc = int(g["n"]), int(g["complete_inv"]), int(g["split"]), int(g["bc_mult_dim"])
    A, pack = _factor(n, ci, split, bc)
    assert np.array_equal(A.to_numpy(), g["A"])
    R = cholinv.construct_R(pack).to_numpy(); Ri = cholinv.construct_Rinv(pack).to_numpy()
    assert pack.last_info() == 0
    assert relerr(R, np.triu(g["R"])) < 1e-14
    assert relerr(Ri, np.triu(g["Rinv"])) < 1e-13
    assert np.array_equal(Ri != 0, np.triu(g["Rinv"]) != 0)
    assert np.array_equal(np.tril(R, -1), np.zeros_like(R))
    res = orc.cholesky_residual(g["A"], R)
    assert res < RES_TOL and res < 10 * float(g["ref_residual"])
    # the extension mode (no inverse) gives the same R
    _, pack2 = _factor(n, -1, split, bc)
    assert relerr(cholinv.construct_R(pack2).to_numpy(), np.triu(g["R"])) < 1e-14


GOLD8 = sorted(os.path.basename(p) for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "cholinv_p8_n*.npz")))


@pytest.mark.parametrize("name", GOLD8)
def test_matches_8rank_reference_dump(golden_dir, name):
    """R and R^-1 of the REAL reference run on 8 MPI ranks (2 x 2 x 2 grid; gathered from the ranks' cyclic pieces): the GPU
    plan with the same knobs gives the same factors (the factor does not depend on the process grid; for complete_inv = 0 the
    reference's empty Rinv block is cut at (local n >> split) * d, which equals n >> split for these even sizes)."""
    from capital_amd import cholinv
    g = np.load(os.path.join(golden_dir, name))
    n, ci, split, bc = int(g["n"]), int(g["complete_inv"]), int(g["split"]), int(g["bc_mult_dim"])
    A, pack = _factor(n, ci, split, bc)
    assert np.array_equal(A.to_numpy(), g["A"])
    R = cholinv.construct_R(pack).to_numpy(); Ri = cholinv.construct_Rinv(pack).to_numpy()
    assert pack.last_info() == 0
    assert relerr(R, np.triu(g["R"])) < 1e-14
    assert relerr(Ri, np.triu(g["Rinv"])) < 1e-13
    assert np.array_equal(Ri != 0, np.triu(g["Rinv"]) != 0)
    res = orc.cholesky_residual(g["A"], R)
    assert res < RES_TOL and res < 10 * float(g["ref_residual"])
    _, pack2 = _factor(n, -1, split, bc)
    assert relerr(cholinv.construct_R(pack2).to_numpy(), np.triu(g["R"])) < 1e-14


@pytest.mark.parametrize("n,ci,split,bc", [(512, 1, 1, -2), (1000, 0, 1, -3), (1000, 1, 2, -3), (777, -1, 1, -2),
                                           (2048, 0, 1, -2), (2048, -1, 1, -3), (1, 1, 1, 0), (3, 0, 1, 0), (130, 1, 1, -1)])
def test_matches_oracle(n, ci, split, bc):
    from capital_amd import cholinv, validate
    A, pack = _factor(n, ci, split, bc)
    a = orc.symmetric_global(n, True)
    r_ref, ri_ref = orc.cholinv(a, max(ci, 0), split, bc, 1, 1)
    R = cholinv.construct_R(pack).to_numpy()
    assert pack.last_info() == 0
    assert relerr(R, r_ref) < 1e-13
    res_gpu = validate.cholesky.residual(A, pack)            # GPU-side validator == oracle's metric
    res = orc.cholesky_residual(a, R)
    assert abs(res_gpu - res) < 1e-16 + 1e-3 * res
    assert res < RES_TOL and res < 10 * orc.cholesky_residual(a, r_ref)
    if ci >= 0:
        Ri = cholinv.construct_Rinv(pack).to_numpy()
        assert relerr(Ri, ri_ref) < 1e-12
        assert np.array_equal(Ri != 0, ri_ref != 0)
    else:
        from capital_amd import _lib
        with pytest.raises(_lib.CapitalError):
            cholinv.construct_Rinv(pack)


@pytest.mark.parametrize("n,ci,split,bc,opts", [
    (2048, 0, 2, -2, {}),                       # root partition n >> 2 = 512 on a panel boundary: unbalanced tree, root node skipped
    (2048, 1, 2, -2, {}),
    (1536, 1, 1, -2, {"nb": 256}),              # 6 panels: non-power-of-two tree
    (1792, 0, 1, -2, {"nb": 256}),              # 7 panels, root partition 896 inside a panel: full tree, root block emptied afterwards
    (1100, 1, 1, -2, {"nb": 128}),              # ragged last panel (76 columns)
    (1100, 0, 1, -2, {"nb": 256}),
    (2048, 1, 1, -2, {"nb": 128, "outer": 256, "tail": 512, "depth2": 1}),
    (4096, 0, 1, -3, {"nb": 512}),
    (1536, 1, 1, -2, {"nb": 256, "lookahead": 0}),
    (3072, 1, 1, -3, {"nb": 256, "inv_overlap": 0}),
    (3072, 1, 1, -3, {"nb": 256, "inv_start_m": 0}),            # overlapped mode that never starts early = flush at the end
    (3072, 0, 1, -3, {"nb": 256, "inv_start_m": 1 << 30}),      # tree enqueued from the first panel on
    (3072, 1, 1, -3, {"nb": 256, "use_sb": 0}),                 # without strip buffers: the tree's events come from the panel stream
    (1100, 0, 1, -2, {"nb": 128, "use_sb": 0}),
])
def test_inverse_tree_on_blocked_factorization(n, ci, split, bc, opts):
    """complete_inv = 0 / 1 on matrices of several panels: blocked right-looking sweep + inverse tree (cholinv.hip,
    factor_with_inverse) against the oracle's recursion (cholinv.hpp:85-165) and against the plain recursion on the GPU."""
    from capital_amd import cholinv
    a = orc.symmetric_global(n, True)
    r_ref, ri_ref = orc.cholinv(a, ci, split, bc, 1, 1)
    A, pack = _factor(n, ci, split, bc, opts=opts)
    assert pack.get_option("inv_fast") == 1 and n >= 2 * pack.get_option("nb")
    assert pack.last_info() == 0
    R = cholinv.construct_R(pack).to_numpy(); Ri = cholinv.construct_Rinv(pack).to_numpy()
    assert relerr(R, r_ref) < 1e-13
    assert relerr(Ri, ri_ref) < 1e-12
    assert np.array_equal(Ri != 0, ri_ref != 0)               # same empty root block (cholinv.hpp:147), same triangle
    assert orc.cholesky_residual(a, R) < RES_TOL
    n1 = n >> split
    for (lo, hi) in ((0, n1), (n1, n)) if ci == 0 else ((0, n),):
        blk = Ri[lo:hi, lo:hi] @ R[lo:hi, lo:hi]
        assert np.linalg.norm(blk - np.eye(hi - lo)) / np.sqrt(hi - lo) < 1e-13
    # plan reuse is deterministic bit for bit (the tree's GEMMs are beta = 0 products: no atomics, fixed order)
    cholinv.factor(A, pack, None)
    assert np.array_equal(cholinv.construct_Rinv(pack).to_numpy(), Ri)
    # the plain recursion (inv_fast = 0) on the same input: same factors up to association order
    _, pack0 = _factor(n, ci, split, bc, opts={"inv_fast": 0})
    assert relerr(cholinv.construct_Rinv(pack0).to_numpy(), Ri) < 1e-12
    assert relerr(cholinv.construct_R(pack0).to_numpy(), R) < 1e-13


def test_inverse_tree_overlap_modes_are_bitwise_identical():
    from capital_amd import cholinv
    n = 2560
    outs = []
    for opts in ({"nb": 256, "inv_overlap": 0}, {"nb": 256, "inv_overlap": 1, "inv_start_m": 0}, {"nb": 256, "inv_overlap": 1, "inv_start_m": 1 << 30},
                 {"nb": 256, "inv_overlap": 1, "inv_start_m": 1024}):
        _, pack = _factor(n, 1, 1, -3, opts=opts)
        outs.append((cholinv.construct_R(pack).to_numpy(), cholinv.construct_Rinv(pack).to_numpy()))
    for (r, ri) in outs[1:]:
        assert np.array_equal(r, outs[0][0]) and np.array_equal(ri, outs[0][1])


_KNOB_R = {}


def _knob_oracle_R(n):
    if n not in _KNOB_R:
        _KNOB_R[n] = orc.cholinv(orc.symmetric_global(n, True), 1, 1, -2, 1, 1)[0]
    return _KNOB_R[n]


@pytest.mark.parametrize("opts", [{"lookahead": 0}, {"lookahead": 1, "nb": 128}, {"nb": 256}, {"nb": 512, "leaf": 32}, {"leaf": 16},
                                  {"nb": 128, "outer": 512, "tail": 256}, {"nb": 128, "outer": 256, "depth2": 1},
                                  {"nb": 128, "outer": 256, "tail": 512, "depth2": 1}, {"nb": 256, "reserve": 8}, {"nb": 512, "fastdiag": 1}, {"nb": 256, "fastdiag": 1, "lookahead": 0},
                                  {"nb": 128, "fastdiag": 1, "outer": 256},
                                  # column-split look-ahead (panel stream on the next chains' columns only, the rest on s_rest)
                                  {"nb": 128, "outer": 256, "inner_la": 1}, {"nb": 128, "outer": 512, "tail": 512, "depth2": 1, "inner_la": 1},
                                  {"nb": 256, "outer": 256, "inner_la": 1}, {"nb": 128, "outer": 1024, "inner_la": 1},
                                  # one bulk workgroup per CU below / above the threshold
                                  {"nb": 128, "outer": 256, "occ1_m": 0}, {"nb": 128, "outer": 256, "occ1_m": 1024},
                                  {"nb": 128, "outer": 256, "occ1_m": 4096, "inner_la": 1},
                                  # strip buffers off (block rows solved through the panel scratch and copied back on the panel stream)
                                  {"nb": 128, "outer": 256, "use_sb": 0}, {"nb": 128, "outer": 512, "tail": 512, "depth2": 1, "use_sb": 0},
                                  {"nb": 128, "outer": 128, "use_sb": 1}, {"nb": 256, "outer": 512, "tail": 0, "use_sb": 1},
                                  # A -> R copy: all of it up front / first strip's rows + the step-0 updates reading A
                                  {"nb": 128, "outer": 256, "fuse_copy": 0}, {"nb": 128, "outer": 256, "depth2": 1, "fuse_copy": 1},
                                  {"nb": 256, "outer": 256, "use_sb": 0, "fuse_copy": 1},
                                  # CU masks for the chain-bound tail only
                                  {"nb": 128, "outer": 256, "reserve": 8, "reserve_m": 768}])
def test_schedule_knobs_do_not_change_the_answer(opts):
    """every schedule knob against the ORACLE's factor (the restated recursion of cholinv.hpp:85-165, computed once)"""
    from capital_amd import cholinv
    n = 1536
    a = orc.symmetric_global(n, True)
    _, pack = _factor(n, -1, 1, -2, opts=opts)
    R = cholinv.construct_R(pack).to_numpy()
    assert orc.cholesky_residual(a, R) < RES_TOL
    assert relerr(R, _knob_oracle_R(n)) < 1e-13


@pytest.mark.parametrize("n,ci,opts", [
    (2048, -1, {"nb": 128, "outer": 256, "depth2": 1}),                       # 8 strips: pairs (0,1) (2,3); the last steps run unpaired
    (2304, -1, {"nb": 128, "outer": 256, "depth2": 1}),                       # 9 strips
    (1280, -1, {"nb": 128, "outer": 256, "depth2": 1}),                       # 5 strips: exactly one pair
    (1024, -1, {"nb": 128, "outer": 256, "depth2": 1}),                       # 4 strips: no step may defer (no strip k + 4)
    (3072, -1, {"nb": 128, "outer": 512, "tail": 1024, "depth2": 1}),         # nb-wide strips in the tail: pairing stops there
    (2048, -1, {"nb": 128, "outer": 256, "depth2": 1, "use_sb": 0}),          # operands inside R: the pair is two adjacent row blocks
    (2048, -1, {"nb": 128, "outer": 256, "depth2": 1, "fuse_copy": 0}),
    (2560, 1, {"nb": 128, "outer": 256, "depth2": 1}),                        # reference semantics: the inverse tree rides on the same sweep
    (2176, -1, {"nb": 128, "outer": 256, "depth2": 1}),                       # ragged last strip
])
def test_paired_far_update_against_the_unpaired_schedule(n, ci, opts):
    """Option "pair_rest" (round 5): the region below strip k + 3 takes the updates of the strips k and k + 1 in ONE product with K = 2 NB
    instead of two with K = NB.  Same products, another association order: R (and R^-1) agree with the unpaired schedule to rounding and
    with the oracle; with strip buffers on / off the paired schedule itself must not change by one bit (same K order either way)."""
    from capital_amd import cholinv
    a = orc.symmetric_global(n, True)
    out = {}
    for pr in (0, 1):
        _, pack = _factor(n, ci, 1, -2, opts=dict(opts, pair_rest=pr))
        assert pack.last_info() == 0 and pack.get_option("pair_rest") == pr
        # an even step k defers when the strips k .. k + 4 exist (k, k + 1 of full height): (nstrip - 3) // 2 paired launches per call
        nstrip = -(-n // opts["outer"])
        if pr == 0:
            assert pack.get_option("count_paired") == 0
        elif "tail" not in opts:
            assert pack.get_option("count_paired") == max(0, (nstrip - 3) // 2), (nstrip, pack.get_option("count_paired"))
        else:
            assert pack.get_option("count_paired") >= 1
        out[pr] = [cholinv.construct_R(pack).to_numpy()] + ([cholinv.construct_Rinv(pack).to_numpy()] if ci >= 0 else [])
        assert orc.cholesky_residual(a, out[pr][0]) < RES_TOL
    for x, y in zip(out[0], out[1]):
        assert relerr(x, y) < 1e-14
    if opts.get("use_sb", 1):
        _, p2 = _factor(n, ci, 1, -2, opts=dict(opts, pair_rest=1, use_sb=0))
        assert np.array_equal(cholinv.construct_R(p2).to_numpy(), out[1][0]), "strip buffers on / off: same K order, same bits"


def test_fused_first_step_copy_is_bitwise_identical():
    """fuse_copy: the step-0 updates read their C input from A and write R (load / add / store) instead of updating a copy with
    fire-and-forget atomics - the same single rounding per element, so R must not change by one bit; A stays untouched."""
    from capital_amd import cholinv
    outs = []
    for ci, n in ((-1, 2048), (1, 1536)):
        for f in (0, 1):
            A, pack = _factor(n, ci, 1, -2, opts={"nb": 128, "outer": 256, "depth2": 1, "fuse_copy": f})
            outs.append(cholinv.construct_R(pack).to_numpy())
            assert np.array_equal(A.to_numpy(), orc.symmetric_global(n, True))
        assert np.array_equal(outs[-1], outs[-2])


@pytest.mark.parametrize("n,nb,ci", [(256, 256, 1), (512, 512, 1), (1024, 1024, 1), (3072, 512, -1), (2048, 1024, 0), (4096, 256, 1)])
def test_one_launch_diagonal_block_chain_is_bitwise_identical(n, nb, ci):
    """"chain_coop" = G > 1 (a PER-PLAN option since round 5): the factor phase of every diagonal block runs as ONE launch of G
    resident workgroups that meet at a counter in global memory after every 64-column step (leaf.hip chain64_coop_kernel) instead of
    one launch per step.  Same blocks, same association order -> R and R^-1 must not differ by one bit, for every G (more workgroups
    than trailing blocks, fewer, one worker); the counters are left zero, so plans can follow each other on the same stream's slot."""
    from capital_amd import cholinv
    def run(G):
        _, pack = _factor(n, ci, 1, -2, opts={"nb": nb, "chain_coop": G})
        assert pack.get_option("chain_coop") == G
        out = [cholinv.construct_R(pack).to_numpy()]
        if ci >= 0:
            out.append(cholinv.construct_Rinv(pack).to_numpy())
        assert pack.last_info() == 0
        return out
    ref = run(0)
    assert orc.cholesky_residual(orc.symmetric_global(n, True), ref[0]) < RES_TOL
    for G in (2, 3, 7, 32, 200):
        got = run(G)
        for x, y in zip(ref, got):
            assert np.array_equal(x, y), (G, float(np.abs(x - y).max()))


def test_chain_workgroup_count_is_a_per_plan_option():
    """Round 4 kept "chain_coop" in a process global set through a per-plan call: one plan's set_option changed every other plan's
    schedule.  Now a plan only sees its own value (or the process default), whatever other plans were told."""
    from capital_amd import cholinv
    a = cholinv.info(-1, 1, -2, 'U'); a._ensure(256)
    b = cholinv.info(-1, 1, -2, 'U'); b._ensure(256)
    default = b.get_option("chain_coop")
    assert default >= 0
    a.set_option("chain_coop", 0 if default else 7)
    assert b.get_option("chain_coop") == default and a.get_option("chain_coop") == (0 if default else 7)
    a.set_option("chain_coop", -1)                     # back to the process default
    assert a.get_option("chain_coop") == default


def test_one_launch_chain_reports_pivots_and_survives_a_busy_gpu():
    """not-SPD input: the failing pivot's index comes out of the one-launch chain like out of the stepwise one (any block, any
    workgroup count); then 40 factorizations back to back next to a stream that keeps every CU busy with large products - the resident
    workgroups only ever wait for finite kernels, results stay bit-identical to the stepwise chain's."""
    from capital_amd import cholinv
    n = 1024
    for row in (0, 63, 64, 200, 1023):
        a = orc.symmetric_global(n, True); a[row, row] = -5.0
        for G in (0, 32):
            _, pack = _factor(n, 1, 1, -2, a=a, opts={"nb": 1024, "chain_coop": G})
            assert pack.last_info() == row + 1
    _, p0 = _factor(2048, 1, 1, -2, opts={"nb": 1024, "chain_coop": 0})          # the stepwise chain: the reference bits
    r0 = cholinv.construct_R(p0).to_numpy(); i0 = cholinv.construct_Rinv(p0).to_numpy()
    side = torch.cuda.Stream()
    x = torch.randn(8192, 8192, device="cuda", dtype=torch.float32)
    A, pack = _factor(2048, 1, 1, -2, opts={"nb": 1024, "chain_coop": 32})
    with torch.cuda.stream(side):
        for _ in range(60):
            y = x @ x
    for _ in range(40):
        cholinv.factor(A, pack, None)
    torch.cuda.synchronize()
    assert np.array_equal(cholinv.construct_R(pack).to_numpy(), r0)
    assert np.array_equal(cholinv.construct_Rinv(pack).to_numpy(), i0)
    assert pack.last_info() == 0
    del y


@pytest.mark.parametrize("n,nb,ci,count", [(2048, 1024, 1, 1), (2048, 512, -1, 3), (2048, 256, 1, 2), (4096, 512, 0, 8)])
def test_chain_that_gives_up_is_restored_and_rerun(n, nb, ci, count):
    """A launch of the one-launch chain whose workgroups are never all resident used to poll for 25 s and leave a half-computed
    factor with info = -64.  Now every launch saves its diagonal block before it touches it, a workgroup gives up after ~ 3 s (its
    peers follow at their next poll), and the recovery launch behind it restores the block and re-runs the chain on two workgroups.
    The test hook makes the next `count` launches give up at their first meeting: same bits as an undisturbed run, info = 0, and
    the event is counted."""
    from capital_amd import _lib, cholinv
    L = _lib.lib()
    _, p0 = _factor(n, ci, 1, -2, opts={"nb": nb, "chain_coop": 32})
    ref = [cholinv.construct_R(p0).to_numpy()] + ([cholinv.construct_Rinv(p0).to_numpy()] if ci >= 0 else [])
    before = p0.get_option("chain_fallbacks")
    assert before >= 0
    _lib.check(L.cap_chain_inject_timeouts(count), "cap_chain_inject_timeouts")
    try:
        _, p1 = _factor(n, ci, 1, -2, opts={"nb": nb, "chain_coop": 32})
        assert p1.last_info() == 0
    finally:
        _lib.check(L.cap_chain_inject_timeouts(0), "cap_chain_inject_timeouts")       # (never leave a pending injection behind)
    got = [cholinv.construct_R(p1).to_numpy()] + ([cholinv.construct_Rinv(p1).to_numpy()] if ci >= 0 else [])
    after = p1.get_option("chain_fallbacks")
    assert after == before + count, (before, after, count)
    for x, y in zip(ref, got):
        assert np.array_equal(x, y), float(np.abs(x - y).max())
    # and the slot is clean again: the next factorization runs undisturbed
    _, p2 = _factor(n, ci, 1, -2, opts={"nb": nb, "chain_coop": 32})
    assert p2.last_info() == 0 and p2.get_option("chain_fallbacks") == before + count
    assert np.array_equal(cholinv.construct_R(p2).to_numpy(), ref[0])


def test_harder_spd_input():
    """A = B^T B + eps I (kappa ~ 1e6): residual stays at fp64 level relative to ||A||."""
    from capital_amd import cholinv
    n = 768
    b = np.random.default_rng(3).standard_normal((n, n))
    a = b.T @ b + 1e-3 * np.eye(n)
    for ci in (-1, 1):
        _, pack = _factor(n, ci, 1, -2, a=a)
        R = cholinv.construct_R(pack).to_numpy()
        assert orc.cholesky_residual(a, R) < 1e-13
        if ci == 1:
            Ri = cholinv.construct_Rinv(pack).to_numpy()
            assert np.linalg.norm(Ri @ R - np.eye(n)) / np.sqrt(n) < 1e-8


@pytest.mark.parametrize("kappa", [1e8, 1e10, 1e12])
def test_ill_conditioned_spd_at_n8192_against_the_oracle(kappa):
    """A = Q diag(1 .. 1/kappa) Q^T at N = 8192 (16 panels of 512): every block-row solve is "invert the 512^2 diagonal block,
    then GEMM" like upstream's TRSM-by-inverse (cholinv.hpp:118-121), whose error grows with the conditioning of R_jj.  The
    bar is therefore the ORACLE's own residual on the same input (the NumPy restatement of upstream's recursion, explicit
    inverses included) x 10 - not a fixed 1e-13 - for the blocked sweep (complete_inv = -1) and for R + R^-1 (complete_inv = 1)."""
    from threadpoolctl import threadpool_limits
    from capital_amd import cholinv
    n = 8192
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    Qm, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g))
    d = torch.logspace(0.0, -float(np.log10(kappa)), n, dtype=torch.float64, device="cuda")
    at = (Qm * d) @ Qm.T
    at = 0.5 * (at + at.T)
    a = at.cpu().numpy()
    del Qm
    with threadpool_limits(limits=32):                       # OpenBLAS on all 256 host threads is slower than on 32
        r_ref, ri_ref = orc.cholinv(a, 1, 1, -4, 1, 1)       # bcMult -4: 512-wide base cases, the GPU schedule's panel width
    rt = torch.from_numpy(r_ref).cuda(); rit = torch.from_numpy(ri_ref).cuda()
    up = torch.triu(torch.ones(n, n, dtype=torch.bool, device="cuda"))
    na = float(at[up].norm())

    def resid(R):        # test/cholesky/validate.hpp:33-46 with torch's fp64 matmul (none of this library's kernels)
        return float((torch.triu(R).T @ torch.triu(R) - at)[up].norm()) / na

    def inv_err(R, Ri):
        return float((torch.triu(Ri) @ torch.triu(R) - torch.eye(n, dtype=torch.float64, device="cuda")).norm()) / np.sqrt(n)
    res_ref, inv_ref = resid(rt), inv_err(rt, rit)
    assert res_ref < 1e-10 and np.isfinite(inv_ref)
    for ci in (-1, 1):
        _, pack = _factor(n, ci, 1, -4, a=a)
        assert pack.last_info() == 0
        R = cholinv.construct_R(pack).view()
        res = resid(R)
        assert res <= max(10.0 * res_ref, 5e-16), (ci, kappa, res, res_ref)
        assert float((R - rt).norm() / rt.norm()) < 1e3 * kappa * 2.2e-16      # forward error of a backward-stable factor: O(kappa eps)
        if ci == 1:
            Ri = cholinv.construct_Rinv(pack).view()
            assert inv_err(R, Ri) <= max(10.0 * inv_ref, 1e-13), (kappa, inv_err(R, Ri), inv_ref)
        del pack, R
        torch.cuda.empty_cache()


def test_n65536_properties():
    """The BASELINE metric's own size through the C ABI (the bench checks the same outside pytest): residual of the reference's
    validator, the independent probe, positive diagonal, strictly-lower part zero."""
    from capital_amd import cholinv, validate
    n = 65536
    A, pack = _factor(n, -1, 1, -5)
    assert pack.last_info() == 0
    res = validate.cholesky.residual(A, pack)
    assert res < RES_TOL, res
    R = cholinv.construct_R(pack)
    assert validate.cholesky.probe(A.view(), R.view()) < 1e-13
    d = torch.diagonal(R.view())
    assert bool((d > 0).all())
    assert float(torch.tril(R.view()[-4096:, -4096:], -1).abs().max()) == 0.0
    del A, pack, R
    torch.cuda.empty_cache()


def test_not_spd_is_reported():
    from capital_amd import cholinv
    n = 300
    a = orc.symmetric_global(n, True); a[200, 200] = -5.0
    for ci in (-1, 1):
        _, pack = _factor(n, ci, 1, -2, a=a)
        assert pack.last_info() == 201


def test_plan_is_reusable_and_input_is_read_only():
    """factor many times on one `info` (bench/cholesky/cholinv.cpp:42-53); A is never modified."""
    from capital_amd import cholinv
    n = 640
    A, pack = _factor(n, 0, 1, -2)
    a0 = A.to_numpy()
    r0 = cholinv.construct_R(pack).to_numpy()
    for _ in range(3):
        cholinv.factor(A, pack, None)
    assert np.array_equal(A.to_numpy(), a0)
    assert np.array_equal(cholinv.construct_R(pack).to_numpy(), r0)      # deterministic, bit for bit


@pytest.mark.parametrize("n,ci", [(8192, -1), (8192, 0), (16384, -1), (32768, -1)])
def test_large_sizes_by_properties(n, ci):
    """BASELINE sizes: too big for the oracle in seconds -> size-independent properties on the GPU:
    residual (reference metric), diag(R) > 0, R upper, and sum of logs of diag(R)^2 == logdet(A) via a
    second independent factorization path (right-looking vs recursive share only the kernels)."""
    from capital_amd import cholinv, validate
    A, pack = _factor(n, ci, 1, -3)
    assert pack.last_info() == 0
    res = validate.cholesky.residual(A, pack)
    assert res < RES_TOL, res
    R = cholinv.construct_R(pack)
    d = torch.diagonal(R.view())
    assert bool((d > 0).all())
    assert float(torch.tril(R.view()[:2048, :2048], -1).abs().max()) == 0.0
    if n <= 8192:
        _, pack2 = _factor(n, 1 if ci < 0 else -1, 1, -2)
        d2 = torch.diagonal(cholinv.construct_R(pack2).view())
        assert abs(float(torch.log(d).sum() - torch.log(d2).sum())) < 1e-9 * n


@pytest.mark.parametrize("n,ci", [(8192, -1), (8192, 1), (16384, -1)])
def test_large_result_against_independent_arithmetic(n, ci):
    """Third opinion above the sizes the NumPy oracle reaches: ||(R^T R - A) X|| / ||A X|| for 8 random vectors with torch's
    fp64 matmul (rocBLAS) - none of this library's kernels is involved in the check (validate.cholesky.residual uses the
    library's own GEMM)."""
    from capital_amd import cholinv, validate
    A, pack = _factor(n, ci, 1, -4)
    assert pack.last_info() == 0
    R = cholinv.construct_R(pack)
    assert validate.cholesky.probe(A.view(), R.view()) < 1e-13
    if ci == 1:
        # R^-1 too: ||R (R^-1 X) - X|| / ||X||
        Ri = cholinv.construct_Rinv(pack)
        g = torch.Generator(device="cpu"); g.manual_seed(3)
        X = torch.rand(n, 8, dtype=torch.float64, generator=g).cuda()
        Y = R.view() @ (Ri.view() @ X)
        assert float((Y - X).norm() / X.norm()) < 1e-13


@pytest.mark.parametrize("n,ci,opts", [(32768, 0, {}), (32768, 1, {}),
                                       # the overlapped tree with >= 32 panels: never started early / started with the first panel
                                       (16384, 1, {"inv_start_m": 0}), (16384, 1, {"inv_start_m": 1 << 30}),
                                       (16384, 0, {"inv_start_m": 1 << 30}), (16384, 1, {"inv_overlap": 0})])
def test_rinv_is_validated_at_the_sizes_it_is_timed_on(n, ci, opts):
    """Half of the reference's output is R^-1 (cholinv.hpp:144-159): at the sizes bench.py quotes reference semantics on, check
    it with arithmetic that is not this library's (torch fp64 matmul): ||R (R^-1 X) - X|| / ||X|| per filled diagonal block, and
    the structure of the root block (exactly empty for complete_inv = 0, cholinv.hpp:147; filled for 1).  A missing event edge
    in the overlapped inverse tree that only bites with many panels shows up here."""
    from capital_amd import cholinv, validate
    A, pack = _factor(n, ci, 1, -5, opts=opts)
    assert pack.last_info() == 0
    assert pack.get_option("inv_fast") == 1 and n // pack.get_option("nb") >= 32
    assert validate.cholesky.residual(A, pack) < RES_TOL
    R = cholinv.construct_R(pack); Ri = cholinv.construct_Rinv(pack)
    probe, root_nz = validate.cholesky.rinv_probe(R.view(), Ri.view(), ci, 1)
    assert probe <= 1e-13, probe
    assert validate.cholesky.rinv_ok(probe, root_nz, ci, n, 1), (probe, root_nz)
    if ci == 1:
        assert root_nz >= 0.999 * (n // 2) * (n // 2)              # a dense block: Rinv12 is filled, not just touched
    # strictly-lower part of R^-1 stays zero, its diagonal is 1 / diag(R)
    assert float(torch.tril(Ri.view()[:4096, :4096], -1).abs().max()) == 0.0
    d = torch.diagonal(R.view()) * torch.diagonal(Ri.view())
    assert float((d - 1.0).abs().max()) < 1e-14
    # second factor call on the same plan: bit-identical R^-1 (the tree's products are beta = 0, fixed order)
    cholinv.factor(A, pack, None)
    Ri2 = cholinv.construct_Rinv(pack)
    assert torch.equal(Ri2.view(), Ri.view())
    del A, pack, R, Ri, Ri2
    torch.cuda.empty_cache()


def test_workspace_follows_the_path_taken():
    """The plan's workspace is sized for the path factor() takes (blocked sweep + tree, or the plain recursion) and grows when an
    option moves the plan to the other one - both orders give the oracle's factors."""
    from capital_amd import cholinv
    n = 2048
    a = orc.symmetric_global(n, True)
    r_ref, ri_ref = orc.cholinv(a, 1, 1, -2, 1, 1)
    A, pack = _factor(n, 1, 1, -2)                   # blocked path first ...
    Ri = cholinv.construct_Rinv(pack).to_numpy()
    pack.set_option("inv_fast", 0)                   # ... then the recursion on the same plan (needs (n/2 + 1)^2 scratch)
    cholinv.factor(A, pack, None)
    assert pack.last_info() == 0
    assert relerr(cholinv.construct_Rinv(pack).to_numpy(), ri_ref) < 1e-12
    assert relerr(Ri, ri_ref) < 1e-12
    pack.set_option("inv_fast", 1); pack.set_option("nb", 128)      # back, with narrower panels (a deeper tree)
    cholinv.factor(A, pack, None)
    assert relerr(cholinv.construct_Rinv(pack).to_numpy(), ri_ref) < 1e-12
    assert relerr(cholinv.construct_R(pack).to_numpy(), r_ref) < 1e-13


def test_profile_launch_by_launch_agrees_with_the_sums():
    """cap_cholinv_profile_launches (one entry per trailing-update launch of the last factor call) against cap_cholinv_profile's totals."""
    import ctypes as C
    from capital_amd import _lib, cholinv
    n = 8192
    A, pack = _factor(n, -1, 1, -5, opts={"profile": 1})
    assert pack.last_info() == 0
    L = _lib.lib()
    nl, ms, fl = C.c_int64(0), C.c_double(0), C.c_double(0)
    _lib.check(L.cap_cholinv_profile(pack._plan, C.byref(nl), C.byref(ms), C.byref(fl)), "cap_cholinv_profile")
    cap = 256
    msv, flv, cnt = (C.c_double * cap)(), (C.c_double * cap)(), C.c_int64(0)
    _lib.check(L.cap_cholinv_profile_launches(pack._plan, msv, flv, cap, C.byref(cnt)), "cap_cholinv_profile_launches")
    assert cnt.value == nl.value and 0 < cnt.value <= cap
    assert abs(sum(flv[i] for i in range(cnt.value)) - fl.value) <= 1e-9 * fl.value
    assert abs(sum(msv[i] for i in range(cnt.value)) - ms.value) <= 1e-3 * ms.value + 1e-3
    assert all(msv[i] > 0 and flv[i] > 0 for i in range(cnt.value))
    # a too small capacity still reports the count and fills what fits
    cnt2 = C.c_int64(0)
    _lib.check(L.cap_cholinv_profile_launches(pack._plan, msv, flv, 1, C.byref(cnt2)), "cap_cholinv_profile_launches")
    assert cnt2.value == cnt.value
    assert L.cap_cholinv_profile_launches(pack._plan, None, flv, cap, C.byref(cnt)) != 0       # argument check
