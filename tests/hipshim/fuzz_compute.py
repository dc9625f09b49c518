"""TEST INFRASTRUCTURE: random configurations through the compute mode of the stand-in (run_compute.py) - sizes, block widths, grids and
schedule options nobody wrote a test for.  python tests/hipshim/fuzz_compute.py SEED COUNT [kind ...]  prints every case that is off."""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import run_compute as rc   # noqa: E402


def cholinv_cfg(rng):
    n = rng.choice([rng.randint(65, 700), rng.randint(700, 1800), rng.choice([256, 512, 1024, 1536, 2048])])
    ci, split, bc = rng.choice([-1, 0, 1]), rng.choice([1, 1, 2]), rng.choice([0, -1, -2, -3, -4])
    nb = rng.choice([64, 128, 128, 256])
    opts = [("nb", nb)]
    for k, vals in [("outer", [nb, 2 * nb, 3 * nb]), ("tail", [0, n // 4, n]), ("depth2", [0, 1]), ("pair_rest", [0, 1]), ("use_sb", [0, 1]), ("lookahead", [0, 1, 1]),
                    ("fastdiag", [0, 1, 1]), ("inner_la", [0, 0, 1]), ("inv_overlap", [0, 1]), ("inv_fast", [0, 1, 1]), ("inv_start_m", [0, n // 2, 1 << 30]),
                    ("chain_coop", [0, 4, 32]), ("serial_m", [0, 0, n // 2]), ("fuse_copy", [0, 1]), ("reserve", [0, 0, 8]), ("leaf", [16, 32, 64, 64]), ("occ1_m", [0, 512, 16384])]:
        if rng.random() < 0.45:
            opts.append((k, rng.choice(vals)))
    return ("cholinv n=%d ci=%d split=%d bc=%d %s" % (n, ci, split, bc, dict(opts)), lambda r, e: rc.cholinv_compute(r, e, n, ci, split, bc, tuple(opts), seed=rng.randint(0, 99)))


def dist_cfg(rng):
    nb = rng.choice([128, 128, 256]); P = rng.randint(1, 8)
    n = rng.choice([rng.randint(nb + 1, 1400), nb * rng.randint(2, 10)])
    ci = rng.choice([-1, -1, 0, 1])
    opts = []
    for k, vals in [("strip", [1, 2]), ("depth2", [0, 1]), ("safe", [0, 1]), ("ipc", [0, 1]), ("split", [1, 2]), ("occ1_m", [0, 256, 16384])]:
        if rng.random() < 0.5:
            opts.append((k, rng.choice(vals)))
    return ("dist n=%d nb=%d P=%d ci=%d %s" % (n, nb, P, ci, dict(opts)), lambda r, e: rc.dist_compute(r, e, n, nb, P, tuple(opts), ci, seed=rng.randint(0, 99)))


def dist2d_cfg(rng):
    nb = rng.choice([128, 128, 256]); Pr = rng.choice([1, 2, 2, 4]); Pc = Pr * rng.choice([k for k in (1, 2, 4, 8) if Pr * k <= 8])
    n = rng.choice([rng.randint(nb + 1, 1400), nb * rng.randint(2, 10)])
    opts = []
    for k, vals in [("strip", [1, 2]), ("depth2", [0, 1]), ("safe", [0, 1]), ("ipc", [0, 1]), ("complete_inv", [0, 1]), ("split", [1, 2]), ("occ1_m", [0, 256, 16384])]:
        if rng.random() < 0.5:
            opts.append((k, rng.choice(vals)))
    return ("dist2d n=%d nb=%d %dx%d %s" % (n, nb, Pr, Pc, dict(opts)), lambda r, e: rc.dist2d_compute(r, e, n, nb, Pr, Pc, tuple(opts), seed=rng.randint(0, 99)))


def cyclic_cfg(rng):
    c, d = rng.choice([(1, 2), (2, 2), (2, 1), (4, 1), (2, 2)])
    n = rng.choice([rng.randint(200, 1300), 128 * rng.randint(2, 9)])
    ci, bc = rng.choice([-1, 0, 0, 1]), rng.choice([0, -1, -2, -3, -4])
    return ("cyclic n=%d ci=%d bc=%d grid %dx%dx%d" % (n, ci, bc, d, d, c), lambda r, e: rc.cholinv_cyclic_compute(r, e, n, ci, c, d, bc, seed=rng.randint(0, 99)))


def dmp_cfg(rng):
    nb = rng.choice([128, 256]); P = rng.randint(1, 8)
    n = 128 * rng.randint(max(2, nb // 128), 12)
    nrhs = rng.choice([1, 5, 8, 130])
    return ("dmp n=%d nb=%d P=%d nrhs=%d" % (n, nb, P, nrhs), lambda r, e: rc.dmp_compute(r, e, n, nb, P, nrhs, seed=rng.randint(0, 99)))


def cacqr_cfg(rng):
    P = rng.randint(1, 8); n = rng.choice([16, 64, 96, 128, 200, 256]); iters = rng.choice([1, 2, 2])
    ml = rng.choice([rng.randint(n, 900), 128 * rng.randint(2, 8), 512])
    ml = max(ml, n)
    return ("cacqr m=%d n=%d iter=%d P=%d" % (ml * P, n, iters, P), lambda r, e: rc.cacqr_compute(r, e, ml * P, n, iters, P, seed=rng.randint(0, 99)))


def summa_cfg(rng):
    c, d = rng.choice([(1, 1), (1, 2), (2, 2), (1, 3), (3, 3), (2, 4), (1, 4)])
    if d % c:
        c = 1
    size = c * d * d
    M, N, K = (rng.randint(d, 400) for _ in range(3))
    chunks = rng.choice([0, 0, 2, 3, 5])
    return ("summa gemm size=%d c=%d %dx%dx%d chunks=%d" % (size, c, M, N, K, chunks), lambda r, e: rc.summa_compute(r, e, size, c, M, N, K, chunks, seed=rng.randint(0, 99)))


def summa_tri_cfg(rng):
    c, d = rng.choice([(1, 1), (1, 2), (2, 2), (1, 3)])
    size = c * d * d
    M, N, K = (rng.randint(2 * d, 300) for _ in range(3))
    chunks = rng.choice([0, 2, 3])
    return ("summa tri size=%d c=%d m=%d n=%d k=%d chunks=%d" % (size, c, M, N, K, chunks), lambda r, e: rc.summa_tri_compute(r, e, size, c, M, N, K, chunks, seed=rng.randint(0, 99)))


KINDS = {"cholinv": (cholinv_cfg, False), "dist": (dist_cfg, True), "dist2d": (dist2d_cfg, True), "cyclic": (cyclic_cfg, True), "dmp": (dmp_cfg, True),
         "cacqr": (cacqr_cfg, True), "summa": (summa_cfg, True), "summa_tri": (summa_tri_cfg, True)}

if __name__ == "__main__":
    seed, count = int(sys.argv[1]), int(sys.argv[2])
    kinds = sys.argv[3:] or list(KINDS)
    rng = random.Random(seed)
    rc.shim.shim_set_compute(1)
    for i in range(count):
        kind = rng.choice(kinds)
        gen, mp = KINDS[kind]
        name, fn = gen(rng)
        (rc.mp_case(name) if mp else rc.case(name, rng.choice([0, 1])))(fn)
        x = rc.RESULTS[-1]
        flag = "BAD " if x["findings"] else "ok  "
        print(flag, name, {k: "%.1e" % v for k, v in x["errors"].items()}, [f[:160] for f in x["findings"][:2]], flush=True)
    bad = [x for x in rc.RESULTS if x["findings"]]
    print("%d cases, %d with findings" % (len(rc.RESULTS), len(bad)))
