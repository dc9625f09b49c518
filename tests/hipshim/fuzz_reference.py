"""TEST INFRASTRUCTURE: DIFFERENTIAL test against the REAL reference (oracle/_ref/cholinv_ref, built from /root/reference by
oracle/ref/build_ref.py; only where that binary and an MPI launcher exist).  Random (n, complete_inv, split, bc_mult_dim): the reference
runs on 1 rank or on its 2 x 2 x 2 grid (8 MPI ranks) and dumps R / R^-1 (the ranks' element-cyclic pieces), the library runs the same
configuration through the stand-in's compute mode (single-GPU plan; 8 rank threads behind option cyclic_c) - same factors, same pattern
of R^-1, piece by piece.  python tests/hipshim/fuzz_reference.py SEED COUNT"""
import os
import random
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "cholinv_ref")
CACQR = os.path.join(ROOT, "oracle", "_ref", "cacqr_ref")
SUMMA = os.path.join(ROOT, "oracle", "_ref", "summa_ref")
MPIEXEC = "/opt/conda/bin/mpiexec"


def available():
    return os.path.exists(REF) and os.path.exists(MPIEXEC)


def reference(n, ci, split, bc, ranks, pol=1):
    """-> dict like the tests/golden dumps (A, R, Rinv [, pieces, rank_coords, c, d])"""
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1")
    with tempfile.TemporaryDirectory() as td:
        dump = os.path.join(td, "d.bin")
        subprocess.check_output([MPIEXEC, "-n", str(ranks), REF, str(n), str(ci), str(split), str(bc), "0", "0", str(pol if ranks > 1 else 0), dump, "1"],
                                env=env, stderr=subprocess.STDOUT, timeout=300)
        if ranks == 1:
            raw = np.fromfile(dump, dtype=np.float64).reshape(3, n, n)
            a, r, ri = (raw[k].T.copy() for k in range(3))
            return {"A": a, "R": r, "Rinv": ri, "n": n, "complete_inv": ci, "split": split, "bc_mult_dim": bc}
        per_rank, coords = [], []
        a = np.zeros((n, n))
        for q in range(ranks):
            raw = open("%s.%d" % (dump, q), "rb").read()
            rank, x, y, z, d, c, rl, cl = (int(v) for v in np.frombuffer(raw[:64], dtype=np.int64))
            body = np.frombuffer(raw[64:], dtype=np.float64).reshape(3, cl, rl)
            pieces = [body[k].T for k in range(3)]
            per_rank.append(np.stack(pieces)); coords.append((rank, x, y, z))
            if z == 0:
                a[y::d, x::d] = pieces[0][:len(range(y, n, d)), :len(range(x, n, d))]
        return {"A": a, "n": n, "complete_inv": ci, "split": split, "bc_mult_dim": bc, "c": c, "d": d, "pieces": np.stack(per_rank),
                "rank_coords": np.array(coords, dtype=np.int64)}


def reference_cacqr(variant, m, n, c, ranks, ci=1, split=1, bc=0):
    """the real CholeskyQR (variant 1) / CholeskyQR2 (2) on a 1 x ranks x 1 grid (c = 1) or the c x d x c grid; the z = 0 layer's pieces reassembled"""
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1")
    with tempfile.TemporaryDirectory() as td:
        dump = os.path.join(td, "q.bin")
        out = subprocess.check_output([MPIEXEC, "-n", str(ranks), CACQR, str(variant), str(m), str(n), str(c), str(ci), str(split), str(bc), dump, "1"], env=env,
                                      stderr=subprocess.STDOUT, timeout=300).decode(errors="replace")
        if "MKL ERROR" in out:
            return None            # upstream handed BLAS an illegal argument (a 0-column split piece reaches cblas_dtrmm with ldb = 0): its result is not one
        import re
        mo = re.search(r"orthogonality=(\S+)", out)
        if not mo or not (float(mo.group(1)) < (1e-13 if variant == 2 else 1e-8)):
            return None            # an input at or beyond CholeskyQR's reach (upstream's generator repeats its stream per rank key: small square
                                   # matrices come out singular or nearly so - kappa^2 eps ~ 1): whether the Gram matrix's last pivot comes out
                                   # +1e-17 or -1e-17 is rounding's choice; upstream drops LAPACKE's info, the library reports it (info > 0)
        if ranks == 1:
            raw = np.fromfile(dump, dtype=np.float64)
            return {"A": raw[:m * n].reshape(n, m).T.copy(), "Q": raw[m * n:2 * m * n].reshape(n, m).T.copy(), "R": raw[2 * m * n:].reshape(n, n).T.copy(),
                    "m": m, "n": n, "variant": variant, "c": 1, "d": 1}
        a = np.zeros((m, n)); q = np.zeros((m, n)); r = np.zeros((n, n))
        for rk in range(ranks):
            raw = open("%s.%d" % (dump, rk), "rb").read()
            rank, x, y, z, d, cc, rl, cl, rr, rc = (int(v) for v in np.frombuffer(raw[:80], dtype=np.int64))
            b = np.frombuffer(raw[80:], dtype=np.float64)
            al = b[:rl * cl].reshape(cl, rl).T; ql = b[rl * cl:2 * rl * cl].reshape(cl, rl).T
            rloc = b[2 * rl * cl:].reshape(rc, rr).T
            if z != 0:
                continue
            nr, nc = len(range(y, m, d)), len(range(x, n, cc))
            a[y::d, x::cc] = al[:nr, :nc]; q[y::d, x::cc] = ql[:nr, :nc]
            if cc == 1:
                if rank == 0:
                    r[:, :] = rloc
            else:
                r[(y % cc)::cc, x::cc] = rloc
    return {"A": a, "Q": q, "R": r, "m": m, "n": n, "variant": variant, "c": cc, "d": d}


def reference_summa(op, m, n, k, c, chunks, alpha, beta, layout=0):
    """the real matmult::summa::invoke (op: see oracle/ref/drv_summa.cpp) on the c x c x c cube; per rank its coordinates and its arrays
    as (local rows, local columns, packed, the doubles as stored)"""
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1")
    ranks = c * c * c
    with tempfile.TemporaryDirectory() as td:
        dump = os.path.join(td, "s.bin")
        subprocess.check_output([MPIEXEC, "-n", str(ranks), SUMMA, str(op), str(m), str(n), str(k), str(c), str(layout), str(chunks), repr(alpha), repr(beta), dump],
                                env=env, stderr=subprocess.STDOUT, timeout=300)
        out = []
        for q in range(ranks):
            raw = open("%s.%d" % (dump, q), "rb").read()
            h = np.frombuffer(raw[:64], dtype=np.int64); off = 64; arrs = []
            for _ in range(int(h[6])):
                rows, cols, packed = (int(v) for v in np.frombuffer(raw[off:off + 24], dtype=np.int64)); off += 24
                cnt = cols * (cols + 1) // 2 if packed else rows * cols
                arrs.append((rows, cols, packed, np.frombuffer(raw[off:off + 8 * cnt], dtype=np.float64).copy())); off += 8 * cnt
            out.append((tuple(int(v) for v in h[:6]), arrs))
    return {"op": op, "m": m, "n": n, "k": k, "c": c, "chunks": chunks, "alpha": alpha, "beta": beta, "ranks": out}


def main(seed, count):
    sys.path.insert(0, HERE)
    import run_compute as rc
    rc.shim.shim_set_compute(1)
    rng = random.Random(seed)
    for i in range(count):
        u = rng.random()
        if u < 0.2 and os.path.exists(SUMMA):                   # matmult::summa's overloads on the cube (upstream's single-step SUMMA needs c == d)
            c = rng.choice([1, 2, 2, 2, 3])
            op = rng.randint(0, 7)
            m, n, k = (rng.choice([rng.randint(1, 12), rng.randint(12, 150), 32 * rng.randint(1, 6)]) for _ in range(3))
            alpha, beta = rng.choice([1.0, -1.0, 0.75]), (rng.choice([0.0, 1.0, -0.5]) if op in (0, 5, 6, 7) else 0.0)
            chunks = rng.choice([0, 0, 1, 2, 3, 5])
            g = reference_summa(op, m, n, k, c, chunks, alpha, beta)
            rc.mp_case("reference vs library: summa op=%d m=%d n=%d k=%d c=%d chunks=%d alpha=%g beta=%g" % (op, m, n, k, c, chunks, alpha, beta))(
                lambda r, e, g=g: rc.golden_summa(r, e, g))
        elif u < 0.45:                                          # CholeskyQR / CholeskyQR2: 1D grids of 1..8 ranks, the c x d x c grids of 8, 16 and 27
            c, ranks = rng.choice([(1, 1), (1, 2), (1, 3), (1, 4), (1, 5), (1, 8), (2, 8), (2, 8), (2, 16), (3, 27)])
            variant = rng.choice([1, 2, 2])
            n = c * rng.randint(1, 48 // c)
            m = n + rng.choice([0, 1, rng.randint(2, 40), rng.randint(40, 600)])
            ci, split, bc = rng.choice([1, 1, 0]), rng.choice([1, 1, 2]), rng.choice([0, -1, -2])      # the Gram matrix's cholinv inside upstream (cacqr.hpp:86-120, solve :122-170)
            g = reference_cacqr(variant, m, n, c, ranks, ci, split, bc)
            if g is None:
                print("ok   (skipped: upstream passes BLAS an illegal argument or its input is singular at cacqr%d m=%d n=%d c=%d ranks=%d ci=%d split=%d)" % (variant, m, n, c, ranks, ci, split), flush=True)
                continue
            if variant == 1 and c > 1 and ci == 0:
                # upstream's defect: cacqr::solve forms Q1 R12 - A2 (alpha = 1, beta = -1, cacqr.hpp:57-65), so the columns behind
                # the split of ONE sweep come out negated (its own validator prints a residual of 0.7 - 0.9; two sweeps flip twice).  The library
                # returns Q with A = Q R; undo the flip on the reference's side: local columns >= (n / c) >> split, i.e. global columns j with j / c >= that
                cut = (n // c) >> split
                g["Q"] = g["Q"] * np.where(np.arange(n) // c >= cut, -1.0, 1.0)[None, :]
            rc.mp_case("reference vs library: cacqr%d m=%d n=%d c=%d ranks=%d (inner cholinv ci=%d split=%d bc=%d)" % (variant, m, n, c, ranks, ci, split, bc))(lambda r, e, g=g: rc.golden_cacqr(r, e, g))
            if rc.RESULTS[-1]["errors"].get("A - QR", 1.0) < 1e-13:            # M close to N: a random square matrix is as ill-conditioned as it likes, and
                f = rc.RESULTS[-1]["findings"]                                 # both sides' Q = A R^-1 carry kappa(A) eps - the residual is what they share
                f[:] = [x for x in f if not (x.startswith("Q vs") or x.startswith("R vs")) or m > 2 * n]
        else:
            ranks = rng.choice([1, 8])
            ci, split, bc = rng.choice([0, 0, 1]), rng.choice([1, 1, 2, 3]), rng.choice([2, 1, 0, -1, -2, -3, -4, -6, -8])
            if ranks == 1:
                n = rng.choice([rng.randint(1, 40), rng.randint(40, 400), 64 * rng.randint(1, 8)])
                g = reference(n, ci, split, bc, 1)
                rc.case("reference vs library: 1 rank n=%d ci=%d split=%d bc=%d" % (n, ci, split, bc), rng.choice([0, 1]))(lambda r, e, g=g: rc.golden_cholinv_1rank(r, e, g))
            else:
                n = rng.choice([rng.randint(1, 16), rng.randint(8, 200), rng.randint(129, 520), 128 * rng.randint(1, 5)])
                g = reference(n, ci, split, bc, 8)
                rc.mp_case("reference vs library: 2x2x2 n=%d ci=%d split=%d bc=%d" % (n, ci, split, bc))(lambda r, e, g=g: rc.golden_cholinv_8ranks(r, e, g))
        x = rc.RESULTS[-1]
        print("BAD " if x["findings"] else "ok  ", x["name"], {k[:14]: "%.1e" % v for k, v in x["errors"].items()}, [f[:200] for f in x["findings"][:2]], flush=True)
    bad = [x for x in rc.RESULTS if x["findings"]]
    print("%d configurations, %d with findings" % (count, len(bad)))
    return rc.RESULTS


if __name__ == "__main__":
    if not available():
        print("oracle/_ref/cholinv_ref or %s missing" % MPIEXEC); sys.exit(0)
    sys.exit(1 if any(x["findings"] for x in main(int(sys.argv[1]), int(sys.argv[2]))) else 0)
