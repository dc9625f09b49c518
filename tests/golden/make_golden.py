#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref, built by
oracle/ref/build_ref.py from /root/reference).  Run in the build container only:

    python oracle/ref/build_ref.py && python tests/golden/make_golden.py

Each fixture stores the reference's own input (its generator), outputs
(construct_R / construct_Rinv or construct_Q / construct_R dumps, column-major)
and the value its own validator printed.  The GPU box has no /root/reference;
tests read only these committed files.
"""
import os
import re
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFDIR = os.path.join(REPO, "oracle", "_ref")
MPIEXEC = "/opt/conda/bin/mpiexec"
ENV = dict(os.environ, MKL_NUM_THREADS="1")


def _kv(line):
    return {k: v for k, v in re.findall(r"(\w+)=([-+.\dEe]+)", line)}


def cholinv_case(name, n, ci, split, bc, pol=0):
    with tempfile.TemporaryDirectory() as td:
        dump = os.path.join(td, "d.bin")
        out = subprocess.check_output([MPIEXEC, "-n", "1", os.path.join(REFDIR, "cholinv_ref"), str(n), str(ci),
                                       str(split), str(bc), "0", "0", str(pol), dump, "1"], env=ENV).decode()
        kv = _kv(out)
        raw = np.fromfile(dump, dtype=np.float64).reshape(3, n, n)
        # dumps are column-major n x n: raw[k] read row-major is the transpose
        a, r, ri = (raw[k].T.copy() for k in range(3))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), A=a, R=r, Rinv=ri, n=n, complete_inv=ci, split=split,
                        bc_mult_dim=bc, policy=pol, ref_residual=float(kv["residual"]), ref_stdout=out.strip())
    print(name, out.strip())


def cholinv_multirank_dump_case(name, n, ci, split, bc, pol=1, ranks=8):
    """8-rank (2 x 2 x 2 grid) run of the real reference; every rank dumps its element-cyclic pieces (oracle/ref/drv_cholinv.cpp),
    the layer z = 0 is reassembled into the global A, R, Rinv (the other layer is checked to hold the same pieces)."""
    with tempfile.TemporaryDirectory() as td:
        dump = os.path.join(td, "d.bin")
        out = subprocess.check_output([MPIEXEC, "-n", str(ranks), os.path.join(REFDIR, "cholinv_ref"), str(n), str(ci),
                                       str(split), str(bc), "0", "0", str(pol), dump, "1"], env=ENV).decode()
        kv = _kv(out)
        mats = [np.zeros((n, n)) for _ in range(3)]
        layers = {}
        per_rank, coords = [], []
        for r in range(ranks):
            raw = open("%s.%d" % (dump, r), "rb").read()
            rank, x, y, z, d, c, rl, cl = (int(v) for v in np.frombuffer(raw[:64], dtype=np.int64))
            body = np.frombuffer(raw[64:], dtype=np.float64).reshape(3, cl, rl)      # column-major local pieces
            pieces = [body[k].T for k in range(3)]
            per_rank.append(np.stack(pieces)); coords.append((rank, x, y, z))
            layers.setdefault((x, y), []).append(pieces)
            if z == 0:
                nr, nc = len(range(y, n, d)), len(range(x, n, d))                     # ragged N: the padded tail is dropped
                for k in range(3):
                    mats[k][y::d, x::d] = pieces[k][:nr, :nc]
        for (x, y), ps in layers.items():                                             # replicated layers agree (A, R exactly)
            for other in ps[1:]:
                assert np.array_equal(ps[0][0], other[0])
                assert np.allclose(ps[0][1], other[1], rtol=0, atol=1e-13)
    a, r, ri = mats
    # pieces[rank] = (A, R, Rinv) exactly as construct_R / construct_Rinv left them on that rank (ceil(n/d) x ceil(n/d), [row, col]):
    # the local upper triangle only - on ranks with y > x the local diagonal holds globally sub-diagonal slots
    np.savez_compressed(os.path.join(HERE, name + ".npz"), A=a, R=r, Rinv=ri, n=n, complete_inv=ci, split=split,
                        bc_mult_dim=bc, policy=pol, ranks=ranks, c=c, d=d, ref_residual=float(kv["residual"]),
                        ref_stdout=out.strip(), pieces=np.stack(per_rank), rank_coords=np.array(coords, dtype=np.int64))
    print(name, out.strip())


def cholinv_multirank_case(name, cases):
    """8-rank (2x2x2) runs, residual only (the dumps of the runs above pin R and Rinv themselves)."""
    rows = []
    for (n, ci, split, bc, pol) in cases:
        out = subprocess.check_output([MPIEXEC, "-n", "8", os.path.join(REFDIR, "cholinv_ref"), str(n), str(ci),
                                       str(split), str(bc), "0", "0", str(pol), "-", "1"], env=ENV).decode()
        kv = _kv(out)
        rows.append((n, ci, split, bc, pol, float(kv["residual"])))
        print(name, out.strip())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), rows=np.array(rows, dtype=np.float64),
                        columns="n,complete_inv,split,bc_mult_dim,policy,ref_residual")


def cacqr_case(name, variant, m, n):
    with tempfile.TemporaryDirectory() as td:
        dump = os.path.join(td, "q.bin")
        out = subprocess.check_output([MPIEXEC, "-n", "1", os.path.join(REFDIR, "cacqr_ref"), str(variant), str(m),
                                       str(n), "1", "1", "1", "0", dump, "1"], env=ENV).decode()
        kv = _kv(out)
        raw = np.fromfile(dump, dtype=np.float64)
        a = raw[: m * n].reshape(n, m).T.copy()
        q = raw[m * n: 2 * m * n].reshape(n, m).T.copy()
        r = raw[2 * m * n:].reshape(n, n).T.copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), A=a, Q=q, R=r, m=m, n=n, variant=variant,
                        ref_residual=float(kv["residual"]), ref_orthogonality=float(kv["orthogonality"]),
                        ref_stdout=out.strip())
    print(name, out.strip())


def cacqr_multirank_dump_case(name, variant, m, n, c, ranks=8):
    """The real reference's CholeskyQR on `ranks` MPI ranks: c = 1 -> the 1D path on a 1 x ranks x 1 grid (cacqr.hpp:229),
    c = 2 -> sweep_3d on the 2 x 2 x 2 grid (cacqr.hpp:75-120).  Every rank dumps its element-cyclic pieces of A and Q and
    what construct_R returns (oracle/ref/drv_cacqr.cpp); layer z = 0 is reassembled here."""
    with tempfile.TemporaryDirectory() as td:
        dump = os.path.join(td, "q.bin")
        out = subprocess.check_output([MPIEXEC, "-n", str(ranks), os.path.join(REFDIR, "cacqr_ref"), str(variant), str(m),
                                       str(n), str(c), "1", "1", "0", dump, "1"], env=ENV).decode()
        kv = _kv(out)
        a = np.zeros((m, n)); q = np.zeros((m, n)); r = np.zeros((n, n))
        keys = []
        for rk in range(ranks):
            raw = open("%s.%d" % (dump, rk), "rb").read()
            rank, x, y, z, d, cc, rl, cl, rr, rc = (int(v) for v in np.frombuffer(raw[:80], dtype=np.int64))
            b = np.frombuffer(raw[80:], dtype=np.float64)
            al = b[:rl * cl].reshape(cl, rl).T; ql = b[rl * cl:2 * rl * cl].reshape(cl, rl).T
            rloc = b[2 * rl * cl:].reshape(rc, rr).T
            keys.append((rank, x, y, z, rank // cc))
            if z != 0:
                continue
            nr, nc = len(range(y, m, d)), len(range(x, n, cc))
            a[y::d, x::cc] = al[:nr, :nc]; q[y::d, x::cc] = ql[:nr, :nc]
            if cc == 1:
                if rank == 0:
                    r[:, :] = rloc                               # replicated n x n factor
            else:
                r[(y % cc)::cc, x::cc] = rloc                    # the c x c cyclic piece (rows y mod c, columns x)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), A=a, Q=q, R=r, m=m, n=n, variant=variant, ranks=ranks, c=cc, d=d,
                        rank_coords=np.array(keys, dtype=np.int64), ref_residual=float(kv["residual"]),
                        ref_orthogonality=float(kv["orthogonality"]), ref_stdout=out.strip())
    print(name, out.strip())


def summa_dump_case(name, op, m, n, k, c, chunks, alpha, beta):
    """The real matmult::summa::invoke (oracle/ref/drv_summa.cpp: GEMM / TRMM / SYRK overloads, operands prepared like upstream's call sites)
    on the c x c x c cube: every rank's pieces of the operands and of the result, as stored (packed upper triangles stay packed)."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tests", "hipshim"))
    import fuzz_reference
    g = fuzz_reference.reference_summa(op, m, n, k, c, chunks, alpha, beta)
    flat = {"op": op, "m": m, "n": n, "k": k, "c": c, "chunks": chunks, "alpha": alpha, "beta": beta,
            "coords": np.array([co for co, _ in g["ranks"]], dtype=np.int64)}
    for q, (_, arrs) in enumerate(g["ranks"]):
        for i, (rows, cols, packed, v) in enumerate(arrs):
            flat["shape_%d_%d" % (q, i)] = np.array([rows, cols, packed], dtype=np.int64); flat["data_%d_%d" % (q, i)] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **flat)
    print(name, "ranks=%d" % len(g["ranks"]))


if __name__ == "__main__":
    cholinv_case("cholinv_n64_ci0_s1_bc-2", 64, 0, 1, -2)
    cholinv_case("cholinv_n64_ci1_s1_bc-3", 64, 1, 1, -3)
    cholinv_case("cholinv_n96_ci1_s1_bc0", 96, 1, 1, 0)
    cholinv_case("cholinv_n100_ci0_s2_bc-4", 100, 0, 2, -4)
    cholinv_multirank_case("cholinv_p8_residuals", [(256, 0, 1, -2, 1), (250, 1, 1, -2, 1), (512, 0, 1, -3, 2),
                                                    (257, 0, 1, -2, 3)])
    # the real reference on its own 2 x 2 x 2 grid, R / Rinv gathered from the ranks' element-cyclic pieces
    cholinv_multirank_dump_case("cholinv_p8_n128_ci1_s1_bc-2", 128, 1, 1, -2, 1)
    cholinv_multirank_dump_case("cholinv_p8_n192_ci0_s1_bc-3", 192, 0, 1, -3, 1)
    cholinv_multirank_dump_case("cholinv_p8_n250_ci1_s1_bc-2", 250, 1, 1, -2, 2)
    # where upstream's rules depend on its GRID (round 5; "grid8": the single-GPU plan with the same knobs gives another pattern of R^-1, so
    # these are not among the cholinv_p8_* dumps the single-GPU tests glob): bcMult = 0 on 2 x 2 x 2 - the base case is n / 4 (bcDimLocal
    # starts at c d, cholinv.hpp:15-18), the root IS partitioned and its block of R^-1 stays empty; a ragged n - the root partition is taken
    # on the local dimension, (ceil(251 / 2) >> 1) 2 = 126 rows, not 251 >> 1 = 125 (cholinv.hpp:107)
    cholinv_multirank_dump_case("cholinv_grid8_n256_ci0_s1_bc0", 256, 0, 1, 0, 1)
    cholinv_multirank_dump_case("cholinv_grid8_n251_ci0_s1_bc-2", 251, 0, 1, -2, 1)
    cacqr_case("cacqr1_m192_n12", 1, 192, 12)
    cacqr_case("cacqr2_m256_n16", 2, 256, 16)
    cacqr_multirank_dump_case("cacqr2_p8_c1_m256_n16", 2, 256, 16, 1)     # 1D grid, 8 ranks
    cacqr_multirank_dump_case("cacqr2_p8_c2_m256_n16", 2, 256, 16, 2)     # 3D: 2 x 2 x 2
    cacqr_multirank_dump_case("cacqr1_p8_c2_m200_n12", 1, 200, 12, 2)     # 3D, one sweep, M not a multiple of d * anything special
    cacqr_multirank_dump_case("cacqr2_p27_c3_m270_n27", 2, 270, 27, 3, ranks=27)   # the 3 x 3 x 3 cube (round 5)
    cacqr_multirank_dump_case("cacqr2_p16_c2_m303_n28", 2, 303, 28, 2, ranks=16)   # the tunable grid c x d x c = 2 x 4 x 2 (sweep_tune), ragged M
    # matmult::summa's three overloads on the 2 x 2 x 2 cube (round 5), ragged sizes, chunked and unchunked
    summa_dump_case("summa_c2_gemm_m51_n43_k35", 0, 51, 43, 35, 2, 2, 1.5, -0.5)
    summa_dump_case("summa_c2_trmm_left_trans_m41_n30", 2, 41, 30, 0, 2, 0, 1.0, 0.0)          # cholinv.hpp:114-119
    summa_dump_case("summa_c2_trmm_left_m40_n30", 1, 40, 30, 0, 2, 2, 1.0, 0.0)               # cholinv.hpp:149-150
    summa_dump_case("summa_c2_trmm_right_m40_n31", 3, 40, 31, 0, 2, 0, -1.0, 0.0)             # cholinv.hpp:151-153
    summa_dump_case("summa_c2_syrk_trans_n31_k47", 5, 0, 31, 47, 2, 0, -1.0, 1.0)             # cholinv.hpp:128-131
    summa_dump_case("summa_c2_syrk_rect_n33_k40", 7, 0, 33, 40, 2, 2, -1.0, 0.0)
    # ... and on the 3 x 3 x 3 cube (27 ranks)
    summa_dump_case("summa_c3_gemm_m52_n41_k37", 0, 52, 41, 37, 3, 3, -1.0, 1.0)
    summa_dump_case("summa_c3_trmm_left_trans_m43_n29", 2, 43, 29, 0, 3, 0, 1.0, 0.0)
