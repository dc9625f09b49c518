"""CPU oracle: NumPy restatement of the reference's Cholesky / CholeskyQR2 hot path.

TEST INFRASTRUCTURE - NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import this module.  The product path
(capital_amd/) never imports it and fails loudly when its HIP extension is
missing.

Parity status: PINNED.  Every function below is checked in
tests/test_oracle.py against (a) outputs of the real reference built and run in
the build container (oracle/ref/build_ref.py -> oracle/_ref/*; dumps committed
as tests/golden/*.npz together with tests/golden/make_golden.py) and (b) the
reference's own validators' values recorded in those fixtures.  The arithmetic
itself lives upstream in Intel MKL (un-vendored, un-pinned: config.mk:11); at
that boundary parity is property-pinned (residuals), not bit-pinned - see
DESIGN.md "Oracle".

All file:line citations are relative to /root/reference.
Axes convention (matrix.h:19,47): X = column index, Y = row index; local
storage is column-major; distribution is element-cyclic: global (row gy, col gx)
lives on grid (x = gx mod d, y = gy mod d) at local (gy div d, gx div d).
"""
import math

import numpy as np

_MASK48 = (1 << 48) - 1
_A = 0x5DEECE66D
_C = 0xB


# --------------------------------------------------------------------------- #
# generators (src/matrix/structure.hpp:68-129)
# --------------------------------------------------------------------------- #
def _mul48(a, x):
    """(a * x) mod 2**48 for uint64 arrays without overflowing 64 bits."""
    a = np.asarray(a, dtype=np.uint64)
    x = np.asarray(x, dtype=np.uint64)
    m24 = np.uint64((1 << 24) - 1)
    s24 = np.uint64(24)
    a_lo, a_hi = a & m24, (a >> s24) & m24
    x_lo, x_hi = x & m24, (x >> s24) & m24
    lo = a_lo * x_lo                                   # < 2^48
    mid = (a_hi * x_lo + a_lo * x_hi) & m24            # only low 24 bits survive the shift
    return (lo + (mid << s24)) & np.uint64(_MASK48)


def drand48_of_seed(seed):
    """Value of `srand48(seed); drand48()` (glibc), vectorised, bit-exact.

    srand48 keeps the low 32 bits of the seed: X0 = (seed32 << 16) | 0x330E;
    drand48 advances X1 = (a*X0 + c) mod 2^48 and returns X1 / 2^48
    (structure.hpp:80-88 calls the pair once per matrix element).
    """
    seed = np.asarray(seed, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    x0 = (seed << np.uint64(16)) | np.uint64(0x330E)
    x1 = (_mul48(np.uint64(_A), x0) + np.uint64(_C)) & np.uint64(_MASK48)
    return x1.astype(np.float64) / float(1 << 48)


def symmetric_global(n, diagonally_dominant=True):
    """Global matrix of `distribute_symmetric` (structure.hpp:68-103).

    A[gy, gx] = u(max(gx,gy) + N*min(gx,gy)) (+ N on the diagonal).  The
    generator is grid-independent; the `key` argument upstream is overwritten by
    the per-element srand48 and is irrelevant.
    """
    gy, gx = np.meshgrid(np.arange(n, dtype=np.uint64), np.arange(n, dtype=np.uint64), indexing="ij")
    hi = np.maximum(gx, gy)
    lo = np.minimum(gx, gy)
    a = drand48_of_seed(hi + np.uint64(n) * lo)
    if diagonally_dominant:
        a[np.arange(n), np.arange(n)] += float(n)
    return a


def local_dim(n_global, d):
    """matrix.hpp:8-11 - local dimension is ceil(N/d) (<= 1 padded row/col, zero filled)."""
    return n_global // d + (1 if n_global % d else 0)


def cyclic_local(a_global, x, y, dx, dy):
    """Element-cyclic local piece (rows y::dy, cols x::dx), zero padded to ceil sizes."""
    m, n = a_global.shape
    ml, nl = local_dim(m, dy), local_dim(n, dx)
    out = np.zeros((ml, nl), dtype=a_global.dtype)
    piece = a_global[y::dy, x::dx]
    out[: piece.shape[0], : piece.shape[1]] = piece
    return out


def cyclic_assemble(pieces, m, n, dx, dy):
    """Inverse of cyclic_local: pieces[(x,y)] -> global m x n."""
    out = np.zeros((m, n), dtype=np.float64)
    for (x, y), p in pieces.items():
        tgt = out[y::dy, x::dx]
        tgt[...] = p[: tgt.shape[0], : tgt.shape[1]]
    return out


def symmetric_local(n, x, y, d, diagonally_dominant=True):
    """Local buffer produced by matrix::distribute_symmetric on grid position (x, y) of d x d."""
    return cyclic_local(symmetric_global(n, diagonally_dominant), x, y, d, d)


def _lcg_stream(key, count):
    """`srand48(key)` followed by `count` successive drand48() values (bit-exact)."""
    seed = int(key) & 0xFFFFFFFF
    x = ((seed << 16) | 0x330E) & _MASK48
    out = np.empty(count, dtype=np.float64)
    # block-vectorised: precompute (A_j, C_j) with X_{k+j} = A_j X_k + C_j for j = 1..B
    bsz = int(min(max(count, 1), 1 << 14))
    aj = np.empty(bsz, dtype=np.uint64)
    cj = np.empty(bsz, dtype=np.uint64)
    a_acc, c_acc = 1, 0
    for j in range(bsz):
        a_acc = (a_acc * _A) & _MASK48
        c_acc = (c_acc * _A + _C) & _MASK48
        aj[j], cj[j] = a_acc, c_acc
    pos = 0
    while pos < count:
        take = min(bsz, count - pos)
        xs = (_mul48(aj[:take], np.uint64(x)) + cj[:take]) & np.uint64(_MASK48)
        out[pos:pos + take] = xs.astype(np.float64) / float(1 << 48)
        x = int(xs[take - 1])
        pos += take
    return out


def random_local(m, n, x, y, dx, dy, key):
    """Local buffer of matrix::distribute_random (structure.hpp:105-129).

    One srand48(key) per rank, then a sequential stream over the un-padded local
    entries in column-major order - grid dependent by construction
    (bench/qr/cacqr.cpp:34 passes key = rank / c).
    Returns an (m_loc, n_loc) array (row index first).
    """
    ml, nl = local_dim(m, dy), local_dim(n, dx)
    pad_x = nl - 1 if (n % dx != 0 and (nl - 1) * dx + x >= n) else nl
    pad_y = ml - 1 if (m % dy != 0 and (ml - 1) * dy + y >= m) else ml
    vals = _lcg_stream(key, pad_x * pad_y)
    out = np.zeros((ml, nl), dtype=np.float64)
    out[:pad_y, :pad_x] = vals.reshape(pad_x, pad_y).T
    return out


# --------------------------------------------------------------------------- #
# packed-triangular storage (src/matrix/structure.h:13,39,59)
# --------------------------------------------------------------------------- #
def pack_upper(a):
    """uppertri: column x holds rows 0..x at offset x(x+1)/2 + y (structure.h:39)."""
    n = a.shape[0]
    return np.concatenate([a[: x + 1, x] for x in range(n)]) if n else np.zeros(0)


def unpack_upper(p, n):
    out = np.zeros((n, n), dtype=np.float64)
    for x in range(n):
        off = x * (x + 1) // 2
        out[: x + 1, x] = p[off: off + x + 1]
    return out


# --------------------------------------------------------------------------- #
# operator seam (src/blas/interface.hpp:43-97, src/lapack/interface.hpp:30-58)
# Column-major semantics are expressed on 2-D numpy arrays (row, col).
# --------------------------------------------------------------------------- #
def gemm(a, b, c, trans_a, trans_b, alpha, beta):
    """C = alpha*op(A)*op(B) + beta*C  (blas/interface.hpp:43-59 -> cblas_dgemm)."""
    opa = a.T if trans_a else a
    opb = b.T if trans_b else b
    return alpha * (opa @ opb) + (beta * c if beta != 0 else 0.0)


def trmm(t, b, side_left, upper, trans, alpha, unit=False):
    """B = alpha*op(T)*B (Left) or alpha*B*op(T) (Right); T triangular (interface.hpp:61-79)."""
    tt = np.triu(t) if upper else np.tril(t)
    if unit:
        tt = tt.copy()
        np.fill_diagonal(tt, 1.0)
    op = tt.T if trans else tt
    return alpha * (op @ b) if side_left else alpha * (b @ op)


def trsm(t, b, side_left, upper, trans, alpha):
    """Solve op(T) X = alpha B (Left) or X op(T) = alpha B (Right). Not called upstream
    (upstream inverts then multiplies) - the north_star's real DTRSM."""
    tt = np.triu(t) if upper else np.tril(t)
    op = tt.T if trans else tt
    if side_left:
        return np.linalg.solve(op, alpha * b)
    return np.linalg.solve(op.T, alpha * b.T).T


def syrk(a, c, upper, trans, alpha, beta):
    """C(tri) = alpha*op(A)op(A)^T + beta*C; only the `uplo` triangle is referenced/written
    (interface.hpp:81-97).  trans=True: A^T A (the only form upstream uses, cacqr.hpp:15)."""
    g = (a.T @ a) if trans else (a @ a.T)
    full = alpha * g + (beta * c if beta != 0 else 0.0)
    out = np.array(c, dtype=np.float64, copy=True)
    mask = np.triu(np.ones_like(out, dtype=bool)) if upper else np.tril(np.ones_like(out, dtype=bool))
    out[mask] = full[mask]
    return out


def potrf_upper(a):
    """LAPACKE_dpotrf('U') (lapack/interface.hpp:30-43): A = R^T R, returns R in the upper
    triangle, the strictly lower triangle of the input is left untouched. info returned too
    (upstream discards it)."""
    out = np.array(a, dtype=np.float64, copy=True)
    sym = np.triu(out) + np.triu(out, 1).T
    try:
        r = np.linalg.cholesky(sym).T
    except np.linalg.LinAlgError:
        return out, 1
    iu = np.triu_indices_from(out)
    out[iu] = r[iu]
    return out, 0


def trtri_upper(a):
    """LAPACKE_dtrtri('U','N') (lapack/interface.hpp:45-58): in-place inverse of the upper
    triangle; the strictly lower triangle is untouched."""
    out = np.array(a, dtype=np.float64, copy=True)
    n = out.shape[0]
    inv = np.linalg.solve(np.triu(out), np.eye(n)) if n else out
    iu = np.triu_indices_from(out)
    out[iu] = np.triu(inv)[iu]
    return out


# --------------------------------------------------------------------------- #
# cholinv (src/alg/cholesky/cholinv/cholinv.hpp:6-165)
# --------------------------------------------------------------------------- #
def cholinv_bc_dimension(n_local, c, d, bc_mult_dim):
    """Base-case global dimension, cholinv.hpp:15-18."""
    bc = c * d
    if bc_mult_dim < 0:
        bc *= 2 ** (-bc_mult_dim)
    else:
        for _ in range(bc_mult_dim):
            bc //= 2
    bc = max(1, bc)
    bc = min(n_local, bc)
    bc = n_local // bc
    return d * bc


def cholinv(a_global, complete_inv=1, split=1, bc_mult_dim=0, c=1, d=1):
    """Mathematical restatement of cholesky::cholinv::factor on the GLOBAL matrix.

    Follows the recursion of cholinv.hpp:85-165 (`invoke`): `n` is the current LOCAL
    size; because the distribution is element-cyclic the global block handled by a
    node of local size n1 is the leading n1*d rows/cols of the current window.
      leaf  (cholinv.hpp:93):  n*d <= bcDimension  or  (n >> split) < split
            -> R = chol_upper(A), Rinv = R^-1        (policy.h: potrf; memcpy; trtri)
      node: n1 = n >> split (cholinv.hpp:107); (R11,Ri11) = cholinv(A11);
            R12 = Ri11^T A12 (TRMM Left/Upper/Trans, :118-121);
            A22 -= R12^T R12 (SYRK Upper/Trans alpha=-1 beta=1, :128-137);
            (R22,Ri22) = cholinv(A22);
            unless (root and complete_inv == 0) (:147): Ri12 = -Ri11 R12 Ri22 (:150-154).
    Only the upper triangle of A is consumed (cholinv.hpp:13).  Non-divisible N: the
    local dimension is ceil(N/d) with zero padding (matrix.hpp:8-11); the padded
    rows/cols are dropped here (the `span` trick of policy.h:196 does the same).
    Returns (R, Rinv) as dense upper-triangular N x N arrays.
    """
    assert split > 0
    n_glob = a_global.shape[0]
    n_loc = local_dim(n_glob, d)
    bc_dim = cholinv_bc_dimension(n_loc, c, d, bc_mult_dim)
    a = np.triu(a_global) + np.triu(a_global, 1).T
    a = np.array(a, dtype=np.float64)
    r = np.zeros_like(a)
    ri = np.zeros_like(a)

    def rec(lo, n_local, is_root):
        # window = global rows/cols [lo, hi) where hi = min(lo + n_local*d, N)
        hi = min(lo + n_local * d, n_glob)
        s1 = n_local >> split
        if n_local * d <= bc_dim or s1 < split:
            blk = a[lo:hi, lo:hi]
            rr = np.linalg.cholesky(blk).T
            r[lo:hi, lo:hi] = rr
            ri[lo:hi, lo:hi] = np.triu(np.linalg.solve(rr, np.eye(hi - lo)))
            return
        s2 = n_local - s1
        mid = min(lo + s1 * d, n_glob)
        rec(lo, s1, False)
        if mid < hi:
            r12 = ri[lo:mid, lo:mid].T @ a[lo:mid, mid:hi]
            r[lo:mid, mid:hi] = r12
            a[mid:hi, mid:hi] -= r12.T @ r12
            rec(mid, s2, False)
            if not (is_root and not complete_inv):
                ri[lo:mid, mid:hi] = -(ri[lo:mid, lo:mid] @ r12) @ ri[mid:hi, mid:hi]

    rec(0, n_loc, True)
    return r, ri


# --------------------------------------------------------------------------- #
# validators (test/cholesky/validate.hpp:7-49, test/qr/validate.hpp:7-52,
#             src/util/util.hpp:25-53)
# --------------------------------------------------------------------------- #
def cholesky_residual(a_global, r_global):
    """sqrt(sum_{upper}(R^T R - A)^2) / sqrt(sum_{upper} A^2)  (validate.hpp:33-46)."""
    e = np.triu(np.triu(r_global).T @ np.triu(r_global) - a_global)
    return float(np.sqrt(np.sum(e * e)) / np.sqrt(np.sum(np.triu(a_global) ** 2)))


def qr_residual(a, q, r):
    """||QR - A||_F / ||A||_F (test/qr/validate.hpp:37-52)."""
    e = q @ np.triu(r) - a
    return float(np.linalg.norm(e) / np.linalg.norm(a))


def qr_orthogonality(q):
    """||Q^T Q - I||_F / sqrt(n*n)  - upstream normalises by control=1 per entry
    (test/qr/validate.hpp:24-31)."""
    n = q.shape[1]
    e = q.T @ q - np.eye(n)
    return float(np.linalg.norm(e) / math.sqrt(n * n))


# --------------------------------------------------------------------------- #
# CholeskyQR / CholeskyQR2, 1D path (src/alg/qr/cacqr/cacqr.hpp:5-29,172-193)
# --------------------------------------------------------------------------- #
def cacqr_1d(a_pieces, num_iter=2):
    """1D CholeskyQR(2): a_pieces = list of row-cyclic local blocks (one per rank).

    sweep_1d (cacqr.hpp:5-29): G_loc = A_loc^T A_loc (syrk upper) -> Allreduce over
    world (policy.h:22) -> potrf -> copy -> trtri -> Q_loc = A_loc R^-1 (trmm Right).
    num_iter == 2 (cacqr.hpp:180-188): second sweep on Q, then R = R2 * R1 (trmm Right).
    Returns (q_pieces, R).
    """
    def sweep(pieces):
        g = sum(np.triu(p.T @ p) for p in pieces)
        g = np.triu(g) + np.triu(g, 1).T
        rr = np.linalg.cholesky(g).T
        rinv = np.triu(np.linalg.solve(rr, np.eye(rr.shape[0])))
        return [p @ rinv for p in pieces], rr

    q, r1 = sweep(a_pieces)
    if num_iter > 1:
        q, r2 = sweep(q)
        return q, r2 @ r1
    return q, r1


def row_cyclic_pieces(a_global, d):
    """1D c=1 grid of cacqr: rows cyclic over d ranks (y = rank), columns replicated."""
    return [cyclic_local(a_global, 0, y, 1, d) for y in range(d)]


def row_cyclic_assemble(pieces, m):
    d = len(pieces)
    n = pieces[0].shape[1]
    out = np.zeros((m, n))
    for y, p in enumerate(pieces):
        tgt = out[y::d]
        tgt[...] = p[: tgt.shape[0]]
    return out
