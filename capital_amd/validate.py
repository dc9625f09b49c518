"""Residual functors computed on the GPU (reference test/cholesky/validate.hpp:7-49,
test/qr/validate.hpp:7-52, src/util/util.hpp:25-53) - the parity metrics of the hot path."""
import math

import torch

from . import _lib
from ._util import cur_stream
from . import cholinv as _cholinv
from . import cacqr as _cacqr


class cholesky:
    @staticmethod
    def residual(A, args, CommInfo=None):
        """sqrt(sum_upper (R^T R - A)^2) / sqrt(sum_upper A^2)  (validate.hpp:33-46)."""
        n = A.num_rows_global()
        R = _cholinv.construct_R(args, CommInfo)
        work = torch.empty(n * n, dtype=torch.float64, device=A.device)
        out = torch.zeros(2, dtype=torch.float64, device=A.device)
        st = _lib.lib().cap_cholesky_residual_terms(A.data_ptr(), A.ld(), R.data_ptr(), R.ld(), n, work.data_ptr(),
                                                    out.data_ptr(), cur_stream())
        _lib.check(st, "cholesky::validate::residual")
        e, c = out.tolist()
        return math.sqrt(e) / math.sqrt(c)


    @staticmethod
    def probe(A_cols, R_cols, gcols=None, nvec=8, allreduce=None, seed=7):
        """Independent check - torch fp64 matmul (rocBLAS), none of this library's kernels:
        ||(R^T R - A) X||_F / ||A X||_F for nvec fixed pseudo-random vectors X.

        A_cols, R_cols: [n, lc] device views of this rank's columns of the symmetric A and of R (zero below the global
        diagonal); gcols: their global column indices (None = all n columns, single rank); allreduce: sums a device
        tensor over the ranks.  Uses (A X)[gcols] = A[:, gcols]^T X (symmetry), so every operand is local."""
        n = A_cols.shape[0]
        g = torch.Generator(device="cpu"); g.manual_seed(seed)
        X = torch.rand(n, nvec, dtype=torch.float64, generator=g).to(A_cols.device) - 0.5
        Xl = X if gcols is None else X[gcols]
        y = R_cols @ Xl                       # y = R X, summed over the ranks' column sets
        if allreduce is not None:
            y = allreduce(y)
        z = R_cols.t() @ y                    # (R^T R X)[gcols]
        w = A_cols.t() @ X                    # (A X)[gcols]
        v = torch.stack([((z - w) ** 2).sum(), (w ** 2).sum()])
        if allreduce is not None:
            v = allreduce(v)
        e, c = v.tolist()
        return math.sqrt(e) / math.sqrt(c)


    @staticmethod
    def rinv_probe(R, Rinv, complete_inv, split=1, nvec=8, seed=11):
        """Independent check of the explicit inverse (cholinv.hpp:144-159) - torch fp64 matmul, none of this library's kernels.

        R, Rinv: [n, n] device views (upper triangular, zero below the diagonal).  Returns (probe, root_nonzeros):
          probe = max over the FILLED diagonal blocks B of ||R_BB (Rinv_BB X) - X||_F / ||X||_F for nvec pseudo-random vectors:
                  complete_inv = 1 -> the whole matrix; complete_inv = 0 -> the two blocks [0, n1) and [n1, n), n1 = n >> split
                  (the inverse of a block-triangular matrix has the blocks' inverses on its diagonal, so each block is checked
                  against R's own diagonal block);
          root_nonzeros = count of non-zero entries of Rinv[0:n1, n1:n]: must be exactly 0 for complete_inv = 0 (the block
                  upstream leaves empty, cholinv.hpp:147) and > 0 for complete_inv = 1."""
        n = R.shape[0]
        n1 = n >> split
        blocks = ((0, n),) if (complete_inv == 1 or n1 <= 0 or n1 >= n) else ((0, n1), (n1, n))
        g = torch.Generator(device="cpu"); g.manual_seed(seed)
        X = (torch.rand(n, nvec, dtype=torch.float64, generator=g) - 0.5).to(R.device)
        worst = 0.0
        for lo, hi in blocks:
            Xb = X[lo:hi]
            Z = R[lo:hi, lo:hi] @ (Rinv[lo:hi, lo:hi] @ Xb)
            worst = max(worst, float((Z - Xb).norm() / Xb.norm()))
        root_nz = 0
        if 0 < n1 < n:
            for c0 in range(n1, n, 4096):                  # column chunks: the comparison's bool temporary stays small
                root_nz += int(torch.count_nonzero(Rinv[0:n1, c0:min(n, c0 + 4096)]))
        return worst, root_nz

    @staticmethod
    def rinv_ok(probe, root_nz, complete_inv, n, split=1, tol=1e-13):
        """the gate bench.py and the tests apply to rinv_probe's result"""
        n1 = n >> split
        if not (probe == probe and probe <= tol):
            return False
        if 0 < n1 < n:
            return root_nz == 0 if complete_inv == 0 else root_nz > 0
        return True


class qr:
    @staticmethod
    def _sumsq(t, ld, m, n, sub_identity=False):
        out = torch.zeros(1, dtype=torch.float64, device=t.device)
        _lib.check(_lib.lib().cap_sumsq(t.data_ptr(), ld, m, n, int(sub_identity), 0, out.data_ptr(), cur_stream()), "sumsq")
        return out

    @staticmethod
    def _grid_terms(A, args, T):
        """c > 1 grids: (Q R)_local by the same layer-wise product the factorization uses, and the Gram blocks of Q.
        Local arithmetic is torch fp64 (validator = test infrastructure); collectives go through the C ABI."""
        L = _lib.lib()
        s = cur_stream()
        Q = _cacqr.construct_Q(args, T); Rd = _cacqr.dense_R(args)
        ml, nl = Q.num_rows_local(), Q.num_columns_local()
        Qz = torch.empty_like(Q.data())
        if T.x == T.z:
            Qz.copy_(Q.data())
        _lib.check(L.cap_comm_bcast(T.row, Qz.data_ptr(), Qz.numel(), T.z, s), "bcast")
        Qzv = Qz[:, :ml].t()
        # (Q R)[rows y, cols x] = sum_z Q[:, cols z] R[z::c, x::c]
        QR = torch.zeros_like(Q.data())
        QR[:, :ml].copy_((Qzv @ Rd.view()[T.z::T.c, T.x::T.c]).t())
        _lib.check(L.cap_comm_allreduce_sum(T.depth, QR.data_ptr(), QR.numel(), s), "allreduce")
        # Gram block G[z::c, x::c] summed over the process rows
        G = (Qzv.t() @ Q.view()).t().contiguous()          # stored column-major nl x nl
        _lib.check(L.cap_comm_allreduce_sum(T.column_contig, G.data_ptr(), G.numel(), s), "allreduce")
        _lib.check(L.cap_comm_allreduce_sum(T.column_alt, G.data_ptr(), G.numel(), s), "allreduce")
        return Q, QR, G

    @staticmethod
    def residual(A, args, CommInfo=None):
        """||QR - A||_F / ||A||_F (test/qr/validate.hpp:37-52); partial sums all-reduced over ranks."""
        if args._grid:
            T = CommInfo
            Q, QR, _ = qr._grid_terms(A, args, T)
            ml = Q.num_rows_local()
            v = torch.stack([((QR[:, :ml] - A.data()[:, :ml]) ** 2).sum(), (A.data()[:, :ml] ** 2).sum()])
            _lib.check(_lib.lib().cap_comm_allreduce_sum(T.slice, v.data_ptr(), 2, cur_stream()), "allreduce")   # one layer: every piece once
            e, c = v.tolist()
            return math.sqrt(e) / math.sqrt(c)
        Q = _cacqr.construct_Q(args, CommInfo); R = _cacqr.construct_R(args, CommInfo)
        m, n = A.num_rows_local(), A.num_columns_local()
        E = torch.empty_like(A.data()); E.copy_(A.data())
        L = _lib.lib()
        _lib.check(L.cap_dgemm(0, 0, m, n, n, 1.0, Q.data_ptr(), Q.ld(), R.data_ptr(), R.ld(), -1.0, E.data_ptr(), A.ld(),
                               cur_stream()), "QR - A")
        num = qr._sumsq(E, A.ld(), m, n); den = qr._sumsq(A.data(), A.ld(), m, n)
        v = torch.cat([num, den])
        comm = getattr(CommInfo, "world", None) if CommInfo is not None else None
        if comm is not None and getattr(CommInfo, "size", 1) > 1:
            _lib.check(L.cap_comm_allreduce_sum(comm, v.data_ptr(), 2, cur_stream()), "allreduce")
        e, c = v.tolist()
        return math.sqrt(e) / math.sqrt(c)

    @staticmethod
    def orthogonality(A, args, CommInfo=None):
        """||Q^T Q - I||_F / sqrt(n*n): upstream normalises by control = 1 per entry (validate.hpp:24-31)."""
        if args._grid:
            T = CommInfo
            Q, _, G = qr._grid_terms(A, args, T)
            nl = Q.num_columns_local()
            Gv = G.t()                                      # [row a, col b] of block (z, x): global (z + c a, x + c b)
            if T.x == T.z:
                Gv = Gv - torch.eye(nl, dtype=torch.float64, device=Gv.device)
            v = (Gv ** 2).sum().reshape(1)
            L = _lib.lib()
            _lib.check(L.cap_comm_allreduce_sum(T.row, v.data_ptr(), 1, cur_stream()), "allreduce")      # blocks (z, x'), every x'
            _lib.check(L.cap_comm_allreduce_sum(T.depth, v.data_ptr(), 1, cur_stream()), "allreduce")    # every z'
            n = Q.num_columns_global()
            return math.sqrt(v.item()) / math.sqrt(n * n)
        Q = _cacqr.construct_Q(args, CommInfo)
        m, n = Q.num_rows_local(), Q.num_columns_local()
        G = torch.zeros(n, n, dtype=torch.float64, device=Q.device)
        L = _lib.lib()
        _lib.check(L.cap_dgemm(1, 0, n, n, m, 1.0, Q.data_ptr(), Q.ld(), Q.data_ptr(), Q.ld(), 0.0, G.data_ptr(), n,
                               cur_stream()), "Q^T Q")
        comm = getattr(CommInfo, "world", None) if CommInfo is not None else None
        if comm is not None and getattr(CommInfo, "size", 1) > 1:
            _lib.check(L.cap_comm_allreduce_sum(comm, G.data_ptr(), n * n, cur_stream()), "allreduce")
        e = qr._sumsq(G, n, n, n, sub_identity=True).item()
        return math.sqrt(e) / math.sqrt(n * n)
