"""qr::cacqr mirror (reference src/alg/qr/cacqr/cacqr.h:18-55, cacqr.hpp:5-248).

    pack = cacqr.info(num_iter, cholinv.info(...))      # num_iter: 1 = CholeskyQR, 2 = CholeskyQR2
    cacqr.factor(A, pack, topo.rect(c))                 # A: local element-cyclic piece: rows over d, columns over c
    Q = cacqr.construct_Q(pack, topo); R = cacqr.construct_R(pack, topo)

c == 1 (cacqr.hpp:229, the shape of BASELINE config 4): the 1D path - rows cyclic over all ranks, one Gram all-reduce
per sweep.  c > 1: the 3D (c == d) / tunable-grid (d > c) path of cacqr.hpp:75-170 on the c x d x c grid of a
`topo.rect` bundle (csrc/cacqr.hip: sweep_grid)."""
import ctypes as C

import torch

from . import _lib
from ._util import cur_stream
from .matrix import matrix, rect


class info:
    def __init__(self, num_iter, cholesky_inverse_args=None):
        self.num_iter = int(num_iter)
        self.cholesky_inverse_args = cholesky_inverse_args
        self._plan = None
        self._shape = None
        self._grid = False

    def _ensure(self, m_local, n, comm):
        if self._plan is not None and self._shape == (m_local, n) and not self._grid:
            return
        self._release()
        h = C.c_void_p()
        _lib.check(_lib.lib().cap_cacqr_plan_create(C.byref(h), m_local, n, self.num_iter, comm), "cap_cacqr_plan_create")
        self._plan, self._shape, self._grid = h, (m_local, n), False

    def _ensure_grid(self, A, topo):
        key = (A.num_rows_local(), A.num_columns_local())
        if self._plan is not None and self._shape == key and self._grid:
            return
        self._release()
        h = C.c_void_p()
        _lib.check(_lib.lib().cap_cacqr_plan_create_grid(C.byref(h), A.num_rows_global(), A.num_columns_global(), self.num_iter,
                                                         topo.handle), "cap_cacqr_plan_create_grid")
        self._plan, self._shape, self._grid = h, key, True

    def _release(self):
        if self._plan is not None:
            _lib.lib().cap_cacqr_plan_destroy(self._plan)
            self._plan = None

    def last_info(self):
        v = C.c_int64(0)
        _lib.check_info(_lib.lib().cap_cacqr_info(self._plan, cur_stream(), C.byref(v)), "cap_cacqr_info")
        return v.value

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass


def factor(A, args, CommInfo=None):
    """cacqr::factor (cacqr.hpp:217-248) for the c == 1 grid."""
    c = getattr(CommInfo, "c", 1) if CommInfo is not None else 1
    args._gm, args._gn = A.num_rows_global(), A.num_columns_global()
    if c != 1:
        if getattr(CommInfo, "handle", None) is None:
            raise _lib.CapitalError("the c > 1 CholeskyQR grids need a topo.rect bundle (sub-communicators)")
        args._ensure_grid(A, CommInfo)
        _lib.check(_lib.lib().cap_cacqr_factor(args._plan, A.data_ptr(), A.ld(), cur_stream()), "cacqr::factor")
        return
    comm = getattr(CommInfo, "world", None) if CommInfo is not None else None
    args._ensure(A.num_rows_local(), A.num_columns_local(), comm)
    _lib.check(_lib.lib().cap_cacqr_factor(args._plan, A.data_ptr(), A.ld(), cur_stream()), "cacqr::factor")


def construct_Q(args, CommInfo=None):
    """cacqr.hpp construct_Q: fresh rect matrix with this rank's row-cyclic piece of Q."""
    m_local, n = args._shape
    d = getattr(CommInfo, "d", 1) if CommInfo is not None else 1
    c = getattr(CommInfo, "c", 1) if CommInfo is not None else 1
    out = matrix(args._gn, args._gm, c, d, rect)
    ld = C.c_int64(0)
    q = _lib.lib().cap_cacqr_Q_ptr(args._plan, C.byref(ld))
    st = _lib.lib().cap_copy_window(q, 0, ld.value, 0, 0, out.data_ptr(), 0, out.ld(), 0, 0, m_local, n, 0, 0, cur_stream())
    _lib.check(st, "construct_Q")
    return out


def construct_R(args, CommInfo=None):
    """the caller's piece of R: the whole n x n factor on the 1D grid, the c x c element-cyclic piece (rows y mod c, columns
    x mod c) on the 3D / tunable grids - what upstream's args.R holds (cacqr.hpp:214)."""
    if args._grid:
        c = CommInfo.c
        out = matrix(args._gn, args._gn, c, c, rect)
        _lib.check(_lib.lib().cap_cacqr_R_piece(args._plan, out.data_ptr(), out.ld(), cur_stream()), "construct_R")
        return out
    m_local, n = args._shape
    out = matrix(n, n, 1, 1, rect)
    ld = C.c_int64(0)
    r = _lib.lib().cap_cacqr_R_ptr(args._plan, C.byref(ld))
    st = _lib.lib().cap_copy_window(r, 0, ld.value, 0, 0, out.data_ptr(), 0, out.ld(), 0, 0, n, n, 1, 1, cur_stream())
    _lib.check(st, "construct_R")
    return out


def dense_R(args):
    """the replicated dense n x n R (device view) - grid and 1D plans alike."""
    n = args._gn
    ld = C.c_int64(0)
    r = _lib.lib().cap_cacqr_R_ptr(args._plan, C.byref(ld))
    out = matrix(n, n, 1, 1, rect)
    st = _lib.lib().cap_copy_window(r, 0, ld.value, 0, 0, out.data_ptr(), 0, out.ld(), 0, 0, n, n, 1, 1, cur_stream())
    _lib.check(st, "dense_R")
    return out
