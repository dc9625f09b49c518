"""qr::cacqr mirror, 1D path (reference src/alg/qr/cacqr/cacqr.h:18-55, cacqr.hpp:5-29,172-248).

    pack = cacqr.info(num_iter, cholinv.info(...))      # num_iter: 1 = CholeskyQR, 2 = CholeskyQR2
    cacqr.factor(A, pack, topo.rect(c=1))               # A: local row-cyclic piece (m_local x n)
    Q = cacqr.construct_Q(pack, topo); R = cacqr.construct_R(pack, topo)

Only the c == 1 grid (cacqr.hpp:229, the shape of BASELINE config 4) is implemented on the
GPU; the 3D / tunable-grid sweeps (cacqr.hpp:75-170) are SURVEY 8f "next"."""
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

    def _ensure(self, m_local, n, comm):
        if self._plan is not None and self._shape == (m_local, n):
            return
        self._release()
        h = C.c_void_p()
        _lib.check(_lib.lib().cap_cacqr_plan_create(C.byref(h), m_local, n, self.num_iter, comm), "cap_cacqr_plan_create")
        self._plan, self._shape = h, (m_local, n)

    def _release(self):
        if self._plan is not None:
            _lib.lib().cap_cacqr_plan_destroy(self._plan)
            self._plan = None

    def last_info(self):
        v = C.c_int64(0)
        _lib.lib().cap_cacqr_info(self._plan, cur_stream(), C.byref(v))
        return v.value

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass


def factor(A, args, CommInfo=None):
    """cacqr::factor (cacqr.hpp:217-248) for the c == 1 grid."""
    c = getattr(CommInfo, "c", 1) if CommInfo is not None else 1
    if c != 1:
        raise _lib.CapitalError("only the 1D (c == 1) CholeskyQR grid is implemented on the GPU (cacqr.hpp:229)")
    comm = getattr(CommInfo, "world", None) if CommInfo is not None else None
    args._gm, args._gn = A.num_rows_global(), A.num_columns_global()
    args._ensure(A.num_rows_local(), A.num_columns_local(), comm)
    _lib.check(_lib.lib().cap_cacqr_factor(args._plan, A.data_ptr(), A.ld(), cur_stream()), "cacqr::factor")


def construct_Q(args, CommInfo=None):
    """cacqr.hpp construct_Q: fresh rect matrix with this rank's row-cyclic piece of Q."""
    m_local, n = args._shape
    d = getattr(CommInfo, "d", 1) if CommInfo is not None else 1
    out = matrix(args._gn, args._gm, 1, d, rect)
    ld = C.c_int64(0)
    q = _lib.lib().cap_cacqr_Q_ptr(args._plan, C.byref(ld))
    st = _lib.lib().cap_copy_window(q, 0, ld.value, 0, 0, out.data_ptr(), 0, out.ld(), 0, 0, m_local, n, 0, 0, cur_stream())
    _lib.check(st, "construct_Q")
    return out


def construct_R(args, CommInfo=None):
    m_local, n = args._shape
    out = matrix(n, n, 1, 1, rect)
    ld = C.c_int64(0)
    r = _lib.lib().cap_cacqr_R_ptr(args._plan, C.byref(ld))
    st = _lib.lib().cap_copy_window(r, 0, ld.value, 0, 0, out.data_ptr(), 0, out.ld(), 0, 0, n, n, 1, 1, cur_stream())
    _lib.check(st, "construct_R")
    return out
