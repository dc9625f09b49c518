"""cholesky::cholinv mirror (reference src/alg/cholesky/cholinv/cholinv.h:16-53, cholinv.hpp:6-46).

    pack = cholinv.info(complete_inv, split, bc_mult_dim, 'U')     # plan handle (create once)
    cholinv.factor(A, pack, topo)                                   # A: matrix (read-only)
    R = cholinv.construct_R(pack, topo); Rinv = cholinv.construct_Rinv(pack, topo)

`info` keeps upstream's four user knobs.  complete_inv = -1 is the documented extension:
blocked right-looking Cholesky (real TRSM/SYRK, no explicit inverse) - the headline
"fp64 Cholesky" path; 0 / 1 reproduce upstream's R + R^-1 semantics."""
import ctypes as C

import torch

from . import _lib
from ._util import cur_stream
from .matrix import matrix, rect


class info:
    def __init__(self, complete_inv, split, bc_mult_dim, dir='U'):
        self.complete_inv, self.split, self.bc_mult_dim, self.dir = int(complete_inv), int(split), int(bc_mult_dim), dir
        self._plan = None
        self._n = None
        self._comm = None
        self.options = {}

    def set_option(self, key, value):
        """GPU-schedule knobs: nb (panel width), leaf (<= 64), lookahead (0/1)."""
        self.options[key] = int(value)
        if self._plan:
            _lib.check(_lib.lib().cap_cholinv_set_option(self._plan, key.encode(), int(value)), "set_option")

    def get_option(self, key):
        if not self._plan:
            return self.options.get(key)
        return _lib.lib().cap_cholinv_get_option(self._plan, key.encode())

    def _ensure(self, n, comm=None):
        if self._plan is not None and self._n == n and self._comm == comm:
            return
        self._release()
        L = _lib.lib()
        h = C.c_void_p()
        # args.R/_Rinv._register_: allocated once, later calls are no-ops (cholinv.hpp:11-12)
        _lib.check(L.cap_cholinv_plan_create(C.byref(h), n, self.complete_inv, self.split, self.bc_mult_dim,
                                             self.dir.encode()[0:1], comm), "cap_cholinv_plan_create")
        self._plan, self._n, self._comm = h, n, comm
        order = sorted(self.options.items(), key=lambda kv: kv[0] == "cyclic_c")      # the layout follows the block width: "nb" first
        for k, v in order:
            _lib.check(L.cap_cholinv_set_option(self._plan, k.encode(), v), "set_option")

    def _release(self):
        if self._plan is not None:
            _lib.lib().cap_cholinv_plan_destroy(self._plan)
            self._plan = None

    def last_info(self):
        """0, or the 1-based index of the first non-positive pivot (upstream drops this, lapack/interface.hpp:39)."""
        v = C.c_int64(0)
        _lib.check_info(_lib.lib().cap_cholinv_info(self._plan, cur_stream(), C.byref(v)), "cap_cholinv_info")
        return v.value

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass


def factor(A, args, CommInfo=None):
    """cholinv::factor (cholinv.hpp:6-28). Asynchronous on the current stream."""
    if args.dir != 'U':
        raise _lib.CapitalError("dir must be 'U' (upstream asserts the same, cholinv.hpp:9)")
    if args.split <= 0:
        raise _lib.CapitalError("split must be > 0 (cholinv.hpp:9)")
    if CommInfo is not None and getattr(CommInfo, "size", 1) != 1:
        # multi-GPU: the 1 x P schedule of csrc/dist.hip runs behind the same plan handle (complete_inv = 0 / 1 add R^-1).
        #  * A created on the d x d grid of a topo::square bundle (matrix(n, n, d, d), bench/cholesky/cholinv.cpp:34-35): the
        #    REFERENCE's layout - A is this rank's element-cyclic piece, construct_R / construct_Rinv return pieces of the same
        #    shape on every rank (option "cyclic_c": distributed redistribution in front of / behind the plan, csrc/redist.hip);
        #  * A on a 1 x 1 grid: this rank's block columns (all n rows), construct_* return block columns.
        n = A.num_rows_global()
        d = getattr(CommInfo, "d", 1)
        args._cyclic = (d > 1 and A._pgridX == d and A._pgridY == d)
        args.options["cyclic_c"] = int(getattr(CommInfo, "c", 1)) if args._cyclic else 0
        args._grid_d = d
        args._ensure(n, getattr(CommInfo, "world", None))
        args.set_option("cyclic_c", args.options["cyclic_c"])          # (a reused plan follows the layout of THIS call's A)
        _lib.check(_lib.lib().cap_cholinv_factor(args._plan, A.data_ptr(), A.ld(), cur_stream()), "cholinv::factor")
        return
    n = A.num_rows_global()
    if n != A.num_columns_global():
        raise _lib.CapitalError("cholinv needs a square matrix")
    args._ensure(n)
    _lib.check(_lib.lib().cap_cholinv_factor(args._plan, A.data_ptr(), A.ld(), cur_stream()), "cholinv::factor")


def _construct(args, which):
    n = args._n
    if args._comm is not None and _lib.lib().cap_comm_size(args._comm) > 1 and getattr(args, "_cyclic", False):
        out = matrix(n, n, args._grid_d, args._grid_d, rect)      # my element-cyclic piece, like upstream's construct_R
    elif args._comm is not None and _lib.lib().cap_comm_size(args._comm) > 1:
        lc = int(_lib.lib().cap_cholinv_get_option(args._plan, b"local_cols"))
        out = matrix(max(lc, 1), n, 1, 1, rect)      # my block-cyclic columns (n x local_cols)
    else:
        out = matrix(n, n, 1, 1, rect)
    fn = _lib.lib().cap_cholinv_get_R if which == "R" else _lib.lib().cap_cholinv_get_Rinv
    _lib.check(fn(args._plan, out.data_ptr(), out.ld(), cur_stream()), "construct_" + which)
    return out


def construct_R(args, CommInfo=None):
    """cholinv.hpp:30-37: fresh rect matrix holding the upper-triangular factor."""
    return _construct(args, "R")


def construct_Rinv(args, CommInfo=None):
    """cholinv.hpp:39-46."""
    return _construct(args, "Rinv")
