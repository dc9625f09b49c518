"""matmult::summa mirror - the three overloads of the reference (src/alg/matmult/summa/summa.h:24-34, summa.hpp:6-161):

    summa.invoke(A, B, C, topo.square(c), blas.ArgPack_gemm(...))     # C = alpha A B + beta C
    summa.invoke(T, B, topo.square(c), blas.ArgPack_trmm(...))        # B = alpha op(T) B  |  alpha B op(T), in place
    summa.invoke(A, C, topo.square(c), blas.ArgPack_syrk(...))        # C = alpha A^T A + beta C  |  alpha A A^T + beta C
    summa.invoke(A, B, C, topo.square(c), blas.ArgPack_syrk(...))     # the same with the caller's second copy B (clobbered upstream)
    summa.transpose(M, topo.square(c))                                 # util::transpose: swap pieces with the transpose partner

Operands are `matrix` objects created on the d x d grid (matrix(K, M, d, d) etc., argument order of bench/matmult/
summa_gemm.cpp:32-34) holding element-cyclic pieces; plans (scratch pieces, streams, events) are cached per shape on the topo.
Triangular operands are rect pieces whose upper triangle is referenced (the C ABI also takes upstream's packed-upper storage:
cap_summa_dtrmm(t_packed = 1)).  For transposeA = AblasTrans the TRMM overload expects T AFTER summa.transpose(T, topo) -
exactly how upstream's call site prepares it (cholinv.hpp:114-120)."""
import ctypes as C

import torch

from . import _lib, blas
from ._util import cur_stream


def _plan(CommInfo, m, n, k):
    key = (int(m), int(n), int(k), int(CommInfo.num_chunks))
    cache = CommInfo.__dict__.setdefault("_summa_plans", {})
    if key not in cache:
        h = C.c_void_p()
        _lib.check(_lib.lib().cap_summa_plan_create(C.byref(h), CommInfo.handle, key[0], key[1], key[2], key[3]), "cap_summa_plan_create")
        cache[key] = h
    return cache[key]


def transpose(mat, CommInfo):
    """util::transpose (util.hpp:232-247): in-place exchange of my piece with rank (y, x, z)'s; the received piece is the
    partner's piece as stored (not transposed)."""
    buf = mat.data()
    tmp = torch.empty_like(buf)
    _lib.check(_lib.lib().cap_util_transpose(CommInfo.handle, buf.data_ptr(), tmp.data_ptr(), buf.numel(), cur_stream()), "util::transpose")
    torch.cuda.current_stream().synchronize()       # tmp is released on return
    return mat


def invoke(*args):
    *mats, CommInfo, srcPackage = args
    L = _lib.lib()
    if isinstance(srcPackage, blas.ArgPack_gemm):
        A, B, Cm = mats
        if int(srcPackage.transposeA) != 0 or int(srcPackage.transposeB) != 0:
            raise _lib.CapitalError("the GEMM overload runs NoTrans x NoTrans (like bench/matmult/summa_gemm.cpp:38); upstream handles "
                                    "transposed operands with the partner exchange summa.transpose (util::transpose)")
        m, n, k = A.num_rows_global(), B.num_columns_global(), A.num_columns_global()
        _lib.check(L.cap_summa_dgemm(_plan(CommInfo, m, n, k), float(srcPackage.alpha), A.data_ptr(), A.ld(), B.data_ptr(), B.ld(),
                                     float(srcPackage.beta), Cm.data_ptr(), Cm.ld(), cur_stream()), "summa::invoke(gemm)")
    elif isinstance(srcPackage, blas.ArgPack_trmm):
        T, B = mats
        m, n = B.num_rows_global(), B.num_columns_global()
        left = int(srcPackage.side) == int(blas.Side.AblasLeft)
        td = m if left else n
        if T.num_rows_global() != td or T.num_columns_global() != td:
            raise _lib.CapitalError("summa::invoke(trmm): the triangular operand must be %d x %d" % (td, td))
        _lib.check(L.cap_summa_dtrmm(_plan(CommInfo, m, n, td), int(srcPackage.side), int(srcPackage.uplo), int(srcPackage.transposeA),
                                     int(srcPackage.diag), float(srcPackage.alpha), T.data_ptr(), T.ld(), 0, B.data_ptr(), B.ld(), cur_stream()),
                   "summa::invoke(trmm)")
    elif isinstance(srcPackage, blas.ArgPack_syrk):
        A, Cm = mats[0], mats[-1]                   # (A, C) or (A, B, C): B is upstream's scratch copy for the partner exchange
        tr = int(srcPackage.transposeA) == int(blas.Transpose.AblasTrans)
        n = Cm.num_rows_global()
        k = A.num_rows_global() if tr else A.num_columns_global()
        if (A.num_columns_global() if tr else A.num_rows_global()) != n or Cm.num_columns_global() != n:
            raise _lib.CapitalError("summa::invoke(syrk): shapes do not match")
        _lib.check(L.cap_summa_dsyrk(_plan(CommInfo, n, n, k), int(srcPackage.uplo), int(srcPackage.transposeA), float(srcPackage.alpha),
                                     A.data_ptr(), A.ld(), float(srcPackage.beta), Cm.data_ptr(), Cm.ld(), 0, cur_stream()), "summa::invoke(syrk)")
    else:
        raise _lib.CapitalError("summa::invoke: unknown ArgPack")


def release(CommInfo):
    for h in CommInfo.__dict__.pop("_summa_plans", {}).values():
        _lib.lib().cap_summa_plan_destroy(h)
