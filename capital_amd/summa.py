"""matmult::summa mirror, GEMM overload (reference src/alg/matmult/summa/summa.h:24-34, summa.hpp:6-44).

    summa.invoke(A, B, C, topo.square(c), blas.ArgPack_gemm(...))        # C = alpha A B + beta C, element-cyclic pieces

A, B, C are `matrix` objects created on the d x d grid (matrix(K, M, d, d) etc., same argument order as
bench/matmult/summa_gemm.cpp:32-34); the plan (scratch pieces, streams, events) is cached per shape on the topo."""
import ctypes as C

from . import _lib
from ._util import cur_stream


def invoke(A, B, Cm, CommInfo, srcPackage):
    L = _lib.lib()
    if int(srcPackage.transposeA) != 0 or int(srcPackage.transposeB) != 0:
        raise _lib.CapitalError("distributed SUMMA runs NoTrans x NoTrans (like bench/matmult/summa_gemm.cpp:38); upstream "
                                "handles the transposed forms with a separate partner exchange (util::transpose)")
    m, n, k = A.num_rows_global(), B.num_columns_global(), A.num_columns_global()
    key = (m, n, k, int(CommInfo.num_chunks))
    cache = CommInfo.__dict__.setdefault("_summa_plans", {})
    if key not in cache:
        h = C.c_void_p()
        _lib.check(L.cap_summa_plan_create(C.byref(h), CommInfo.handle, m, n, k, int(CommInfo.num_chunks)), "cap_summa_plan_create")
        cache[key] = h
    _lib.check(L.cap_summa_dgemm(cache[key], float(srcPackage.alpha), A.data_ptr(), A.ld(), B.data_ptr(), B.ld(), float(srcPackage.beta),
                                 Cm.data_ptr(), Cm.ld(), cur_stream()), "summa::invoke")


def release(CommInfo):
    for h in CommInfo.__dict__.pop("_summa_plans", {}).values():
        _lib.lib().cap_summa_plan_destroy(h)
