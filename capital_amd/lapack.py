"""lapack::engine mirror (reference src/lapack/engine.h:23-102, src/lapack/interface.h:49-59).

_potrf / _trtri run on the GPU (wavefront-cooperative in-LDS leaves + MFMA GEMM recursion).
Unlike upstream (which drops LAPACKE's return value, lapack/interface.hpp:39,54) _potrf
returns `info`.  _geqrf / _orgqr are never called by any upstream algorithm
and are out of scope."""
import enum

import torch

from . import _lib
from ._util import dptr, cur_stream, scratch


class Order(enum.IntEnum):
    AlapackRowMajor = 0x0
    AlapackColumnMajor = 0x1


class UpLo(enum.IntEnum):
    AlapackLower = 0x0
    AlapackUpper = 0x1


class Diag(enum.IntEnum):
    AlapackNonUnit = 0x0
    AlapackUnit = 0x1


class Method(enum.IntEnum):
    AlapackPotrf = 0x0
    AlapackTrtri = 0x1
    AlapackGeqrf = 0x10
    AlapackOrgqr = 0x11


class ArgPack_potrf:
    def __init__(self, order, uplo):
        self.method = Method.AlapackPotrf
        self.order, self.uplo = Order(order), UpLo(uplo)


class ArgPack_trtri:
    def __init__(self, order, uplo, diag):
        self.method = Method.AlapackTrtri
        self.order, self.uplo, self.diag = Order(order), UpLo(uplo), Diag(diag)


class engine:
    @staticmethod
    def _potrf(matrixA, n, lda, srcPackage, stream=None):
        if srcPackage.order != Order.AlapackColumnMajor:
            raise _lib.CapitalError("only AlapackColumnMajor is supported")
        L = _lib.lib()
        work = scratch(L.cap_dpotrf_work_size(n), matrixA)
        info = torch.zeros(1, dtype=torch.int32, device=work.device)
        st = L.cap_dpotrf(int(srcPackage.uplo), n, dptr(matrixA), lda, info.data_ptr(), dptr(work), cur_stream(stream))
        _lib.check(st, "lapack::engine::_potrf")
        return int(info.item())

    @staticmethod
    def _trtri(matrixA, n, lda, srcPackage, stream=None):
        if srcPackage.order != Order.AlapackColumnMajor or srcPackage.diag != Diag.AlapackNonUnit:
            raise _lib.CapitalError("only AlapackColumnMajor / AlapackNonUnit is supported")
        L = _lib.lib()
        work = scratch(L.cap_dtrtri_work_size(n), matrixA)
        st = L.cap_dtrtri(int(srcPackage.uplo), n, dptr(matrixA), lda, dptr(work), cur_stream(stream))
        _lib.check(st, "lapack::engine::_trtri")
