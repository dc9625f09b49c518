"""blas::engine mirror (reference src/blas/engine.h:23-130, src/blas/interface.h:58-66).

Same enum names / values and ArgPack fields as upstream; the engine methods take DEVICE
buffers (torch tensors or raw device addresses) and run the hand-written MFMA kernels of
libcapital_amd.so.  No CPU path: missing library or non-device pointers fail loudly."""
import enum

from . import _lib
from ._util import dptr, cur_stream, scratch


class Order(enum.IntEnum):
    AblasRowMajor = 0x0
    AblasColumnMajor = 0x1


class Transpose(enum.IntEnum):
    AblasNoTrans = 0x0
    AblasTrans = 0x1


class Side(enum.IntEnum):
    AblasLeft = 0x0
    AblasRight = 0x1


class UpLo(enum.IntEnum):
    AblasLower = 0x0
    AblasUpper = 0x1


class Diag(enum.IntEnum):
    AblasNonUnit = 0x0
    AblasUnit = 0x1


class Method(enum.IntEnum):
    AblasGemm = 0x0
    AblasTrmm = 0x1
    AblasSyrk = 0x10


class ArgPack_gemm:
    def __init__(self, order, transposeA, transposeB, alpha, beta):
        self.method = Method.AblasGemm
        self.order, self.transposeA, self.transposeB = Order(order), Transpose(transposeA), Transpose(transposeB)
        self.alpha, self.beta = float(alpha), float(beta)


class ArgPack_trmm:
    def __init__(self, order, side, uplo, transposeA, diag, alpha):
        self.method = Method.AblasTrmm
        self.order, self.side, self.uplo = Order(order), Side(side), UpLo(uplo)
        self.transposeA, self.diag, self.alpha = Transpose(transposeA), Diag(diag), float(alpha)


class ArgPack_syrk:
    def __init__(self, order, uplo, transposeA, alpha, beta):
        self.method = Method.AblasSyrk
        self.order, self.uplo, self.transposeA = Order(order), UpLo(uplo), Transpose(transposeA)
        self.alpha, self.beta = float(alpha), float(beta)


def _require_colmajor(order):
    if Order(order) != Order.AblasColumnMajor:
        raise _lib.CapitalError("only AblasColumnMajor is supported (every upstream call site uses it)")


class engine:
    """Static methods with upstream's signatures (blas/interface.h:58-66)."""

    @staticmethod
    def _gemm(matrixA, matrixB, matrixC, m, n, k, lda, ldb, ldc, srcPackage, stream=None):
        _require_colmajor(srcPackage.order)
        st = _lib.lib().cap_dgemm(int(srcPackage.transposeA), int(srcPackage.transposeB), m, n, k, srcPackage.alpha,
                                  dptr(matrixA), lda, dptr(matrixB), ldb, srcPackage.beta, dptr(matrixC), ldc,
                                  cur_stream(stream))
        _lib.check(st, "blas::engine::_gemm")

    @staticmethod
    def _trmm(matrixA, matrixB, m, n, lda, ldb, srcPackage, stream=None):
        _require_colmajor(srcPackage.order)
        L = _lib.lib()
        work = scratch(L.cap_dtrmm_work_size(int(srcPackage.side), m, n), matrixB)
        st = L.cap_dtrmm(int(srcPackage.side), int(srcPackage.uplo), int(srcPackage.transposeA), int(srcPackage.diag), m, n,
                         srcPackage.alpha, dptr(matrixA), lda, dptr(matrixB), ldb, dptr(work), cur_stream(stream))
        _lib.check(st, "blas::engine::_trmm")

    @staticmethod
    def _syrk(matrixA, matrixC, n, k, lda, ldc, srcPackage, stream=None):
        _require_colmajor(srcPackage.order)
        st = _lib.lib().cap_dsyrk(int(srcPackage.uplo), int(srcPackage.transposeA), n, k, srcPackage.alpha, dptr(matrixA), lda,
                                  srcPackage.beta, dptr(matrixC), ldc, cur_stream(stream))
        _lib.check(st, "blas::engine::_syrk")

    @staticmethod
    def _trsm(matrixA, matrixB, m, n, lda, ldb, srcPackage, stream=None):
        """Real triangular solve (upstream stubs it: trsm/diaginvert/diaginvert.hpp:7-10). Takes an ArgPack_trmm."""
        _require_colmajor(srcPackage.order)
        L = _lib.lib()
        work = scratch(L.cap_dtrsm_work_size(int(srcPackage.side), m, n), matrixB)
        st = L.cap_dtrsm(int(srcPackage.side), int(srcPackage.uplo), int(srcPackage.transposeA), m, n, srcPackage.alpha,
                         dptr(matrixA), lda, dptr(matrixB), ldb, dptr(work), cur_stream(stream))
        _lib.check(st, "blas::engine::_trsm")
