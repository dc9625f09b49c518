"""GPU-resident element-cyclic matrix descriptor (reference src/matrix/matrix.h:9-97,
structure.h:8-72, structure.hpp:68-129, serialize.hpp).

Same constructor argument order and accessor names as upstream:
    matrix(globalDimensionX = #columns, globalDimensionY = #rows, globalPgridX, globalPgridY)
X = column index, Y = row index everywhere (matrix.h:19,47).  Local storage is column-major
in HBM with the reference's ceil(N/d) zero-padded local dims (matrix.hpp:8-11).  The
reference's three raw buffers data/scratch/pad (pointer-rotation protocol, matrix.h:55-56)
do not exist here: workspaces are owned by plans behind the C ABI."""
import numpy as np
import torch

from . import _lib
from ._util import cur_stream


class rect:
    name = "rect"

    @staticmethod
    def _offset(x, y, dimX, dimY):   # structure.h:13
        return x * dimY + y

    @staticmethod
    def _num_elems(dimX, dimY):
        return dimX * dimY


class uppertri:
    name = "uppertri"

    @staticmethod
    def _offset(x, y, dimX, dimY):   # structure.h:39
        return (x * (x + 1)) // 2 + y

    @staticmethod
    def _num_elems(dimX, dimY):      # structure.h:37
        return (dimY * (dimY + 1)) // 2


def _local(n_global, p):
    return n_global // p + (1 if n_global % p else 0)


class matrix:
    def __init__(self, globalDimensionX, globalDimensionY, globalPgridX=1, globalPgridY=1, structure=rect, device=None):
        if structure is not rect:
            raise _lib.CapitalError("device matrices are rect; packed-upper windows are handled by serialize()")
        self._globalDimensionX, self._globalDimensionY = int(globalDimensionX), int(globalDimensionY)
        self._dimensionX = _local(self._globalDimensionX, globalPgridX)
        self._dimensionY = _local(self._globalDimensionY, globalPgridY)
        self._pgridX, self._pgridY = int(globalPgridX), int(globalPgridY)
        self.structure = structure
        self.device = torch.device(device if device is not None else ("cuda:%d" % torch.cuda.current_device()))
        if self.device.type != "cuda":
            raise _lib.CapitalError("capital_amd matrices live in HBM; no CPU path")
        self._ld = self._dimensionY + (self._dimensionY & 1)          # even ld keeps 16-byte loads aligned
        self._buf = torch.zeros(self._dimensionX, self._ld, dtype=torch.float64, device=self.device)

    # --- accessors with upstream names (matrix.h:44-51) ---
    def num_rows_local(self): return self._dimensionY
    def num_columns_local(self): return self._dimensionX
    def num_rows_global(self): return self._globalDimensionY
    def num_columns_global(self): return self._globalDimensionX
    def num_elems(self): return self._dimensionX * self._dimensionY
    def ld(self): return self._ld
    def data(self): return self._buf           # device buffer (columns, ld); data().data_ptr() is the raw address
    def data_ptr(self): return self._buf.data_ptr()

    def view(self):
        """[row, col] indexed device view of the local piece."""
        return self._buf[:, :self._dimensionY].t()

    # --- generators (structure.hpp:68-129), computed on the GPU ---
    def distribute_symmetric(self, localPgridX, localPgridY, globalPgridX, globalPgridY, key=0, diagonallyDominant=True):
        if globalPgridX != globalPgridY or self._globalDimensionX != self._globalDimensionY:
            raise _lib.CapitalError("distribute_symmetric needs a square matrix on a square grid")
        st = _lib.lib().cap_fill_symmetric(self._buf.data_ptr(), self._ld, self._globalDimensionX, localPgridX, localPgridY,
                                           globalPgridX, 1 if diagonallyDominant else 0, cur_stream())
        _lib.check(st, "distribute_symmetric")

    def distribute_random(self, localPgridX, localPgridY, globalPgridX, globalPgridY, key=0):
        st = _lib.lib().cap_fill_random(self._buf.data_ptr(), self._ld, self._globalDimensionY, self._globalDimensionX,
                                        localPgridX, localPgridY, globalPgridX, globalPgridY, key, cur_stream())
        _lib.check(st, "distribute_random")

    # --- host transfer through the C descriptor's pinned double-buffered staging (cap_desc_import/export_host) ---
    def _desc(self):
        if getattr(self, "_cdesc", None) is None:
            import ctypes as C
            h = C.c_void_p()
            _lib.check(_lib.lib().cap_desc_create_view(C.byref(h), self._globalDimensionX, self._globalDimensionY, self._pgridX,
                                                       self._pgridY, self._buf.data_ptr(), self._ld), "cap_desc_create_view")
            self._cdesc = h
        return self._cdesc

    def to_numpy(self):
        """column-major export, returned as a [row, col] C-ordered array"""
        out = np.empty((self._dimensionX, self._dimensionY), dtype=np.float64)       # [col][row] = column-major, ld = rows
        _lib.check(_lib.lib().cap_desc_export_host(self._desc(), out.ctypes.data, self._dimensionY, cur_stream()), "export_host")
        return np.ascontiguousarray(out.T)

    def from_numpy(self, a):
        a = np.asarray(a, dtype=np.float64)
        assert a.shape == (self._dimensionY, self._dimensionX), (a.shape, self._dimensionY, self._dimensionX)
        colmajor = np.ascontiguousarray(a.T)                                          # [col][row]
        _lib.check(_lib.lib().cap_desc_import_host(self._desc(), colmajor.ctypes.data, self._dimensionY, cur_stream()), "import_host")
        torch.cuda.current_stream().synchronize()                                     # `colmajor` may be released
        return self

    def __del__(self):
        try:
            if getattr(self, "_cdesc", None) is not None:
                _lib.lib().cap_desc_destroy(self._cdesc)
                self._cdesc = None
        except Exception:
            pass


def serialize(src, dst, src_window, dst_origin, tri_only=False, zero_lower=False, src_packed=False, dst_packed=False,
              src_ld=None, dst_ld=None):
    """serialize<S1,S2>::invoke (serialize.hpp:12-150) on device buffers.

    src/dst: torch tensors (device); window = (row0, row1, col0, col1) in the source; origin = (row0, col0) in dst.
    packed buffers use uppertri::_offset (structure.h:39)."""
    r0, r1, c0, c1 = src_window
    st = _lib.lib().cap_copy_window(src.data_ptr(), int(src_packed), src_ld or 0, r0, c0, dst.data_ptr(), int(dst_packed),
                                    dst_ld or 0, dst_origin[0], dst_origin[1], r1 - r0, c1 - c0, int(tri_only), int(zero_lower),
                                    cur_stream())
    _lib.check(st, "serialize")


def cyclic_import(piece, dense, x, y, dx, dy):
    """Scatter the element-cyclic piece (x, y) of a dx x dy grid (a `matrix` created with those grid dims) into `dense`
    (a `matrix` on a 1 x 1 grid) - util::block_to_cyclic_* / cyclic_to_local of the reference, on the GPU."""
    st = _lib.lib().cap_cyclic_import(piece.data_ptr(), piece.ld(), dense.data_ptr(), dense.ld(), dense.num_rows_global(),
                                      dense.num_columns_global(), x, y, dx, dy, cur_stream())
    _lib.check(st, "cyclic_import")


def cyclic_export(dense, piece, x, y, dx, dy):
    st = _lib.lib().cap_cyclic_export(dense.data_ptr(), dense.ld(), piece.data_ptr(), piece.ld(), dense.num_rows_global(),
                                      dense.num_columns_global(), x, y, dx, dy, cur_stream())
    _lib.check(st, "cyclic_export")
