"""Distributed redistribution between the reference's element-cyclic d x d x c pieces (matrix.hpp:8-11, topology.h:67-143)
and the block-cyclic layouts of the multi-GPU plans (csrc/redist.hip): one all-to-all over the world communicator.

    rp = redist.plan(n, nb, topo_or_comm, c, Pr=1)        # Pr = 1: block columns (cap_dist_* / cap_dmp_*), Pr > 1: cap_dist2d_*
    rp.cyclic_to_bc(piece_matrix, bc_tensor)                # piece: `matrix` on the d x d grid; bc_tensor: (cols, ld) device buffer
    rp.bc_to_cyclic(bc_tensor, piece_matrix)

upstream assembles the same moves from MPI_Allgather + util::block_to_cyclic_* / cyclic_to_local (util.hpp:56-230)."""
import ctypes as C

import torch

from . import _lib
from ._util import cur_stream


class plan:
    def __init__(self, n, nb, world, c=1, Pr=1):
        """world: a cap_comm handle (or an object with .handle / .world)."""
        h = getattr(world, "world", None) or getattr(world, "handle", None) or world
        self._h = C.c_void_p()
        _lib.check(_lib.lib().cap_redist_plan_create(C.byref(self._h), int(n), int(nb), h, int(c), int(Pr)), "cap_redist_plan_create")
        g = lambda w: int(_lib.lib().cap_redist_get(self._h, w))
        self.n, self.nb = int(n), int(nb)
        self.piece, self.bc_rows, self.bc_cols = g(0), g(1), g(2)
        self.d, self.c, self.x, self.y, self.z = g(3), g(4), g(5), g(6), g(7)
        self.Pr, self.Pc, self.pr, self.pc = g(8), g(9), g(10), g(11)
        self.sent = (g(12), g(13)); self.received = (g(14), g(15))

    def new_bc(self, device=None):
        """(cols, ld = rows) zero device buffer of my block-cyclic piece (column-major rows x cols)."""
        dev = device or torch.device("cuda", torch.cuda.current_device())
        return torch.zeros(max(self.bc_cols, 1), max(self.bc_rows, 1), dtype=torch.float64, device=dev)

    def cyclic_to_bc(self, piece, bc):
        _lib.check(_lib.lib().cap_redistribute_cyclic_to_bc(self._h, piece.data_ptr(), piece.ld(), bc.data_ptr(), bc.shape[1], cur_stream()),
                   "cap_redistribute_cyclic_to_bc")

    def bc_to_cyclic(self, bc, piece):
        _lib.check(_lib.lib().cap_redistribute_bc_to_cyclic(self._h, bc.data_ptr(), bc.shape[1], piece.data_ptr(), piece.ld(), cur_stream()),
                   "cap_redistribute_bc_to_cyclic")

    def close(self):
        if self._h:
            _lib.lib().cap_redist_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
