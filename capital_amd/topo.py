"""Communicator bundles (reference src/util/topology.h:16-143) for one-process-per-GPU runs.

`square` / `rect` keep upstream's public fields (rank, size, c, d, x, y, z, layout,
num_chunks) and rank->(x,y,z) maps (layout 0; layouts 1-2 are numerically wrong upstream,
SURVEY App. C #9, and are rejected).  The MPI communicators become ONE RCCL communicator over
xGMI created behind the C ABI (cap_comm_*); torch.distributed is used only to ship the
128-byte RCCL unique id from rank 0 (plumbing).  On a single process everything degenerates
to a self communicator and no RCCL call is made.

What the GPU schedules support today: P = 1 for cholinv; the 1D c = 1 grid (all ranks in
one column, rows cyclic) for CholeskyQR2 - the shape of BASELINE config 4.  Upstream itself
only supports cubic grids c == d (P = 1, 8, 27...) for cholinv."""
import ctypes as C
import math

import torch

from . import _lib


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


class _comm_base:
    def _make_comm(self):
        L = _lib.lib()
        h = C.c_void_p()
        if self.size == 1:
            _lib.check(L.cap_comm_create_self(C.byref(h)), "cap_comm_create_self")
        else:
            dist = _dist()
            if dist is None:
                raise _lib.CapitalError("multi-rank topology needs torch.distributed to be initialised")
            idbuf = (C.c_ubyte * 128)()
            if self.rank == 0:
                _lib.check(L.cap_comm_unique_id(idbuf), "cap_comm_unique_id")
            # ship the id: works on both gloo (CPU tensors) and nccl/RCCL (device tensors)
            backend = dist.get_backend()
            t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8)
            if backend == "nccl":
                t = t.cuda()
            dist.broadcast(t, src=0)
            raw = bytes(t.cpu().tolist())
            idbuf = (C.c_ubyte * 128).from_buffer_copy(raw)
            _lib.check(L.cap_comm_create(C.byref(h), idbuf, self.rank, self.size, None), "cap_comm_create")
        self.world = h

    def __del__(self):
        try:
            if getattr(self, "world", None):
                _lib.lib().cap_comm_destroy(self.world)
                self.world = None
        except Exception:
            pass


def rank_size():
    dist = _dist()
    return (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)


def square_coords(rank, size, c):
    """topology.h:75-83 (layout 0): d = ceil(sqrt(size/c)); z = rank%c, y = rank/(d*c), x = (rank%(d*c))/c."""
    d = int(round(math.ceil(math.sqrt(size / c) - 1e-12)))
    top = d * c
    return dict(c=c, d=d, z=rank % c, y=rank // top, x=(rank % top) // c)


def rect_coords(rank, size, c):
    """topology.h:44-50: d = size/(c*c); z = rank%c, y = rank/(c*c), x = (rank%(c*c))/c."""
    return dict(c=c, d=size // (c * c), z=rank % c, y=rank // (c * c), x=(rank % (c * c)) // c)


class square(_comm_base):
    def __init__(self, c=1, layout=0, num_chunks=0, create_comm=True):
        if layout != 0:
            raise _lib.CapitalError("rank layouts 1/2 are numerically wrong upstream (SURVEY App. C #9); use layout 0")
        self.rank, self.size = rank_size()
        self.layout, self.num_chunks = layout, num_chunks
        self.__dict__.update(square_coords(self.rank, self.size, c))
        if self.c * self.d * self.d != self.size:
            raise _lib.CapitalError("topo::square needs size == c*d*d (got size=%d c=%d d=%d)" % (self.size, self.c, self.d))
        self.world = None
        if create_comm:
            self._make_comm()


class rect(_comm_base):
    def __init__(self, c=1, layout=0, num_chunks=0, create_comm=True):
        self.rank, self.size = rank_size()
        self.layout, self.num_chunks = layout, num_chunks
        self.__dict__.update(rect_coords(self.rank, self.size, c))
        if self.c * self.c * self.d != self.size:
            raise _lib.CapitalError("topo::rect needs size == c*c*d")
        self.world = None
        if create_comm:
            self._make_comm()
