"""Communicator bundles (reference src/util/topology.h:16-143) for one-process-per-GPU runs.

`square` / `rect` keep upstream's public fields (rank, size, c, d, x, y, z, layout, num_chunks; world, row,
column, depth, slice and - rect - column_contig, column_alt, cube) and its rank -> (x, y, z) maps (layout 0;
layouts 1-2 are numerically wrong upstream, SURVEY App. C #9, and are rejected).  The MPI communicators become
RCCL communicators over xGMI behind the C ABI: the world communicator is created from a unique id shipped
through torch.distributed (plumbing only), the sub-communicators are split off it with ncclCommSplit inside
cap_topo_create exactly where upstream calls MPI_Comm_split.

backend="auto": RCCL when torch.distributed runs on nccl, the RCCL-free self communicator on a single process,
and - tests only - host-staged gloo groups (several ranks sharing one GPU) when torch.distributed runs on gloo."""
import ctypes as C
import math

import torch

from . import _lib


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


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


_WHICH = {"world": 0, "row": 1, "column": 2, "depth": 3, "slice": 4, "column_contig": 5, "column_alt": 6, "cube": 7}


def _gloo_split(color_of, key_of, size, rank):
    """MPI_Comm_split over the gloo world: every rank creates every group (in the same order) and keeps its own."""
    dist = _dist()
    mine = None
    for color in sorted(set(color_of(r) for r in range(size))):
        ranks = sorted((r for r in range(size) if color_of(r) == color), key=key_of)
        g = dist.new_group(ranks)
        if rank in ranks:
            mine = g
    return mine


class _bundle:
    kind = 0

    def __init__(self, c=1, layout=0, num_chunks=0, create_comm=True, force_rccl=False, comm_factory=None):
        """comm_factory: tests only - a callable (group=None) -> object with .handle / .close() that builds a communicator
        through cap_comm_create_callbacks (tests/host_staged.py: several ranks sharing one GPU over gloo); the sub-groups of
        the bundle are then assembled by the host and handed to cap_topo_create_from.  Default: RCCL (ncclCommSplit)."""
        if self.kind == 0 and layout != 0:
            raise _lib.CapitalError("rank layouts 1/2 are numerically wrong upstream (SURVEY App. C #9); use layout 0")
        self.rank, self.size = rank_size()
        self.layout, self.num_chunks = layout, num_chunks
        self.__dict__.update((square_coords if self.kind == 0 else rect_coords)(self.rank, self.size, c))
        if self.kind == 0 and self.c * self.d * self.d != self.size:
            raise _lib.CapitalError("topo::square needs size == c*d*d (got size=%d c=%d d=%d)" % (self.size, self.c, self.d))
        if self.kind == 1 and self.c * self.c * self.d != self.size:
            raise _lib.CapitalError("topo::rect needs size == c*c*d")
        self.world = self.handle = None
        self._comm_obj = None
        self._subs = []
        if create_comm:
            self._make(force_rccl, comm_factory)

    def _make(self, force_rccl, comm_factory):
        from . import dist_cholesky as dc
        L = _lib.lib()
        h = C.c_void_p()
        if comm_factory is not None:
            self._comm_obj = comm_factory()
            self.world = self._comm_obj.handle
            subs = (C.c_void_p * 7)()
            for i, (color_of, key_of) in enumerate(self._splits()):
                if color_of is None:
                    continue
                g = _gloo_split(color_of, key_of, self.size, self.rank)
                sub = comm_factory(group=g)
                self._subs.append(sub)
                subs[i] = sub.handle
            _lib.check(L.cap_topo_create_from(C.byref(h), self.kind, self.world, self.c, self.layout, self.num_chunks, subs, 7),
                       "cap_topo_create_from")
        else:
            self._comm_obj = dc.RcclComm(force_rccl=force_rccl)
            self.world = self._comm_obj.handle
            _lib.check(L.cap_topo_create(C.byref(h), self.kind, self.world, self.c, self.layout, self.num_chunks), "cap_topo_create")
        self.handle = h
        for name, which in _WHICH.items():
            if name != "world":
                setattr(self, name, L.cap_topo_comm(h, which))

    def _splits(self):
        raise NotImplementedError

    def close(self):
        if self.handle:
            _lib.lib().cap_topo_destroy(self.handle)
            self.handle = None
        for s in self._subs:
            s.close()
        self._subs = []
        if self._comm_obj is not None:
            self._comm_obj.close()
            self._comm_obj = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class square(_bundle):
    """topo::square (topology.h:67-143): d x d x c grid."""
    kind = 0

    def _splits(self):
        c, d = self.c, self.d
        co = lambda r: square_coords(r, self.size, c)
        # order: row, column, depth, slice, column_contig, column_alt, cube  (topology.h:84-94)
        return [(lambda r: (co(r)["y"], co(r)["z"]), lambda r: co(r)["x"]),
                (lambda r: (co(r)["x"], co(r)["z"]), lambda r: co(r)["y"]),
                (lambda r: r // c, lambda r: r),
                (lambda r: co(r)["z"], lambda r: r),
                (None, None), (None, None), (None, None)]


class rect(_bundle):
    """topo::rect (topology.h:16-65): c x d x c grid for CholeskyQR."""
    kind = 1

    def _splits(self):
        c = self.c
        cube_sz, slice_sz = c * c * c, c * c
        cube_rank = lambda r: r % cube_sz
        col_rank = lambda r: r // slice_sz          # rank inside the temporary `column` communicator (key = world rank)
        return [(lambda r: (r // cube_sz, (cube_rank(r) % c) + c * (cube_rank(r) // slice_sz)), lambda r: cube_rank(r)),   # row
                (None, None),                                                                                           # column (square only)
                (lambda r: (r // cube_sz, cube_rank(r) // c), lambda r: cube_rank(r)),                                  # depth
                (lambda r: r % c, lambda r: r),                                                                         # slice
                (lambda r: (r % slice_sz, col_rank(r) // c), lambda r: col_rank(r)),                                    # column_contig
                (lambda r: (r % slice_sz, col_rank(r) % c), lambda r: col_rank(r)),                                     # column_alt
                (lambda r: r // cube_sz, lambda r: r)]                                                                  # cube
