"""Multi-GPU blocked Cholesky driver (one process per GPU): 1 x P block-column-cyclic layout, RCCL over xGMI.

    ctx = dist_cholesky.setup(n, nb=512)      # uses torch.distributed's rank/size; builds the RCCL comm bundle
    ctx.fill_symmetric()                      # upstream's distribute_symmetric, generated on each GPU
    ctx.factor(); ctx.last_info()
    R_local = ctx.local_R()                   # n x local_cols device view

The factorization schedule itself lives behind the C ABI (csrc/dist.hip).  `HostStagedComm` swaps RCCL for
gloo-over-host-memory callbacks so that the SAME schedule can be exercised by several ranks sharing one GPU
(tests only - it is slow by construction)."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._util import cur_stream


# ------------------------------------------------------------------ block-cyclic index helpers (pure)
def owner(J, P):
    return J % P


def local_block(J, P):
    return J // P


def num_local_blocks(nblk, P, p):
    return (nblk - 1 - p) // P + 1 if p < nblk else 0


def global_cols_of_rank(n, nb, P, p):
    """Global column indices stored on rank p, in local storage order."""
    nblk = (n + nb - 1) // nb
    cols = []
    for lb in range(num_local_blocks(nblk, P, p)):
        J = lb * P + p
        cols.extend(range(J * nb, min(n, (J + 1) * nb)))
    return np.asarray(cols, dtype=np.int64)


def assemble_global(pieces, n, nb, P):
    """pieces[p] = (n x local_cols_p) array -> global n x n."""
    out = np.zeros((n, n))
    for p, a in enumerate(pieces):
        cols = global_cols_of_rank(n, nb, P, p)
        if cols.size:
            out[:, cols] = a[:, :cols.size]
    return out


# ------------------------------------------------------------------ communicators
def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


class RcclComm:
    """cap_comm over RCCL: unique id from rank 0 shipped through torch.distributed (plumbing only).

    force_rccl=True builds a REAL RCCL communicator (ncclCommInitRank) even for one rank, so that every collective of
    the schedules runs through ncclBroadcast / ncclAllGather / ncclAllReduce on a one-GPU box; the default for one
    rank is the RCCL-free self communicator."""

    def __init__(self, force_rccl=False):
        L = _lib.lib()
        dist = _dist()
        self.rank, self.size = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)
        h = C.c_void_p()
        if self.size == 1 and not force_rccl:
            _lib.check(L.cap_comm_create_self(C.byref(h)), "cap_comm_create_self")
        else:
            idbuf = (C.c_ubyte * 128)()
            if self.rank == 0:
                _lib.check(L.cap_comm_unique_id(idbuf), "cap_comm_unique_id")
            if self.size > 1:
                t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8)
                if dist.get_backend() == "nccl":
                    t = t.cuda()
                dist.broadcast(t, src=0)
                idbuf = (C.c_ubyte * 128).from_buffer_copy(bytes(t.cpu().tolist()))
            _lib.check(L.cap_comm_create(C.byref(h), idbuf, self.rank, self.size, None), "cap_comm_create")
        self.handle = h

    def close(self):
        if self.handle:
            _lib.lib().cap_comm_destroy(self.handle)
            self.handle = None


class HostStagedComm:
    """cap_comm whose collectives are gloo calls on host copies (several ranks may share one GPU).

    Each callback waits for the stream it is handed - and nothing else - before it reads the device buffer, exactly
    the ordering an RCCL kernel enqueued on that stream would have; the other streams of the schedule keep running, so
    a missing event edge between them shows up as a wrong result (tests add random per-stream delays on top).
    group: a torch.distributed process group (sub-communicators of a grid bundle); ranks are group-local."""
    _AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
    _BC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)
    _AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)

    def __init__(self, group=None):
        dist = _dist()
        if dist is None or dist.get_backend() != "gloo":
            raise _lib.CapitalError("HostStagedComm needs torch.distributed initialised with the gloo backend")
        self.group = group
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)
        self.calls = {"allgather": 0, "bcast": 0, "allreduce": 0}

        def sync(stream):
            if stream:
                torch.cuda.ExternalStream(int(stream)).synchronize()
            else:
                torch.cuda.default_stream().synchronize()

        def ag(ctx, send, recv, count, stream):
            try:
                sync(stream)
                mine = _DevView(send, count).to_host()
                outs = [torch.empty(count, dtype=torch.float64) for _ in range(self.size)]
                dist.all_gather(outs, mine, group=self.group)
                _DevView(recv, count * self.size).from_host(torch.cat(outs))
                self.calls["allgather"] += 1
                return 0
            except Exception as e:  # pragma: no cover
                print("HostStagedComm allgather failed:", e, flush=True)
                return 1

        def bc(ctx, buf, count, root, stream):
            try:
                sync(stream)
                v = _DevView(buf, count)
                t = v.to_host()
                dist.broadcast(t, src=dist.get_global_rank(self.group, root) if self.group is not None else root, group=self.group)
                if self.rank != root:
                    v.from_host(t)
                self.calls["bcast"] += 1
                return 0
            except Exception as e:  # pragma: no cover
                print("HostStagedComm bcast failed:", e, flush=True)
                return 1

        def ar(ctx, buf, count, stream):
            try:
                sync(stream)
                v = _DevView(buf, count)
                t = v.to_host()
                dist.all_reduce(t, group=self.group)
                v.from_host(t)
                self.calls["allreduce"] += 1
                return 0
            except Exception as e:  # pragma: no cover
                print("HostStagedComm allreduce failed:", e, flush=True)
                return 1

        self._cbs = (self._AG(ag), self._BC(bc), self._AR(ar))   # keep alive
        h = C.c_void_p()
        _lib.check(_lib.lib().cap_comm_create_callbacks(C.byref(h), self.rank, self.size,
                                                        C.cast(self._cbs[0], C.c_void_p), C.cast(self._cbs[1], C.c_void_p),
                                                        C.cast(self._cbs[2], C.c_void_p), None), "cap_comm_create_callbacks")
        self.handle = h

    def close(self):
        if self.handle:
            _lib.lib().cap_comm_destroy(self.handle)
            self.handle = None


class _DevView:
    """Raw device pointer + element count <-> host tensor, through hipMemcpy (torch's runtime)."""

    def __init__(self, ptr, count):
        self.ptr, self.count = int(ptr), int(count)

    def to_host(self):
        t = torch.empty(self.count, dtype=torch.float64)
        if self.count:
            _memcpy(t.data_ptr(), self.ptr, self.count * 8, 2)
        return t

    def from_host(self, t):
        t = t.contiguous()
        if self.count:
            _memcpy(self.ptr, t.data_ptr(), self.count * 8, 1)


def _memcpy(dst, src, nbytes, kind):
    rt = torch.cuda.cudart()
    err = rt.cudaMemcpy(dst, src, nbytes, kind) if hasattr(rt, "cudaMemcpy") else None
    if err is None:  # fall back to ctypes on the HIP runtime torch already loaded
        hip = C.CDLL("libamdhip64.so")
        e = hip.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(nbytes), C.c_int(kind))
        if e != 0:
            raise _lib.CapitalError("hipMemcpy failed: %d" % e)
    elif int(err) != 0:
        raise _lib.CapitalError("cudaMemcpy failed: %s" % err)


# ------------------------------------------------------------------ driver
class Context:
    def __init__(self, n, nb, comm):
        L = _lib.lib()
        self.n, self.nb, self.comm = int(n), int(nb), comm
        self.rank, self.size = comm.rank, comm.size
        h = C.c_void_p()
        _lib.check(L.cap_dist_plan_create(C.byref(h), self.n, self.nb, comm.handle), "cap_dist_plan_create")
        self.plan = h
        self.local_cols = int(L.cap_dist_local_cols(h))
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.A = torch.zeros(max(self.local_cols, 1), self.n, dtype=torch.float64, device=self.device)  # (cols, ld = n)

    def fill_symmetric(self, diagonally_dominant=True):
        _lib.check(_lib.lib().cap_fill_symmetric_bc(self.A.data_ptr(), self.n, self.n, self.nb, self.size, self.rank,
                                                    1 if diagonally_dominant else 0, cur_stream()), "cap_fill_symmetric_bc")

    def set_local(self, a):
        """a: numpy (n x local_cols)"""
        self.A[: self.local_cols].copy_(torch.from_numpy(np.ascontiguousarray(a.T)).to(self.device))

    def set_option(self, key, value):
        _lib.check(_lib.lib().cap_dist_set_option(self.plan, key.encode(), int(value)), "cap_dist_set_option")

    def factor(self):
        _lib.check(_lib.lib().cap_dist_factor(self.plan, self.A.data_ptr(), self.n, cur_stream()), "cap_dist_factor")

    def last_info(self):
        v = C.c_int64(0)
        _lib.lib().cap_dist_info(self.plan, cur_stream(), C.byref(v))
        return v.value

    def local_R_device(self):
        """(local_cols, n) device buffer = this rank's columns of R (column-major n x local_cols), zero below the diagonal."""
        out = torch.empty(max(self.local_cols, 1), self.n, dtype=torch.float64, device=self.device)
        _lib.check(_lib.lib().cap_dist_get_R(self.plan, out.data_ptr(), self.n, cur_stream()), "cap_dist_get_R")
        return out

    def local_R(self):
        """numpy (n x local_cols) copy of this rank's columns of R, zero below the global diagonal (construct_R)."""
        out = torch.empty(max(self.local_cols, 1), self.n, dtype=torch.float64, device=self.device)
        _lib.check(_lib.lib().cap_dist_get_R(self.plan, out.data_ptr(), self.n, cur_stream()), "cap_dist_get_R")
        torch.cuda.synchronize()
        return out[: self.local_cols].cpu().numpy().T.copy()

    def close(self):
        if self.plan:
            _lib.lib().cap_dist_plan_destroy(self.plan)
            self.plan = None


def setup(n, nb=0, comm=None):
    """Build the communicator (RCCL unless given) and the plan; fill the reference's SPD test matrix."""
    comm = comm or RcclComm()
    ctx = Context(n, nb or 512, comm)
    ctx.fill_symmetric(True)
    return ctx
