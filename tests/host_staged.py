"""TEST INFRASTRUCTURE: a cap_comm whose collectives are gloo calls on host copies, so that the multi-rank schedules
(csrc/dist.hip, csrc/summa.hip, csrc/cacqr.hip) can be exercised by several processes sharing ONE GPU.

Built on the library's callback constructor (cap_comm_create_callbacks, include/capital_amd.h).  Not part of the product:
the product path is RCCL (capital_amd.dist_cholesky.RcclComm).  Slow by construction."""
import ctypes as C

import torch

from capital_amd import _lib


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


class HostStagedComm:
    """cap_comm whose collectives are gloo calls on host copies (several ranks may share one GPU).

    Each callback waits for the stream it is handed - and nothing else - before it reads the device buffer, exactly
    the ordering an RCCL kernel enqueued on that stream would have; the other streams of the schedule keep running, so
    a missing event edge between them shows up as a wrong result (tests add random per-stream delays on top).
    group: a torch.distributed process group (sub-communicators of a grid bundle); ranks are group-local."""
    _AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
    _BC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)
    _AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
    _A2A = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64),
                       C.POINTER(C.c_int64), C.c_void_p)

    def __init__(self, group=None):
        dist = _dist()
        if dist is None or dist.get_backend() != "gloo":
            raise _lib.CapitalError("HostStagedComm needs torch.distributed initialised with the gloo backend")
        self.group = group
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)
        self.calls = {"allgather": 0, "bcast": 0, "allreduce": 0, "alltoallv": 0}

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

        def a2a(ctx, send, sc, sd, recv, rc, rd, stream):
            # personalised all-to-all as gloo point-to-point pairs (the self piece was already copied by the library)
            try:
                sync(stream)
                g = (lambda r: dist.get_global_rank(self.group, r)) if self.group is not None else (lambda r: r)
                reqs, rbufs = [], []
                for r in range(self.size):
                    if r == self.rank:
                        continue
                    if rc[r] > 0:
                        t = torch.empty(rc[r], dtype=torch.float64)
                        rbufs.append((r, t))
                        reqs.append(dist.irecv(t, src=g(r), group=self.group))
                for r in range(self.size):
                    if r == self.rank or sc[r] <= 0:
                        continue
                    t = _DevView(int(send) + 8 * sd[r], sc[r]).to_host()
                    reqs.append(dist.isend(t, dst=g(r), group=self.group))
                for q in reqs:
                    q.wait()
                for r, t in rbufs:
                    _DevView(int(recv) + 8 * rd[r], rc[r]).from_host(t)
                self.calls["alltoallv"] += 1
                return 0
            except Exception as e:  # pragma: no cover
                print("HostStagedComm alltoallv failed:", e, flush=True)
                return 1

        self._cbs = (self._AG(ag), self._BC(bc), self._AR(ar), self._A2A(a2a))   # keep alive
        h = C.c_void_p()
        _lib.check(_lib.lib().cap_comm_create_callbacks(C.byref(h), self.rank, self.size,
                                                        C.cast(self._cbs[0], C.c_void_p), C.cast(self._cbs[1], C.c_void_p),
                                                        C.cast(self._cbs[2], C.c_void_p), None), "cap_comm_create_callbacks")
        _lib.check(_lib.lib().cap_comm_set_alltoallv_callback(h, C.cast(self._cbs[3], C.c_void_p)), "cap_comm_set_alltoallv_callback")
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


def grid_groups(Pr, factory=None):
    """Row / column communicators of the Pr x Pc process grid (rank = pr * Pc + pc) over the gloo world: every rank creates
    every group in the same order (torch.distributed.new_group is collective) and keeps its own two."""
    dist = _dist()
    factory = factory or HostStagedComm
    rank, size = dist.get_rank(), dist.get_world_size()
    Pc = size // Pr
    pr, pc = rank // Pc, rank % Pc
    row = col = None
    for r in range(Pr):
        g = dist.new_group(ranks=[r * Pc + c for c in range(Pc)], backend="gloo")
        if r == pr:
            row = factory(group=g)
    for c in range(Pc):
        g = dist.new_group(ranks=[r * Pc + c for r in range(Pr)], backend="gloo")
        if c == pc:
            col = factory(group=g)
    return row, col
