"""Multi-GPU blocked Cholesky driver (one process per GPU): 1 x P block-column-cyclic layout, RCCL over xGMI.

    ctx = dist_cholesky.setup(n, nb=512)      # uses torch.distributed's rank/size; builds the RCCL comm bundle
    ctx.fill_symmetric()                      # upstream's distribute_symmetric, generated on each GPU
    ctx.factor(); ctx.last_info()
    R_local = ctx.local_R()                   # n x local_cols device view

The factorization schedule itself lives behind the C ABI (csrc/dist.hip).  The tests swap RCCL for gloo-over-host-memory
callbacks (tests/host_staged.py, built on cap_comm_create_callbacks) so that the SAME schedule can be exercised by several
ranks sharing one GPU; that communicator is test infrastructure and lives under tests/."""
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


# ------------------------------------------------------------------ driver
class Context:
    def __init__(self, n, nb, comm, grid_rows=1):
        L = _lib.lib()
        self.n, self.nb, self.comm = int(n), int(nb), comm
        self.rank, self.size = comm.rank, comm.size
        self.grid_rows = int(grid_rows)
        if self.grid_rows != 1:
            raise _lib.CapitalError("this driver handles the 1 x P layout; the Pr x Pc layout is driven by dist_cholesky.Context2D (setup(grid_rows=...))")
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

    def get_option(self, key):
        return int(_lib.lib().cap_dist_get_option(self.plan, key.encode()))

    def factor(self):
        _lib.check(_lib.lib().cap_dist_factor(self.plan, self.A.data_ptr(), self.n, cur_stream()), "cap_dist_factor")

    def last_info(self):
        v = C.c_int64(0)
        _lib.check_info(_lib.lib().cap_dist_info(self.plan, cur_stream(), C.byref(v)), "cap_dist_info")
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

    def launch_counts(self):
        """Kernels / collectives this rank issued in the last factor call (the same four counters as Context2D)."""
        return {"mfma_kernels": self.get_option("count_gemm"), "chains": self.get_option("count_chain"),
                "copies": self.get_option("count_copy"), "collectives": self.get_option("count_coll")}

    def local_Rinv(self):
        """numpy (n x local_cols) copy of this rank's columns of R^-1 (options complete_inv = 0 / 1), construct_Rinv."""
        out = torch.zeros(max(self.local_cols, 1), self.n, dtype=torch.float64, device=self.device)
        _lib.check(_lib.lib().cap_dist_get_Rinv(self.plan, out.data_ptr(), self.n, cur_stream()), "cap_dist_get_Rinv")
        torch.cuda.synchronize()
        return out[: self.local_cols].cpu().numpy().T.copy()

    def probe(self, allreduce=None):
        """||(R^T R - A) X||_F / ||A X||_F for 8 random vectors with torch's fp64 matmul over this rank's columns, summed over
        the ranks by `allreduce` (validate.cholesky.probe): none of the library's kernels takes part in the check."""
        from . import validate
        Rl = self.local_R_device()
        gcols = torch.from_numpy(global_cols_of_rank(self.n, self.nb, self.size, self.rank)).to(self.device)
        return validate.cholesky.probe(self.A[: self.local_cols, : self.n].t(), Rl[: self.local_cols, : self.n].t(), gcols, allreduce=allreduce)

    def roofline(self, L, Cc, peak_tf=78.6):
        """One more factor call in profile mode: this rank's share of the trailing update (HIP events on its launch stream)
        plus the busy milliseconds of every stream role."""
        self.set_option("profile", 1)
        self.factor(); torch.cuda.synchronize()
        nl, ms, fl = Cc.c_int64(0), Cc.c_double(0), Cc.c_double(0)
        _lib.check(L.cap_dist_profile(self.plan, Cc.byref(nl), Cc.byref(ms), Cc.byref(fl)))
        busy = (Cc.c_double * 6)()
        _lib.check(L.cap_dist_profile_streams(self.plan, busy))
        self.set_option("profile", 0)
        if not nl.value:
            return None
        ach = fl.value / (ms.value * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": "dgemm_tn_dma_kernel<1> (rank %d's share of the trailing update)" % self.rank,
                "achieved": ach, "peak": peak_tf, "unit": "TFLOP/s", "frac": ach / peak_tf, "launches": nl.value,
                "avg_launch_ms": ms.value / nl.value, "algorithmic_flops_per_launch_avg": fl.value / nl.value, "traffic": None,
                "busy_ms_of_step": {"panel_chain": busy[0], "panel_solve": busy[1], "panel_head": busy[2], "msg_bcast": busy[3],
                                    "strip_exchange": busy[4], "main_bulk": busy[5]}}

    def close(self):
        if self.plan:
            _lib.lib().cap_dist_plan_destroy(self.plan)
            self.plan = None


# ------------------------------------------------------------------ Pr x Pc block-cyclic (csrc/dist2d.hip)
def global_index_2d(n, nb, Q, q):
    """Global row (or column) indices stored on process coordinate q of Q, in local storage order."""
    nblk = (n + nb - 1) // nb
    idx = []
    for lb in range(num_local_blocks(nblk, Q, q)):
        I = lb * Q + q
        idx.extend(range(I * nb, min(n, (I + 1) * nb)))
    return np.asarray(idx, dtype=np.int64)


class Context2D:
    """Driver of the Pr x Pc block-cyclic plan: rank = pr * Pc + pc; `row` / `col` are optional communicator objects of my
    process row / column (tests: host-staged groups), otherwise the library splits them off the world communicator."""

    def __init__(self, n, nb, comm, grid_rows, row=None, col=None):
        L = _lib.lib()
        self.n, self.nb, self.comm = int(n), int(nb), comm
        self.rank, self.size = comm.rank, comm.size
        self.grid_rows = self.Pr = int(grid_rows)
        self.Pc = self.size // self.Pr
        self.pr, self.pc = self.rank // self.Pc, self.rank % self.Pc
        self._row, self._col = row, col
        h = C.c_void_p()
        _lib.check(L.cap_dist2d_plan_create(C.byref(h), self.n, self.nb, comm.handle, self.Pr, row.handle if row else None,
                                            col.handle if col else None), "cap_dist2d_plan_create")
        self.plan = h
        self.local_rows, self.local_cols = int(L.cap_dist2d_get(h, 0)), int(L.cap_dist2d_get(h, 1))
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.lda = max(self.local_rows, 1)
        self.A = torch.zeros(max(self.local_cols, 1), self.lda, dtype=torch.float64, device=self.device)   # (cols, ld)

    def fill_symmetric(self, diagonally_dominant=True):
        _lib.check(_lib.lib().cap_fill_symmetric_bc2d(self.A.data_ptr(), self.lda, self.n, self.nb, self.Pr, self.Pc, self.pr, self.pc,
                                                      1 if diagonally_dominant else 0, cur_stream()), "cap_fill_symmetric_bc2d")

    def set_local(self, a):
        """a: numpy (local_rows x local_cols)"""
        if self.local_rows and self.local_cols:
            self.A[: self.local_cols, : self.local_rows].copy_(torch.from_numpy(np.ascontiguousarray(a.T)).to(self.device))

    def set_option(self, key, value):
        _lib.check(_lib.lib().cap_dist2d_set_option(self.plan, key.encode(), int(value)), "cap_dist2d_set_option")

    def factor(self):
        _lib.check(_lib.lib().cap_dist2d_factor(self.plan, self.A.data_ptr(), self.lda, cur_stream()), "cap_dist2d_factor")

    def last_info(self):
        v = C.c_int64(0)
        _lib.check_info(_lib.lib().cap_dist2d_info(self.plan, cur_stream(), C.byref(v)), "cap_dist2d_info")
        return v.value

    def local_R_device(self):
        out = torch.zeros(max(self.local_cols, 1), self.lda, dtype=torch.float64, device=self.device)
        _lib.check(_lib.lib().cap_dist2d_get_R(self.plan, out.data_ptr(), self.lda, cur_stream()), "cap_dist2d_get_R")
        return out

    def local_R(self):
        """numpy (local_rows x local_cols): my piece of R, zero below the global diagonal."""
        out = self.local_R_device()
        torch.cuda.synchronize()
        return out[: self.local_cols, : self.local_rows].cpu().numpy().T.copy()

    def local_Rinv(self):
        """numpy (local_rows x local_cols): my piece of R^-1 (options complete_inv = 0 / 1), zero below the global diagonal."""
        out = torch.zeros(max(self.local_cols, 1), self.lda, dtype=torch.float64, device=self.device)
        _lib.check(_lib.lib().cap_dist2d_get_Rinv(self.plan, out.data_ptr(), self.lda, cur_stream()), "cap_dist2d_get_Rinv")
        torch.cuda.synchronize()
        return out[: self.local_cols, : self.local_rows].cpu().numpy().T.copy()

    def launch_counts(self):
        L = _lib.lib()
        return {"mfma_kernels": int(L.cap_dist2d_get(self.plan, 8)), "chains": int(L.cap_dist2d_get(self.plan, 9)),
                "copies": int(L.cap_dist2d_get(self.plan, 10)), "collectives": int(L.cap_dist2d_get(self.plan, 11))}

    def probe(self, allreduce=None):
        """||(R^T R - A) X||_F / ||A X||_F over the distributed pieces with torch fp64 matmul (no library kernel):
        R X needs the row-block sums over process columns, R^T (R X) the sums over process rows - both by `allreduce` of
        zero-padded global-length vectors (cheap: 8 columns)."""
        n, dev = self.n, self.device
        rows = torch.from_numpy(global_index_2d(n, self.nb, self.Pr, self.pr)).to(dev)
        cols = torch.from_numpy(global_index_2d(n, self.nb, self.Pc, self.pc)).to(dev)
        g = torch.Generator(device="cpu"); g.manual_seed(11)
        X = torch.rand(n, 8, dtype=torch.float64, generator=g).to(dev)
        red = allreduce if allreduce is not None else (lambda t: t)
        Rl = self.local_R_device()[: self.local_cols, : self.local_rows].t()        # local_rows x local_cols
        Al = self.A[: self.local_cols, : self.local_rows].t()
        # A is symmetric and fully stored: (A X)[rows] = sum over process columns of A_loc X[cols]
        ax = torch.zeros(n, 8, dtype=torch.float64, device=dev); rx = torch.zeros_like(ax)
        if rows.numel() and cols.numel():
            ax[rows] += Al @ X[cols]; rx[rows] += Rl @ X[cols]
        ax = red(ax); rx = red(rx)                  # every entry summed over all ranks: each (row, col) block counted once
        rtrx = torch.zeros(n, 8, dtype=torch.float64, device=dev)
        if rows.numel() and cols.numel():
            rtrx[cols] += Rl.t() @ rx[rows]
        rtrx = red(rtrx)
        return float((rtrx - ax).norm() / ax.norm())

    def roofline(self, L, Cc, peak_tf=78.6):
        return None

    def close(self):
        if self.plan:
            _lib.lib().cap_dist2d_plan_destroy(self.plan)
            self.plan = None


def setup(n, nb=0, comm=None, grid_rows=1, row=None, col=None):
    """Build the communicator (RCCL unless given) and the plan; fill the reference's SPD test matrix."""
    comm = comm or RcclComm()
    if grid_rows and grid_rows != 1:
        ctx = Context2D(n, nb or 512, comm, grid_rows, row, col)
    else:
        ctx = Context(n, nb or 512, comm, 1)
    ctx.fill_symmetric(True)
    return ctx
