"""Mixed-precision Cholesky solve (BASELINE config 5): bf16 MFMA factorization + fp64 iterative refinement.

    plan = mixed.plan(n, nrhs_max)
    plan.factor(A)                       # A: `matrix` (n x n fp64, full symmetric, resident in HBM)
    X, iters, relres = plan.solve(A, B)  # B: `matrix` (n x nrhs); X to fp64 accuracy

Not in the reference (its solve path is a stub, trsm/diaginvert/diaginvert.hpp:7-10); the fp64 path of this package
(cholinv + blas::engine::_trsm) is its oracle.  Everything runs behind the C ABI (csrc/mixed.hip)."""
import ctypes as C

from . import _lib
from ._util import cur_stream
from .matrix import matrix


class plan:
    def __init__(self, n, nrhs_max=128):
        self.n, self.nrhs_max = int(n), int(nrhs_max)
        h = C.c_void_p()
        _lib.check(_lib.lib().cap_mpchol_plan_create(C.byref(h), self.n, self.nrhs_max), "cap_mpchol_plan_create")
        self._h = h

    def factor(self, A):
        _lib.check(_lib.lib().cap_mpchol_factor(self._h, A.data_ptr(), A.ld(), cur_stream()), "mpchol::factor")

    def set_option(self, key, value):
        """"strip": panels per bf16 trailing update (1 | 2, default 2: K = 2048), "profile"."""
        _lib.check(_lib.lib().cap_mpchol_set_option(self._h, key.encode(), int(value)), "mpchol::set_option")

    def last_info(self):
        v = C.c_int64(0)
        _lib.check_info(_lib.lib().cap_mpchol_info(self._h, cur_stream(), C.byref(v)), "cap_mpchol_info")
        return v.value

    def solve(self, A, B, max_iter=30, tol=1e-15):
        nrhs = B.num_columns_global()
        X = matrix(nrhs, self.n, 1, 1)
        it, rr = C.c_int(0), C.c_double(0)
        _lib.check(_lib.lib().cap_mpchol_solve(self._h, A.data_ptr(), A.ld(), B.data_ptr(), B.ld(), X.data_ptr(), X.ld(), nrhs, int(max_iter),
                                               float(tol), C.byref(it), C.byref(rr), cur_stream()), "mpchol::solve")
        return X, it.value, rr.value

    def profile_update(self, A):
        """One more factor call with HIP events around every bf16 trailing update (bf16_tn_kernel) on its launch stream:
        (launches, summed ms, summed algorithmic flops, summed algorithmic bytes)."""
        import torch
        L = _lib.lib()
        _lib.check(L.cap_mpchol_set_option(self._h, b"profile", 1), "mpchol::set_option")
        self.factor(A)
        torch.cuda.synchronize()
        nl, ms, fl, by = C.c_int64(0), C.c_double(0), C.c_double(0), C.c_double(0)
        _lib.check(L.cap_mpchol_profile(self._h, C.byref(nl), C.byref(ms), C.byref(fl), C.byref(by)), "mpchol::profile")
        _lib.check(L.cap_mpchol_set_option(self._h, b"profile", 0), "mpchol::set_option")
        return nl.value, ms.value, fl.value, by.value

    def R32(self):
        """[row, col] torch view of the fp32 factor (upper)."""
        import torch
        ld = C.c_int64(0)
        p = _lib.lib().cap_mpchol_R32_ptr(self._h, C.byref(ld))
        out = torch.empty(self.n, self.n, dtype=torch.float32, device="cuda")
        rt = C.CDLL("libamdhip64.so")
        rt.hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(p), C.c_size_t(self.n * self.n * 4), C.c_int(3))
        return out.t()

    def close(self):
        if self._h:
            _lib.lib().cap_mpchol_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class dist_plan:
    """Mixed-precision solve on P GPUs (csrc/dist_mixed.hip): 1 x P block-column-cyclic bf16 factorization + distributed fp64
    refinement.  `comm`: a communicator object with .handle / .rank / .size (dist_cholesky.RcclComm).  The local operand is
    this rank's block columns of the FULL symmetric matrix as a (local_cols, n) torch tensor (column-major n x local_cols)."""

    def __init__(self, n, comm, nb=0, nrhs_max=128):
        self.n, self.comm, self.nrhs_max = int(n), comm, int(nrhs_max)
        h = C.c_void_p()
        _lib.check(_lib.lib().cap_dmp_plan_create(C.byref(h), self.n, int(nb), self.nrhs_max, comm.handle), "cap_dmp_plan_create")
        self._h = h
        self.local_cols = int(_lib.lib().cap_dmp_local_cols(h))

    def factor(self, A_local):
        _lib.check(_lib.lib().cap_dmp_factor(self._h, A_local.data_ptr(), self.n, cur_stream()), "dmp::factor")

    def last_info(self):
        v = C.c_int64(0)
        _lib.check_info(_lib.lib().cap_dmp_info(self._h, cur_stream(), C.byref(v)), "cap_dmp_info")
        return v.value

    def solve(self, A_local, B, max_iter=30, tol=1e-15):
        """B: `matrix` (n x nrhs), the same on every rank; returns (X matrix, sweeps, relres) - X identical on every rank."""
        nrhs = B.num_columns_global()
        X = matrix(nrhs, self.n, 1, 1)
        it, rr = C.c_int(0), C.c_double(0)
        _lib.check(_lib.lib().cap_dmp_solve(self._h, A_local.data_ptr(), self.n, B.data_ptr(), B.ld(), X.data_ptr(), X.ld(), nrhs, int(max_iter),
                                            float(tol), C.byref(it), C.byref(rr), cur_stream()), "dmp::solve")
        return X, it.value, rr.value

    def R32_local(self):
        """numpy (n x local_cols) copy of this rank's block columns of the fp32 factor."""
        import torch
        ld = C.c_int64(0)
        p = _lib.lib().cap_dmp_R32_ptr(self._h, C.byref(ld))
        out = torch.empty(max(self.local_cols, 1), ld.value, dtype=torch.float32, device="cuda")
        rt = C.CDLL("libamdhip64.so")
        torch.cuda.synchronize()
        rt.hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(p), C.c_size_t(max(self.local_cols, 1) * ld.value * 4), C.c_int(3))
        return out[: self.local_cols, : self.n].cpu().numpy().T.copy()

    def close(self):
        if self._h:
            _lib.lib().cap_dmp_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
