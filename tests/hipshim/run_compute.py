"""TEST INFRASTRUCTURE: the library's host side computing REAL factors on a machine without a GPU.

The product's object files linked against the recording stand-in (as in run_scenarios.py) with the stand-in in COMPUTE MODE: every
launch runs a CPU model of its kernel at enqueue time (kernels_cpu.cpp - the kernels' contract incl. launch geometry: the GEMM models
walk the launch's blocks through the library's own tile enumeration), every copy / memset is carried out, fresh "device" memory is
filled with NaN patterns.  What comes out of a plan is compared with NumPy / the CPU oracle:

    python tests/hipshim/run_compute.py out.json

One entry per case: {"name", "errors": {...}, "findings": [...]}.  A finding is a result off by more than the tolerance, a NaN that
survived (memory the schedule read before anything wrote it), a kernel without a model, an exception, or anything the structural
replay of the same trace reports.  Its own process (no torch)."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import run_scenarios as rs          # noqa: E402  (loads the libraries, installs the access hook)
from oracle import capital_oracle as orc   # noqa: E402  (the checker)

L, shim = rs.L, rs.shim
shim.shim_set_compute.argtypes = [C.c_int]
shim.shim_unmodelled.restype = C.c_longlong
RESULTS = []
TOL = 2e-11


def view(ptr, rows, cols, ld=None):
    """(rows x cols) NumPy view of column-major "device" memory at ptr (leading dimension ld)"""
    ld = ld or rows
    a = np.ctypeslib.as_array((C.c_double * (ld * cols)).from_address(ptr.value if hasattr(ptr, "value") else ptr))
    return a.reshape((cols, ld)).T[:rows]


def spd(n, seed=0, cond=10.0):
    """a well-conditioned SPD matrix that is NOT diagonally dominant: every Schur update matters"""
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((n, n))
    a = g @ g.T / n + (4.0 / cond) * np.eye(n)
    return 0.5 * (a + a.T)


def rel(x, ref):
    d = np.linalg.norm(ref)
    return float(np.linalg.norm(x - ref) / (d if d > 0 else 1.0))


def case(name, user_stream=0):
    def deco(fn):
        if rs.FILTER and rs.FILTER not in name:
            return fn
        r = rs.Run(name, user_stream)
        before = int(shim.shim_unmodelled())
        errors, findings = {}, []
        try:
            fn(r, errors)
        except Exception as e:
            findings.append("exception: %r" % (e,))
        out = r.finish()
        findings += out["findings"]
        if int(shim.shim_unmodelled()) > before:
            findings += sorted({l for l in r.lines if l.startswith("UNMODELLED")})
        for k, v in errors.items():
            if not (v == v) or v > TOL:
                findings.append("%s off by %.3e (tolerance %.1e)" % (k, v, TOL))
        RESULTS.append({"name": name, "errors": errors, "findings": findings, "stats": {k: v for k, v in out["stats"].items() if not isinstance(v, dict)}})
        return fn
    return deco


def cholinv_compute(r, errors, n, ci, split, bc, opts=(), seed=0, reps=2):
    a = spd(n, seed)
    plan = C.c_void_p()
    rs.ok(L.cap_cholinv_plan_create(C.byref(plan), n, ci, split, bc, b"U", None), "cap_cholinv_plan_create")
    for k, v in opts:
        rs.ok(L.cap_cholinv_set_option(plan, k.encode(), v), "set_option " + k)
    lda = n + 2
    A = rs.dmalloc(8 * lda * n); out = rs.dmalloc(8 * n * n)
    av = view(A, n, n, lda)
    av[:, :] = np.triu(a) + np.tril(np.full((n, n), np.nan), -1)        # only the upper triangle may be consumed (cholinv.hpp:13)
    info = C.c_int64(-1)
    for _ in range(reps):                                               # the second call starts from the first one's leftovers
        rs.ok(r.call("cholinv_factor", L.cap_cholinv_factor, plan, A, lda, r.stream), "cap_cholinv_factor")
    r.call("cholinv_info", L.cap_cholinv_info, plan, r.stream, C.byref(info))
    errors["info"] = float(abs(info.value))
    rref, riref = orc.cholinv(a, max(ci, 0), split, bc)
    rs.ok(r.call("cholinv_get_R", L.cap_cholinv_get_R, plan, out, n, r.stream), "cap_cholinv_get_R")
    errors["R"] = rel(view(out, n, n), np.linalg.cholesky(a).T)
    if ci >= 0:
        rs.ok(r.call("cholinv_get_Rinv", L.cap_cholinv_get_Rinv, plan, out, n, r.stream), "cap_cholinv_get_Rinv")
        ri = view(out, n, n).copy()
        errors["Rinv"] = rel(ri, riref)
        errors["Rinv_pattern"] = float(np.count_nonzero((ri != 0) != (riref != 0)))
    rs.ok(L.cap_cholinv_plan_destroy(plan), "cap_cholinv_plan_destroy")
    shim.hipFree(A); shim.hipFree(out)


def operators_compute(r, errors, m, n, k, seed=1):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((m, k)); b = rng.standard_normal((k, n)); c0 = rng.standard_normal((m, n))
    A = rs.dmalloc(8 * m * k); B = rs.dmalloc(8 * k * n); Cc = rs.dmalloc(8 * m * n)
    view(A, m, k)[:] = a; view(B, k, n)[:] = b; view(Cc, m, n)[:] = c0
    rs.ok(r.call("dgemm NN", L.cap_dgemm, 0, 0, m, n, k, 1.5, A, m, B, k, -0.5, Cc, m, r.stream), "cap_dgemm")
    errors["dgemm NN"] = rel(view(Cc, m, n), 1.5 * a @ b - 0.5 * c0)
    At = rs.dmalloc(8 * k * m); view(At, k, m)[:] = a.T
    c1 = view(Cc, m, n).copy()
    rs.ok(r.call("dgemm TN", L.cap_dgemm, 1, 0, m, n, k, -1.0, At, k, B, k, 1.0, Cc, m, r.stream), "cap_dgemm")
    errors["dgemm TN beta=1"] = rel(view(Cc, m, n), c1 - a @ b)
    Bt = rs.dmalloc(8 * n * k); view(Bt, n, k)[:] = b.T
    rs.ok(r.call("dgemm NT", L.cap_dgemm, 0, 1, m, n, k, 1.0, A, m, Bt, n, 0.0, Cc, m, r.stream), "cap_dgemm")
    errors["dgemm NT"] = rel(view(Cc, m, n), a @ b)
    rs.ok(r.call("dgemm TT", L.cap_dgemm, 1, 1, m, n, k, 2.0, At, k, Bt, n, 0.0, Cc, m, r.stream), "cap_dgemm")
    errors["dgemm TT"] = rel(view(Cc, m, n), 2.0 * a @ b)
    # SYRK upper, trans: C = alpha A^T A + beta C on the upper triangle, the lower one untouched
    S = rs.dmalloc(8 * m * m); s0 = rng.standard_normal((m, m)); view(S, m, m)[:] = s0
    rs.ok(r.call("dsyrk", L.cap_dsyrk, 1, 1, m, k, -1.0, At, k, 1.0, S, m, r.stream), "cap_dsyrk")
    errors["dsyrk"] = rel(view(S, m, m), np.triu(s0 - a @ a.T) + np.tril(s0, -1))
    # the triangular operators on an m x m upper triangle with garbage below it
    t = np.linalg.cholesky(spd(m, seed + 3)).T            # (a RANDOM triangle has a condition number of 1e17 at m = 1024)
    T = rs.dmalloc(8 * m * m); view(T, m, m)[:] = t + np.tril(np.full((m, m), np.nan), -1)
    X = rs.dmalloc(8 * m * n); x0 = rng.standard_normal((m, n))
    for (side, trans, nm) in [(0, 0, "L N"), (0, 1, "L T")]:
        view(X, m, n)[:] = x0
        W = rs.dmalloc(8 * max(int(L.cap_dtrmm_work_size(side, m, n)), 1))
        rs.ok(r.call("dtrmm " + nm, L.cap_dtrmm, side, 1, trans, 0, m, n, 0.75, T, m, X, m, W, r.stream), "cap_dtrmm")
        errors["dtrmm " + nm] = rel(view(X, m, n), 0.75 * (t.T if trans else t) @ x0)
        view(X, m, n)[:] = x0
        W2 = rs.dmalloc(8 * max(int(L.cap_dtrsm_work_size(side, m, n)), 1))
        rs.ok(r.call("dtrsm " + nm, L.cap_dtrsm, side, 1, trans, m, n, 1.25, T, m, X, m, W2, r.stream), "cap_dtrsm")
        errors["dtrsm " + nm] = rel(view(X, m, n), np.linalg.solve(t.T if trans else t, 1.25 * x0))
        shim.hipFree(W); shim.hipFree(W2)
    Y = rs.dmalloc(8 * n * m); y0 = rng.standard_normal((n, m))
    for (trans, nm) in [(0, "R N"), (1, "R T")]:
        view(Y, n, m)[:] = y0
        W = rs.dmalloc(8 * max(int(L.cap_dtrmm_work_size(1, n, m)), 1))
        rs.ok(r.call("dtrmm " + nm, L.cap_dtrmm, 1, 1, trans, 0, n, m, 1.0, T, m, Y, n, W, r.stream), "cap_dtrmm")
        errors["dtrmm " + nm] = rel(view(Y, n, m), y0 @ (t.T if trans else t))
        view(Y, n, m)[:] = y0
        W2 = rs.dmalloc(8 * max(int(L.cap_dtrsm_work_size(1, n, m)), 1))
        rs.ok(r.call("dtrsm " + nm, L.cap_dtrsm, 1, 1, trans, n, m, 1.0, T, m, Y, n, W2, r.stream), "cap_dtrsm")
        errors["dtrsm " + nm] = rel(view(Y, n, m), np.linalg.solve((t.T if trans else t).T, y0.T).T)
        shim.hipFree(W); shim.hipFree(W2)
    # LAPACK-shaped: potrf in place on caller memory (lower part untouched), trtri in place
    a2 = spd(m, seed + 7)
    P = rs.dmalloc(8 * m * m); view(P, m, m)[:] = np.triu(a2) + np.tril(np.full((m, m), 7.0), -1)
    W3 = rs.dmalloc(8 * max(int(L.cap_dpotrf_work_size(m)), 1)); info = rs.dmalloc(8)
    rs.ok(r.call("dpotrf", L.cap_dpotrf, 1, m, P, m, info, W3, r.stream), "cap_dpotrf")
    rr = np.linalg.cholesky(a2).T
    errors["dpotrf"] = rel(view(P, m, m), rr + np.tril(np.full((m, m), 7.0), -1))
    W4 = rs.dmalloc(8 * max(int(L.cap_dtrtri_work_size(m)), 1))
    rs.ok(r.call("dtrtri", L.cap_dtrtri, 1, m, P, m, W4, r.stream), "cap_dtrtri")
    errors["dtrtri"] = rel(view(P, m, m), np.linalg.inv(rr) + np.tril(np.full((m, m), 7.0), -1))
    for q in (A, B, Cc, At, Bt, S, T, X, Y, P, W3, W4, info):
        shim.hipFree(q)


# ---------------------------------------------------------------------------------------------------------------------------------
# Multi-rank plans: one THREAD per rank inside this process (ctypes releases the GIL inside the library), communicators whose
# collectives really move the data between the ranks' "device" memory and block until their partners have arrived - the way the
# host-staged communicator of the GPU emulations does.  Every rank executes its launches at enqueue time, so by the time it enters a
# collective everything it enqueued before has happened; the IPC exchanges copy straight into the peer's buffer (one address space: a
# handle opens as the peer's own pointer), ordered by the token all-reduces exactly as on the GPU.
import queue       # noqa: E402
import threading   # noqa: E402


class Group:
    def __init__(self, size):
        self.size = size
        self.bar = threading.Barrier(size, timeout=120)
        self.slot = [None] * size
        self.tmp = None
        self.mail = {}
        self.lock = threading.Lock()

    def box(self, key):
        with self.lock:
            return self.mail.setdefault(key, queue.Queue())


GROUPS, GLOCK = {}, threading.Lock()


def group(label, size):
    with GLOCK:
        return GROUPS.setdefault(label, Group(size))


def dview(ptr, n):
    return np.ctypeslib.as_array((C.c_double * n).from_address(ptr))


class TComm:
    """communicator of one rank thread: `me` of `size` in the group named `label` (unique per configuration and group)"""

    def __init__(self, me, size, label):
        g = group(label, size)
        self.rank, self.size = me, size

        def ag_(ctx, s, r, n, st):
            g.slot[me] = s
            g.bar.wait()
            for q in range(size):
                if n > 0:
                    C.memmove(r + q * n * 8, g.slot[q], n * 8)
            g.bar.wait()
            return 0

        def bc_(ctx, b, n, root, st):
            g.slot[me] = b
            g.bar.wait()
            if me != root and n > 0:
                C.memmove(b, g.slot[root], n * 8)
            g.bar.wait()
            return 0

        def ar_(ctx, b, n, st):
            g.slot[me] = b
            g.bar.wait()
            tot = dview(g.slot[0], n).copy()
            for q in range(1, size):
                tot += dview(g.slot[q], n)
            g.bar.wait()
            dview(b, n)[:] = tot
            g.bar.wait()
            return 0

        def a2a_(ctx, s, sc, sd, r, rc, rd, st):
            for q in range(size):
                if sc[q] > 0:
                    g.box(("data", me, q)).put(s + 8 * sd[q])
            for q in range(size):
                if rc[q] > 0:
                    src = g.box(("data", q, me)).get(timeout=120)
                    C.memmove(r + 8 * rd[q], src, 8 * rc[q])
                    g.box(("ack", q, me)).put(1)
            for q in range(size):
                if sc[q] > 0:
                    g.box(("ack", me, q)).get(timeout=120)
            return 0
        self._cb = (rs._AG(ag_), rs._BC(bc_), rs._AR(ar_), rs._A2A(a2a_))
        rs._KEEP.extend(self._cb)
        self.handle = C.c_void_p()
        rs.ok(L.cap_comm_create_callbacks(C.byref(self.handle), me, size, self._cb[0], self._cb[1], self._cb[2], None), "cap_comm_create_callbacks")
        rs.ok(L.cap_comm_set_alltoallv_callback(self.handle, self._cb[3]), "cap_comm_set_alltoallv_callback")

    def close(self):
        L.cap_comm_destroy(self.handle)


def run_ranks(nranks, fn):
    """fn(rank) on one thread per rank -> list of per-rank results; an exception on any rank is re-raised"""
    out, err = [None] * nranks, []

    share = max(1, (os.cpu_count() or 8) // nranks)

    def work(q):
        try:
            shim.shim_set_threads(share)       # the kernel models of this rank: its share of the host cores
            out[q] = fn(q)
        except Exception as e:          # a broken barrier on the other ranks follows from the first failure
            err.append((q, repr(e)))
    th = [threading.Thread(target=work, args=(q,)) for q in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if err:
        raise RuntimeError("rank failures: %s" % err[:3])
    return out


def bc_indices(n, nb, Q, q):
    """global indices of the rows / columns process coordinate q of Q owns in the block-cyclic layout"""
    idx = [j for J in range(q, (n + nb - 1) // nb, Q) for j in range(J * nb, min(n, (J + 1) * nb))]
    return np.array(idx, dtype=np.int64)


def ref_factors(a, ci, split):
    """R, and R^-1 with upstream's pattern (root block empty for complete_inv = 0 unless the root is a base case)"""
    r, ri = orc.cholinv(a, max(ci, 0), split, 0)
    return np.linalg.cholesky(a).T, ri


def dist_compute(r_unused, errors, n, nb, P, opts=(), ci=-1, seed=0, uid=[0]):
    uid[0] += 1
    a = spd(n, seed)
    rref = np.linalg.cholesky(a).T
    riref = np.linalg.inv(rref)

    def rank(p):
        comm = TComm(p, P, "dist%d" % uid[0])
        plan = C.c_void_p()
        rs.ok(L.cap_dist_plan_create(C.byref(plan), n, nb, comm.handle), "cap_dist_plan_create")
        for k, v in opts:
            rs.ok(L.cap_dist_set_option(plan, k.encode(), v), "dist set_option " + k)
        if ci >= 0:
            rs.ok(L.cap_dist_set_option(plan, b"complete_inv", ci), "dist complete_inv")
        cols = bc_indices(n, nb, P, p)
        lc = len(cols)
        assert lc == int(L.cap_dist_local_cols(plan))
        A = rs.dmalloc(8 * n * max(lc, 1)); out = rs.dmalloc(8 * n * max(lc, 1))
        if lc:
            view(A, n, lc)[:] = a[:, cols]
        info = C.c_int64(-1)
        for _ in range(2):
            rs.ok(L.cap_dist_factor(plan, A, n, None), "cap_dist_factor")
        L.cap_dist_info(plan, None, C.byref(info))
        if dict(opts).get("ipc") and P > 1 and int(L.cap_dist_get_option(plan, b"ipc_active")) != 1:
            raise RuntimeError("the IPC strip exchange is not active")
        rs.ok(L.cap_dist_get_R(plan, out, n, None), "cap_dist_get_R")
        R = view(out, n, lc).copy() if lc else np.zeros((n, 0))
        Ri = None
        if ci >= 0:
            rs.ok(L.cap_dist_get_Rinv(plan, out, n, None), "cap_dist_get_Rinv")
            Ri = view(out, n, lc).copy() if lc else np.zeros((n, 0))
        rs.ok(L.cap_dist_plan_destroy(plan), "cap_dist_plan_destroy")
        comm.close(); shim.hipFree(A); shim.hipFree(out)
        return cols, R, Ri, info.value
    res = run_ranks(P, rank)
    R = np.zeros((n, n)); Ri = np.zeros((n, n))
    for cols, Rp, Rip, info in res:
        R[:, cols] = Rp
        if Rip is not None:
            Ri[:, cols] = Rip
        errors["info"] = max(errors.get("info", 0.0), float(abs(info)))
    errors["R"] = rel(R, rref)
    if ci >= 0:
        n1 = n >> dict(opts).get("split", 1)
        want = riref.copy()
        if ci == 0 and 0 < n1 < n:
            want[:n1, n1:] = 0.0                 # cholinv.hpp:147: the root block of R^-1 stays empty
        errors["Rinv"] = rel(Ri, want)


def dist2d_compute(r_unused, errors, n, nb, Pr, Pc, opts=(), seed=0, uid=[0]):
    uid[0] += 1
    a = spd(n, seed)
    rref = np.linalg.cholesky(a).T
    riref = np.linalg.inv(rref)
    ci = dict(opts).get("complete_inv", -1)

    def rank(q):
        pr, pc = q // Pc, q % Pc
        world = TComm(q, Pr * Pc, "w%d" % uid[0]); row = TComm(pc, Pc, "r%d_%d" % (uid[0], pr)); col = TComm(pr, Pr, "c%d_%d" % (uid[0], pc))
        plan = C.c_void_p()
        rs.ok(L.cap_dist2d_plan_create(C.byref(plan), n, nb, world.handle, Pr, row.handle, col.handle), "cap_dist2d_plan_create")
        for k, v in opts:
            rs.ok(L.cap_dist2d_set_option(plan, k.encode(), v), "dist2d set_option " + k)
        rows, cols = bc_indices(n, nb, Pr, pr), bc_indices(n, nb, Pc, pc)
        lr, lc = len(rows), len(cols)
        assert (lr, lc) == (int(L.cap_dist2d_get(plan, 0)), int(L.cap_dist2d_get(plan, 1)))
        A = rs.dmalloc(8 * max(lr, 1) * max(lc, 1)); out = rs.dmalloc(8 * max(lr, 1) * max(lc, 1))
        if lr and lc:
            view(A, lr, lc)[:] = a[np.ix_(rows, cols)]
        info = C.c_int64(-1)
        for _ in range(2):
            rs.ok(L.cap_dist2d_factor(plan, A, max(lr, 1), None), "cap_dist2d_factor")
        L.cap_dist2d_info(plan, None, C.byref(info))
        if dict(opts).get("ipc") and Pr * Pc > 1 and int(L.cap_dist2d_get(plan, 12)) != 1:
            raise RuntimeError("the IPC operand moves are not active")
        rs.ok(L.cap_dist2d_get_R(plan, out, max(lr, 1), None), "cap_dist2d_get_R")
        R = view(out, lr, lc).copy() if lr and lc else np.zeros((lr, lc))
        Ri = None
        if ci >= 0:
            rs.ok(L.cap_dist2d_get_Rinv(plan, out, max(lr, 1), None), "cap_dist2d_get_Rinv")
            Ri = view(out, lr, lc).copy() if lr and lc else np.zeros((lr, lc))
        rs.ok(L.cap_dist2d_plan_destroy(plan), "cap_dist2d_plan_destroy")
        for c in (world, row, col):
            c.close()
        shim.hipFree(A); shim.hipFree(out)
        return rows, cols, R, Ri, info.value
    res = run_ranks(Pr * Pc, rank)
    R = np.zeros((n, n)); Ri = np.zeros((n, n))
    for rows, cols, Rp, Rip, info in res:
        if len(rows) and len(cols):
            R[np.ix_(rows, cols)] = Rp
            if Rip is not None:
                Ri[np.ix_(rows, cols)] = Rip
        errors["info"] = max(errors.get("info", 0.0), float(abs(info)))
    errors["R"] = rel(R, rref)
    if ci >= 0:
        n1 = n >> dict(opts).get("split", 1)
        want = riref.copy()
        if ci == 0 and 0 < n1 < n:
            want[:n1, n1:] = 0.0
        errors["Rinv"] = rel(Ri, want)


class TTopo:
    """topo::square (kind 0) / topo::rect (kind 1) bundle of one rank thread over TComm groups (the splits of run_scenarios.Topo)"""

    def __init__(self, tag, kind, rank, size, c, num_chunks=0):
        def co(q):
            dd, xx, yy, zz = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
            rs.ok(L.cap_topo_coords(kind, q, size, c, C.byref(dd), C.byref(xx), C.byref(yy), C.byref(zz)), "cap_topo_coords")
            return dd.value, xx.value, yy.value, zz.value
        self.d, self.x, self.y, self.z = co(rank)
        if kind == 0:
            splits = [(lambda q: (co(q)[2], co(q)[3]), lambda q: co(q)[1]), (lambda q: (co(q)[1], co(q)[3]), lambda q: co(q)[2]),
                      (lambda q: q // c, lambda q: q), (lambda q: co(q)[3], lambda q: q), None, None, None]
        else:
            cube, sl = c * c * c, c * c
            splits = [(lambda q: (q // cube, ((q % cube) % c) + c * ((q % cube) // sl)), lambda q: q % cube), None,
                      (lambda q: (q // cube, (q % cube) // c), lambda q: q % cube), (lambda q: q % c, lambda q: q),
                      (lambda q: (q % sl, (q // sl) // c), lambda q: q // sl), (lambda q: (q % sl, (q // sl) % c), lambda q: q // sl),
                      (lambda q: q // cube, lambda q: q)]
        self.world = TComm(rank, size, "%s:world" % tag)
        self.subs = []
        arr = (C.c_void_p * 7)()
        for i, sp in enumerate(splits):
            if sp is None:
                continue
            me, n, members = rs.group_of(sp[0], sp[1], size, rank)
            cm = TComm(me, n, "%s:sub%d:%s" % (tag, i, str(sp[0](rank)).replace(" ", "")))
            self.subs.append(cm); arr[i] = cm.handle
        self.handle = C.c_void_p()
        rs.ok(L.cap_topo_create_from(C.byref(self.handle), kind, self.world.handle, c, 0, num_chunks, arr, 7), "cap_topo_create_from")

    def close(self):
        L.cap_topo_destroy(self.handle)
        for cm in self.subs + [self.world]:
            cm.close()


def cyc_piece(a, x, y, d):
    """element-cyclic piece (x, y) of a d x d grid, ceil-sized and zero padded (matrix.hpp:8-11): piece[r, c] = a[y + r d, x + c d]"""
    m, n = a.shape
    out = np.zeros(((m + d - 1) // d, (n + d - 1) // d))
    sub = a[y::d, x::d]
    out[:sub.shape[0], :sub.shape[1]] = sub
    return out


def summa_compute(r_unused, errors, size, c, M, N, K, chunks, seed=2, uid=[0]):
    uid[0] += 1
    rng = np.random.default_rng(seed)
    a, b, c0 = rng.standard_normal((M, K)), rng.standard_normal((K, N)), rng.standard_normal((M, N))
    alpha, beta = 1.5, -0.5
    want = alpha * a @ b + beta * c0

    def rank(q):
        t = TTopo("summa%d" % uid[0], 0, q, size, c, chunks)
        plan = C.c_void_p()
        rs.ok(L.cap_summa_plan_create(C.byref(plan), t.handle, M, N, K, chunks), "cap_summa_plan_create")
        ml, nl, kl = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        L.cap_summa_local_dims(plan, C.byref(ml), C.byref(nl), C.byref(kl))
        ml, nl, kl = ml.value, nl.value, kl.value
        pa, pb, pc = cyc_piece(a, t.x, t.y, t.d), cyc_piece(b, t.x, t.y, t.d), cyc_piece(c0, t.x, t.y, t.d)
        assert pa.shape == (ml, kl) and pb.shape == (kl, nl) and pc.shape == (ml, nl), (pa.shape, pb.shape, pc.shape, ml, nl, kl)
        A = rs.dmalloc(8 * ml * kl); B = rs.dmalloc(8 * kl * nl); Cc = rs.dmalloc(8 * ml * nl)
        view(A, ml, kl)[:] = pa; view(B, kl, nl)[:] = pb; view(Cc, ml, nl)[:] = pc
        rs.ok(L.cap_summa_dgemm(plan, alpha, A, ml, B, kl, beta, Cc, ml, None), "cap_summa_dgemm")
        got = view(Cc, ml, nl).copy()
        rs.ok(L.cap_summa_plan_destroy(plan), "cap_summa_plan_destroy")
        xy = (t.x, t.y, t.d)
        t.close()
        for z in (A, B, Cc):
            shim.hipFree(z)
        return xy, got
    worst = 0.0
    for (x, y, d), got in run_ranks(size, rank):
        worst = max(worst, rel(got, cyc_piece(want, x, y, d)))
    errors["C"] = worst


def desc_compute(r, errors, m, n, nb, Pr, Pc, kind, seed=8):
    """descriptors with pinned staging: host GLOBAL matrix -> every position's piece -> back (desc.hip: pure host packing code and
    copies - no kernel at all), for the block-cyclic kind (nb x nb blocks on Pr x Pc) and upstream's element-cyclic kind"""
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((m, n))
    hostg = np.asfortranarray(g)
    back = np.zeros((m, n), order="F")
    worst_piece = worst_local = 0.0
    for pr in range(Pr):
        for pc in range(Pc):
            d = C.c_void_p()
            if kind == 1:
                rs.ok(L.cap_desc_create_bc(C.byref(d), n, m, nb, Pr, Pc, pr, pc, None, 0), "cap_desc_create_bc")
                rows, cols = bc_indices(m, nb, Pr, pr), bc_indices(n, nb, Pc, pc)
                want = g[np.ix_(rows, cols)]
            else:
                rs.ok(L.cap_desc_create(C.byref(d), n, m, Pc, Pr), "cap_desc_create")
                rs.ok(L.cap_desc_set_position(d, pc, pr), "cap_desc_set_position")
                want = np.zeros(((m + Pr - 1) // Pr, (n + Pc - 1) // Pc)); sub = g[pr::Pr, pc::Pc]; want[:sub.shape[0], :sub.shape[1]] = sub
            lr, lc, ld = int(L.cap_desc_get(d, 3)), int(L.cap_desc_get(d, 2)), int(L.cap_desc_get(d, 4))
            assert (lr, lc) == want.shape, ((lr, lc), want.shape)
            rs.ok(r.call("desc_import_host_global", L.cap_desc_import_host_global, d, hostg.ctypes.data_as(C.c_void_p), m, r.stream), "import_host_global")
            L.cap_desc_data.restype = C.c_void_p
            if lr and lc:
                worst_piece = max(worst_piece, rel(view(L.cap_desc_data(d), lr, lc, ld), want))
            rs.ok(r.call("desc_export_host_global", L.cap_desc_export_host_global, d, back.ctypes.data_as(C.c_void_p), m, r.stream), "export_host_global")
            # the local import / export pair on a compact host piece
            loc = np.asfortranarray(rng.standard_normal((max(lr, 1), max(lc, 1)))); got = np.zeros_like(loc, order="F")
            rs.ok(r.call("desc_import_host", L.cap_desc_import_host, d, loc.ctypes.data_as(C.c_void_p), max(lr, 1), r.stream), "import_host")
            rs.ok(r.call("desc_export_host", L.cap_desc_export_host, d, got.ctypes.data_as(C.c_void_p), max(lr, 1), r.stream), "export_host")
            if lr and lc:
                worst_local = max(worst_local, rel(got[:lr, :lc], loc[:lr, :lc]))
            rs.ok(L.cap_desc_destroy(d), "cap_desc_destroy")
    errors["pieces"] = worst_piece
    errors["every element comes back to its global place"] = rel(back, g)
    errors["local round trip"] = worst_local


def summa_tri_compute(r_unused, errors, size, c, M, N, K, chunks, seed=5, uid=[0]):
    """matmult::summa TRMM and SYRK overloads + util::transpose on the d x d x c grid (summa.hpp:46-161, util.hpp:232-247)"""
    uid[0] += 1
    rng = np.random.default_rng(seed)
    tm = np.linalg.cholesky(spd(M, seed)).T; tn = np.linalg.cholesky(spd(N, seed + 1)).T       # upper triangular, globally
    b = rng.standard_normal((M, N)); a_t = rng.standard_normal((K, N)); a_n = rng.standard_normal((N, K))
    c0 = rng.standard_normal((N, N)); c0 = c0 + c0.T
    LEFT, RIGHT, UPPER, NT, TR, NONUNIT = 0, 1, 1, 0, 1, 0

    def rank(q):
        t = TTopo("stri%d" % uid[0], 0, q, size, c, chunks)
        d, x, y = t.d, t.x, t.y
        out = {}

        def dev(piece):
            p = rs.dmalloc(8 * max(piece.size, 1))
            view(p, piece.shape[0], piece.shape[1])[:] = piece
            return p
        for (side, trans, tg, nm) in ((LEFT, NT, tm, "L N"), (LEFT, TR, tm, "L T"), (RIGHT, NT, tn, "R N"), (RIGHT, TR, tn, "R T")):
            td = tg.shape[0]
            tp_, bp = cyc_piece(tg, x, y, d), cyc_piece(b, x, y, d)
            T, B = dev(tp_), dev(bp)
            if trans == TR:                                   # upstream's call-site preparation (cholinv.hpp:115): my partner's piece
                tmp = rs.dmalloc(8 * tp_.size)
                rs.ok(L.cap_util_transpose(t.handle, T, tmp, tp_.size, None), "cap_util_transpose")
                out["transpose " + nm] = rel(view(T, *tp_.shape), cyc_piece(tg, y, x, d))
                shim.hipFree(tmp)
            plan = C.c_void_p()
            rs.ok(L.cap_summa_plan_create(C.byref(plan), t.handle, M, N, td, chunks), "cap_summa_plan_create")
            rs.ok(L.cap_summa_dtrmm(plan, side, UPPER, trans, NONUNIT, 0.75, T, tp_.shape[0], 0, B, bp.shape[0], None), "cap_summa_dtrmm")
            op = tg.T if trans == TR else tg
            want = 0.75 * (op @ b if side == LEFT else b @ op)
            out["trmm " + nm] = rel(view(B, *bp.shape), cyc_piece(want, x, y, d))
            rs.ok(L.cap_summa_plan_destroy(plan), "cap_summa_plan_destroy")
            shim.hipFree(T); shim.hipFree(B)
        for (trans, ag, nm) in ((TR, a_t, "T"), (NT, a_n, "N")):
            for beta in (1.0, 0.0):
                ap, cp = cyc_piece(ag, x, y, d), cyc_piece(c0, x, y, d)
                A, Cc = dev(ap), dev(cp)
                plan = C.c_void_p()
                rs.ok(L.cap_summa_plan_create(C.byref(plan), t.handle, N, N, K, chunks), "cap_summa_plan_create")
                rs.ok(L.cap_summa_dsyrk(plan, UPPER, trans, -1.0, A, ap.shape[0], beta, Cc, cp.shape[0], 0, None), "cap_summa_dsyrk")
                g = ag.T @ ag if trans == TR else ag @ ag.T
                out["syrk %s beta=%g" % (nm, beta)] = rel(view(Cc, *cp.shape), cyc_piece(-g + beta * c0, x, y, d))
                out["syrk %s: A untouched" % nm] = rel(view(A, *ap.shape), ap)
                rs.ok(L.cap_summa_plan_destroy(plan), "cap_summa_plan_destroy")
                shim.hipFree(A); shim.hipFree(Cc)
        t.close()
        return out
    for o in run_ranks(size, rank):
        for k, v in o.items():
            errors[k] = max(errors.get(k, 0.0), v)


def cacqr_compute(r_unused, errors, m, n, iters, P, seed=3, uid=[0], a=None, want=None):
    """CholeskyQR / CholeskyQR2 on the 1D grid: row-cyclic pieces of a tall matrix; Q^T Q = I, Q R = A, R upper with a positive diagonal"""
    uid[0] += 1
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((m, n)) if a is None else a
    ml = (m + P - 1) // P                                     # ragged M: zero rows pad the last pieces, as upstream's generator does (structure.hpp:96-101)

    def rank(p):
        comm = TComm(p, P, "cacqr%d" % uid[0])
        plan = C.c_void_p()
        rs.ok(L.cap_cacqr_plan_create(C.byref(plan), ml, n, iters, comm.handle), "cap_cacqr_plan_create")
        A = rs.dmalloc(8 * ml * n)
        view(A, ml, n)[:] = 0.0
        view(A, ml, n)[:len(range(p, m, P))] = a[p::P]
        info = C.c_int64(-1)
        for _ in range(2):
            rs.ok(L.cap_cacqr_factor(plan, A, ml, None), "cap_cacqr_factor")
        L.cap_cacqr_info(plan, None, C.byref(info))
        ldq, ldr = C.c_int64(0), C.c_int64(0)
        L.cap_cacqr_Q_ptr.restype = C.c_void_p; L.cap_cacqr_R_ptr.restype = C.c_void_p
        qp = L.cap_cacqr_Q_ptr(plan, C.byref(ldq)); rp = L.cap_cacqr_R_ptr(plan, C.byref(ldr))
        Q = view(qp, ml, n, ldq.value).copy(); R = view(rp, n, n, ldr.value).copy()
        rs.ok(L.cap_cacqr_plan_destroy(plan), "cap_cacqr_plan_destroy")
        comm.close(); shim.hipFree(A)
        return Q, R, info.value
    res = run_ranks(P, rank)
    Q = np.zeros((m, n))
    for p, (Qp, R, info) in enumerate(res):
        Q[p::P] = Qp[:len(range(p, m, P))]
        errors["rows of padding stay zero"] = max(errors.get("rows of padding stay zero", 0.0), float(np.abs(Qp[len(range(p, m, P)):]).max(initial=0.0)))
        errors["info"] = max(errors.get("info", 0.0), float(abs(info)))
    R = np.triu(res[0][1])
    errors["R replicated"] = max(rel(np.triu(x[1]), R) for x in res)
    errors["A - QR"] = rel(Q @ R, a)
    errors["Q^T Q - I"] = float(np.linalg.norm(Q.T @ Q - np.eye(n)) / np.sqrt(n)) if iters >= 2 else 0.0
    qr_r = np.linalg.qr(a, mode="r")
    errors["R vs LAPACK"] = rel(R, qr_r * np.sign(np.diag(qr_r))[:, None]) if iters >= 2 else 0.0
    if want is not None:
        errors["Q vs the reference's"] = rel(Q, want[0]); errors["R vs the reference's"] = rel(R, np.triu(want[1]))


def cholinv_cyclic_compute(r_unused, errors, n, ci, c, d, bc=-2, nb=128, seed=0, uid=[0]):
    """the reference's layout end to end: element-cyclic pieces on the d x d x c grid in, pieces of R and R^-1 out (cholinv.hpp:30-46),
    the 1 x P block-column plan and the redistribution (redist.hip) in between"""
    uid[0] += 1
    size = d * d * c
    a = spd(n, seed)
    rref = np.linalg.cholesky(a).T
    _, riref = orc.cholinv(a, max(ci, 0), 1, bc, c, d)        # upstream's pattern of R^-1 on this grid (root block, base-case rule)

    def rank(q):
        dd, x, y, z = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        rs.ok(L.cap_topo_coords(0, q, size, c, C.byref(dd), C.byref(x), C.byref(y), C.byref(z)), "cap_topo_coords")
        comm = TComm(q, size, "cyc%d" % uid[0])
        plan = C.c_void_p()
        rs.ok(L.cap_cholinv_plan_create(C.byref(plan), n, ci, 1, bc, b"U", comm.handle), "cap_cholinv_plan_create")
        rs.ok(L.cap_cholinv_set_option(plan, b"nb", nb), "nb")
        rs.ok(L.cap_cholinv_set_option(plan, b"cyclic_c", c), "cyclic_c")
        pa = cyc_piece(a, x.value, y.value, d)
        e = pa.shape[0]
        A = rs.dmalloc(8 * e * e); out = rs.dmalloc(8 * e * e)
        view(A, e, e)[:] = pa
        info = C.c_int64(-1)
        rs.ok(L.cap_cholinv_factor(plan, A, e, None), "cap_cholinv_factor")
        L.cap_cholinv_info(plan, None, C.byref(info))
        rs.ok(L.cap_cholinv_get_R(plan, out, e, None), "cap_cholinv_get_R")
        R = view(out, e, e).copy()
        Ri = None
        if ci >= 0:
            rs.ok(L.cap_cholinv_get_Rinv(plan, out, e, None), "cap_cholinv_get_Rinv")
            Ri = view(out, e, e).copy()
        rs.ok(L.cap_cholinv_plan_destroy(plan), "cap_cholinv_plan_destroy")
        comm.close(); shim.hipFree(A); shim.hipFree(out)
        return x.value, y.value, R, Ri, info.value
    eR = eRi = 0.0
    for x, y, R, Ri, info in run_ranks(size, rank):
        eR = max(eR, rel(R, cyc_piece(rref, x, y, d)))
        if Ri is not None:
            eRi = max(eRi, rel(Ri, cyc_piece(riref, x, y, d)))
        errors["info"] = max(errors.get("info", 0.0), float(abs(info)))
    errors["R pieces"] = eR
    if ci >= 0:
        errors["Rinv pieces"] = eRi


def mixed_compute(r, errors, n, nrhs, opts=(), seed=4, reps=1):
    """bf16 factorization + fp64 refinement on one GPU (mixed.hip): a low-precision factor, a solution to fp64 accuracy"""
    a = spd(n, seed); rng = np.random.default_rng(seed + 1); b = rng.standard_normal((n, nrhs))
    plan = C.c_void_p()
    rs.ok(L.cap_mpchol_plan_create(C.byref(plan), n, nrhs), "cap_mpchol_plan_create")
    for k, v in opts:
        rs.ok(L.cap_mpchol_set_option(plan, k.encode(), v), "mpchol set_option " + k)
    A = rs.dmalloc(8 * n * n); B = rs.dmalloc(8 * n * nrhs); X = rs.dmalloc(8 * n * nrhs)
    view(A, n, n)[:] = a; view(B, n, nrhs)[:] = b
    info = C.c_int64(-1); it = C.c_int(0); rr = C.c_double(0)
    for _ in range(reps):
        rs.ok(r.call("mpchol_factor", L.cap_mpchol_factor, plan, A, n, r.stream), "cap_mpchol_factor")
    r.call("mpchol_info", L.cap_mpchol_info, plan, r.stream, C.byref(info))
    rs.ok(r.call("mpchol_solve", L.cap_mpchol_solve, plan, A, n, B, n, X, n, nrhs, 30, 1e-15, C.byref(it), C.byref(rr), r.stream), "cap_mpchol_solve")
    x = view(X, n, nrhs).copy()
    ld = C.c_int64(0)
    L.cap_mpchol_R32_ptr.restype = C.c_void_p
    rp = L.cap_mpchol_R32_ptr(plan, C.byref(ld))
    r32 = np.ctypeslib.as_array((C.c_float * (ld.value * n)).from_address(rp)).reshape((n, ld.value)).T[:n]
    e32 = rel(np.triu(r32.astype(np.float64)), np.linalg.cholesky(a).T)
    errors["info"] = float(abs(info.value))
    errors["factor is a bf16-update factor (1e-6 < err < 5e-2)"] = 0.0 if 1e-6 < e32 < 5e-2 else e32 + 1.0
    errors["sweeps within 1..25"] = 0.0 if 1 <= it.value <= 25 else float(it.value) + 1.0
    errors["B - A X (scaled to the tolerance)"] = float(np.linalg.norm(a @ x - b) / np.linalg.norm(b)) * 1e-3      # <= 2e-14 passes the 2e-11 gate
    errors["X vs fp64 solve (scaled)"] = rel(x, np.linalg.solve(a, b)) * 1e-2
    rs.ok(L.cap_mpchol_plan_destroy(plan), "cap_mpchol_plan_destroy")
    for q in (A, B, X):
        shim.hipFree(q)


def dmp_compute(r_unused, errors, n, nb, P, nrhs=5, seed=6, uid=[0]):
    """the same on P ranks (dist_mixed.hip): block columns of A per rank, B and X replicated"""
    uid[0] += 1
    a = spd(n, seed); rng = np.random.default_rng(seed + 1); b = rng.standard_normal((n, nrhs))

    def rank(p):
        comm = TComm(p, P, "dmp%d" % uid[0])
        plan = C.c_void_p()
        rs.ok(L.cap_dmp_plan_create(C.byref(plan), n, nb, nrhs, comm.handle), "cap_dmp_plan_create")
        cols = bc_indices(n, nb, P, p); lc = len(cols)
        assert lc == int(L.cap_dmp_local_cols(plan))
        A = rs.dmalloc(8 * n * max(lc, 1)); B = rs.dmalloc(8 * n * nrhs); X = rs.dmalloc(8 * n * nrhs)
        if lc:
            view(A, n, lc)[:] = a[:, cols]
        view(B, n, nrhs)[:] = b
        info = C.c_int64(-1); it = C.c_int(0); rr = C.c_double(0)
        for _ in range(2):
            rs.ok(L.cap_dmp_factor(plan, A, n, None), "cap_dmp_factor")
        L.cap_dmp_info(plan, None, C.byref(info))
        rs.ok(L.cap_dmp_solve(plan, A, n, B, n, X, n, nrhs, 30, 1e-15, C.byref(it), C.byref(rr), None), "cap_dmp_solve")
        x = view(X, n, nrhs).copy()
        rs.ok(L.cap_dmp_plan_destroy(plan), "cap_dmp_plan_destroy")
        comm.close()
        for q in (A, B, X):
            shim.hipFree(q)
        return x, it.value, info.value
    res = run_ranks(P, rank)
    x = res[0][0]
    errors["info"] = float(max(abs(v[2]) for v in res))
    errors["every rank ends with the same X"] = max(rel(v[0], x) for v in res)
    errors["sweeps within 1..25"] = 0.0 if all(1 <= v[1] <= 25 for v in res) else 99.0
    errors["B - A X (scaled to the tolerance)"] = float(np.linalg.norm(a @ x - b) / np.linalg.norm(b)) * 1e-3
    errors["X vs fp64 solve (scaled)"] = rel(x, np.linalg.solve(a, b)) * 1e-2


def cacqr_grid_compute(r_unused, errors, size, c, m, n, iters=2, seed=9, uid=[0], a=None, want=None):
    """qr::cacqr on the c x d x c grid of a topo::rect bundle (cacqr.hpp:44-215): rows cyclic over d, columns over c, layers replicas"""
    uid[0] += 1
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((m, n)) if a is None else a
    d = size // (c * c)

    def rank(q):
        t = TTopo("cqg%d" % uid[0], 1, q, size, c)
        plan = C.c_void_p()
        rs.ok(L.cap_cacqr_plan_create_grid(C.byref(plan), m, n, iters, t.handle), "cap_cacqr_plan_create_grid")
        ml, nl = (m + d - 1) // d, n // c
        pa = np.zeros((ml, nl)); sub = a[t.y::d, t.x::c]; pa[:sub.shape[0], :sub.shape[1]] = sub
        A = rs.dmalloc(8 * ml * nl); out = rs.dmalloc(8 * n * n)
        view(A, ml, nl)[:] = pa
        info = C.c_int64(-1)
        for _ in range(2):
            rs.ok(L.cap_cacqr_factor(plan, A, ml, None), "cap_cacqr_factor")
        L.cap_cacqr_info(plan, None, C.byref(info))
        ldq, ldr = C.c_int64(0), C.c_int64(0)
        L.cap_cacqr_Q_ptr.restype = C.c_void_p; L.cap_cacqr_R_ptr.restype = C.c_void_p
        qp = L.cap_cacqr_Q_ptr(plan, C.byref(ldq)); rp = L.cap_cacqr_R_ptr(plan, C.byref(ldr))
        Q = view(qp, ml, nl, ldq.value).copy(); R = view(rp, n, n, ldr.value).copy()
        rs.ok(L.cap_cacqr_R_piece(plan, out, nl, None), "cap_cacqr_R_piece")
        Rp = view(out, nl, nl).copy()
        rs.ok(L.cap_cacqr_plan_destroy(plan), "cap_cacqr_plan_destroy")
        xyz = (t.x, t.y, t.z)
        t.close(); shim.hipFree(A); shim.hipFree(out)
        return xyz, Q, R, Rp, info.value
    res = run_ranks(size, rank)
    Q = np.zeros((m, n)); Rg = np.zeros((n, n))
    R = np.triu(res[0][2])
    for (x, y, z), Qp, Rd, Rp, info in res:
        rows = len(range(y, m, d))
        if z == 0:
            Q[y::d, x::c] = Qp[:rows]
        if z == 0 and y < c:
            Rg[y::c, x::c] = Rp
        errors["info"] = max(errors.get("info", 0.0), float(abs(info)))
        errors["R replicated"] = max(errors.get("R replicated", 0.0), rel(np.triu(Rd), R))
    errors["layers are replicas"] = max(rel(Qp[:len(range(y, m, d))], Q[y::d, x::c]) for (x, y, z), Qp, _, _, _ in res)
    errors["R pieces (c x c cyclic)"] = rel(Rg, R)
    errors["A - QR"] = rel(Q @ R, a)
    if iters >= 2:
        errors["Q^T Q - I"] = float(np.linalg.norm(Q.T @ Q - np.eye(n)) / np.sqrt(n))
        qr_r = np.linalg.qr(a, mode="r")
        errors["R vs LAPACK"] = rel(R, qr_r * np.sign(np.diag(qr_r))[:, None])
    if want is not None:
        errors["Q vs the reference's"] = rel(Q, want[0]); errors["R vs the reference's"] = rel(R, np.triu(want[1]))


def cyclic2d_compute(r_unused, errors, n, nb, size, c, Pr, seed=0, uid=[0]):
    """a caller with the reference's element-cyclic pieces on the d x d x c grid reaches the Pr x Pc plan: redistribute (redist.hip, one
    all-to-all), factor through the DESCRIPTOR entry points (cap_desc_create_bc view + cap_dist2d_factor_desc / get_R_desc), redistribute
    back - the pieces of R as construct_R returns them"""
    uid[0] += 1
    a = spd(n, seed)
    rref = np.linalg.cholesky(a).T
    Pc = size // Pr

    def rank(q):
        world = TComm(q, size, "c2d%d" % uid[0])
        rp = C.c_void_p()
        rs.ok(L.cap_redist_plan_create(C.byref(rp), n, nb, world.handle, c, Pr), "cap_redist_plan_create")
        e, lr, lc, d, x, y = (int(L.cap_redist_get(rp, w)) for w in (0, 1, 2, 3, 5, 6))
        pr, pc = int(L.cap_redist_get(rp, 10)), int(L.cap_redist_get(rp, 11))
        row = TComm(pc, Pc, "c2d%d_r%d" % (uid[0], pr)); col = TComm(pr, Pr, "c2d%d_c%d" % (uid[0], pc))
        plan = C.c_void_p()
        rs.ok(L.cap_dist2d_plan_create(C.byref(plan), n, nb, world.handle, Pr, row.handle, col.handle), "cap_dist2d_plan_create")
        pa = cyc_piece(a, x, y, d)
        assert pa.shape == (e, e)
        P = rs.dmalloc(8 * e * e); bc = rs.dmalloc(8 * max(lr, 1) * max(lc, 1)); out = rs.dmalloc(8 * max(lr, 1) * max(lc, 1)); Pout = rs.dmalloc(8 * e * e)
        view(P, e, e)[:] = pa
        rs.ok(L.cap_redistribute_cyclic_to_bc(rp, P, e, bc, max(lr, 1), None), "cyclic_to_bc")
        rows, cols = bc_indices(n, nb, Pr, pr), bc_indices(n, nb, Pc, pc)
        e_bc = rel(view(bc, lr, lc), a[np.ix_(rows, cols)]) if lr and lc else 0.0
        dA, dR = C.c_void_p(), C.c_void_p()
        rs.ok(L.cap_desc_create_bc(C.byref(dA), n, n, nb, Pr, Pc, pr, pc, bc, max(lr, 1)), "cap_desc_create_bc")
        rs.ok(L.cap_desc_create_bc(C.byref(dR), n, n, nb, Pr, Pc, pr, pc, out, max(lr, 1)), "cap_desc_create_bc")
        rs.ok(L.cap_dist2d_factor_desc(plan, dA, None), "cap_dist2d_factor_desc")
        info = C.c_int64(-1)
        L.cap_dist2d_info(plan, None, C.byref(info))
        rs.ok(L.cap_dist2d_get_R_desc(plan, dR, None), "cap_dist2d_get_R_desc")
        rs.ok(L.cap_redistribute_bc_to_cyclic(rp, out, max(lr, 1), Pout, e, None), "bc_to_cyclic")
        Rp = view(Pout, e, e).copy()
        for h in (dA, dR):
            rs.ok(L.cap_desc_destroy(h), "cap_desc_destroy")
        rs.ok(L.cap_dist2d_plan_destroy(plan), "cap_dist2d_plan_destroy"); rs.ok(L.cap_redist_plan_destroy(rp), "cap_redist_plan_destroy")
        for cm in (row, col, world):
            cm.close()
        for z in (P, bc, out, Pout):
            shim.hipFree(z)
        return x, y, d, Rp, e_bc, info.value
    res = run_ranks(size, rank)
    errors["block-cyclic pieces after the all-to-all"] = max(v[4] for v in res)
    errors["info"] = float(max(abs(v[5]) for v in res))
    errors["R pieces"] = max(rel(Rp, cyc_piece(rref, x, y, d)) for x, y, d, Rp, _, _ in res)


def utils_compute(r, errors, n, seed=12):
    """serialize (packed-upper windows), util::remove_triangle, the element-cyclic import / export, the validator's residual terms"""
    rng = np.random.default_rng(seed)
    a = spd(n, seed); rr = np.linalg.cholesky(a).T
    A = rs.dmalloc(8 * n * n); R = rs.dmalloc(8 * n * n); W = rs.dmalloc(8 * n * n); out2 = rs.dmalloc(16)
    view(A, n, n)[:] = a; view(R, n, n)[:] = rr
    rs.ok(r.call("cholesky_residual_terms", L.cap_cholesky_residual_terms, A, n, R, n, n, W, out2, r.stream), "cap_cholesky_residual_terms")
    t = view(out2, 2, 1)[:, 0]
    errors["residual of the exact factor"] = float(np.sqrt(t[0] / t[1])) * 1e4          # 1e-15 passes the gate
    view(R, n, n)[0, n - 1] += 1e-3
    rs.ok(r.call("cholesky_residual_terms", L.cap_cholesky_residual_terms, A, n, R, n, n, W, out2, r.stream), "cap_cholesky_residual_terms")
    t = view(out2, 2, 1)[:, 0]
    errors["residual reacts to a perturbed factor"] = 0.0 if np.sqrt(t[0] / t[1]) > 1e-7 else 1.0
    # packed upper storage: window (r0, c0, rows, cols) of a rect matrix -> packed -> back into a zero matrix
    pk = rs.dmalloc(8 * n * (n + 1) // 2); Z = rs.dmalloc(8 * n * n)
    rs.ok(r.call("copy_window rect->packed", L.cap_copy_window, A, 0, n, 0, 0, pk, 1, 0, 0, 0, n, n, 1, 0, r.stream), "cap_copy_window")
    view(Z, n, n)[:] = 0.0
    rs.ok(r.call("copy_window packed->rect", L.cap_copy_window, pk, 1, 0, 0, 0, Z, 0, n, 0, 0, n, n, 1, 1, r.stream), "cap_copy_window")
    errors["packed round trip"] = rel(view(Z, n, n), np.triu(a))
    h = n // 3
    view(Z, n, n)[:] = 7.0
    rs.ok(r.call("copy_window sub-window", L.cap_copy_window, A, 0, n, h, h, Z, 0, n, 2, 5, h, h + 3, 0, 0, r.stream), "cap_copy_window")
    want = np.full((n, n), 7.0); want[2:2 + h, 5:5 + h + 3] = a[h:2 * h, h:2 * h + 3]
    errors["window copy touches its window only"] = rel(view(Z, n, n), want)
    view(Z, n, n)[:] = a
    rs.ok(r.call("remove_triangle", L.cap_remove_triangle, Z, n, n, n, 0, 0, 1, 1, r.stream), "cap_remove_triangle")
    errors["remove_triangle (upper kept)"] = rel(view(Z, n, n), np.triu(a))
    # element-cyclic piece (x, y) of a 3 x 2 grid out of a dense matrix and back
    dx, dy, x, y = 3, 2, 1, 1
    pl_r, pl_c = (n + dy - 1) // dy, (n + dx - 1) // dx
    Pm = rs.dmalloc(8 * pl_r * pl_c)
    rs.ok(r.call("cyclic_export", L.cap_cyclic_export, A, n, Pm, pl_r, n, n, x, y, dx, dy, r.stream), "cap_cyclic_export")
    want = np.zeros((pl_r, pl_c)); sub = a[y::dy, x::dx]; want[:sub.shape[0], :sub.shape[1]] = sub
    errors["cyclic export"] = rel(view(Pm, pl_r, pl_c), want)
    view(Z, n, n)[:] = 0.0
    rs.ok(r.call("cyclic_import", L.cap_cyclic_import, Pm, pl_r, Z, n, n, n, x, y, dx, dy, r.stream), "cap_cyclic_import")
    want = np.zeros((n, n)); want[y::dy, x::dx] = a[y::dy, x::dx]
    errors["cyclic import"] = rel(view(Z, n, n), want)
    for q in (A, R, W, out2, pk, Z, Pm):
        shim.hipFree(q)


GOLD = os.path.join(ROOT, "tests", "golden")


def golden_cholinv_1rank(r, errors, fname):
    """the single-GPU plan against what the REAL reference left on one rank (tests/golden/*.npz, dumped by oracle/_ref)"""
    g = np.load(os.path.join(GOLD, fname)) if isinstance(fname, str) else fname
    n, ci, split, bc = int(g["n"]), int(g["complete_inv"]), int(g["split"]), int(g["bc_mult_dim"])
    a = np.array(g["A"])
    plan = C.c_void_p()
    rs.ok(L.cap_cholinv_plan_create(C.byref(plan), n, ci, split, bc, b"U", None), "cap_cholinv_plan_create")
    A = rs.dmalloc(8 * n * n); out = rs.dmalloc(8 * n * n)
    view(A, n, n)[:] = a
    rs.ok(r.call("cholinv_factor", L.cap_cholinv_factor, plan, A, n, r.stream), "cap_cholinv_factor")
    rs.ok(r.call("cholinv_get_R", L.cap_cholinv_get_R, plan, out, n, r.stream), "cap_cholinv_get_R")
    errors["R vs the reference's"] = rel(view(out, n, n), np.triu(g["R"]))
    rs.ok(r.call("cholinv_get_Rinv", L.cap_cholinv_get_Rinv, plan, out, n, r.stream), "cap_cholinv_get_Rinv")
    ri = view(out, n, n).copy(); ref = np.triu(g["Rinv"])
    errors["Rinv vs the reference's"] = rel(ri, ref)
    errors["same empty block"] = float(np.count_nonzero((ri != 0) != (ref != 0)))
    rs.ok(L.cap_cholinv_plan_destroy(plan), "cap_cholinv_plan_destroy")
    shim.hipFree(A); shim.hipFree(out)


def golden_cholinv_8ranks(r_unused, errors, fname, uid=[0]):
    """c d^2 ranks on the reference's own d x d x c grid (the committed dumps: 2 x 2 x 2): every rank's pieces of R and R^-1 against the PIECES the real reference dumped"""
    uid[0] += 1
    g = np.load(os.path.join(GOLD, fname)) if isinstance(fname, str) else fname
    n, ci, split, bc, c, d = (int(g[k]) for k in ("n", "complete_inv", "split", "bc_mult_dim", "c", "d"))
    a = np.array(g["A"]); pieces = np.array(g["pieces"]); coords = np.array(g["rank_coords"])
    size = c * d * d

    def rank(q):
        assert int(coords[q][0]) == q
        x, y = int(coords[q][1]), int(coords[q][2])
        comm = TComm(q, size, "gold%d" % uid[0])
        plan = C.c_void_p()
        rs.ok(L.cap_cholinv_plan_create(C.byref(plan), n, ci, split, bc, b"U", comm.handle), "cap_cholinv_plan_create")
        rs.ok(L.cap_cholinv_set_option(plan, b"nb", 128), "nb")
        rs.ok(L.cap_cholinv_set_option(plan, b"cyclic_c", c), "cyclic_c")
        pa = cyc_piece(a, x, y, d); e = pa.shape[0]
        A = rs.dmalloc(8 * e * e); out = rs.dmalloc(8 * e * e)
        view(A, e, e)[:] = pa
        rs.ok(L.cap_cholinv_factor(plan, A, e, None), "cap_cholinv_factor")
        rs.ok(L.cap_cholinv_get_R(plan, out, e, None), "cap_cholinv_get_R")
        R = view(out, e, e).copy()
        rs.ok(L.cap_cholinv_get_Rinv(plan, out, e, None), "cap_cholinv_get_Rinv")
        Ri = view(out, e, e).copy()
        rs.ok(L.cap_cholinv_plan_destroy(plan), "cap_cholinv_plan_destroy")
        comm.close(); shim.hipFree(A); shim.hipFree(out)
        gi = np.arange(e)[:, None] * d + y; gj = np.arange(e)[None, :] * d + x
        upper = (gi <= gj) & (gi < n) & (gj < n)               # util::remove_triangle's mask: the dump keeps the raw local triangle
        ref_r, ref_ri = pieces[q][1], pieces[q][2]
        tiny = 1e-300                                          # (a piece with nothing on or above the diagonal: N < d)
        return (float(np.linalg.norm((R - ref_r)[upper]) / max(np.linalg.norm(ref_r[upper]), tiny)), float(np.linalg.norm((Ri - ref_ri)[upper]) / max(np.linalg.norm(ref_ri[upper]), tiny)),
                float(np.count_nonzero((Ri[upper] != 0) != (ref_ri[upper] != 0))), float(np.count_nonzero(R[~upper]) + np.count_nonzero(Ri[~upper])),
                rel(pa, pieces[q][0]))
    res = run_ranks(size, rank)
    errors["R pieces vs the reference's"] = max(v[0] for v in res)
    errors["Rinv pieces vs the reference's"] = max(v[1] for v in res)
    errors["same empty root block, piece by piece"] = max(v[2] for v in res)
    errors["nothing below the global diagonal"] = max(v[3] for v in res)
    errors["same input pieces"] = max(v[4] for v in res)


def golden_cacqr(r_unused, errors, fname, uid=[0]):
    """CholeskyQR / CholeskyQR2 against the real reference's 8-rank runs (1D grid and the c x d x c grid)"""
    g = np.load(os.path.join(GOLD, fname)) if isinstance(fname, str) else fname
    m, n, variant, c, d = (int(g[k]) for k in ("m", "n", "variant", "c", "d"))
    a = np.array(g["A"])
    e = {}
    if c == 1:
        cacqr_compute(None, e, m, n, variant, d, a=a, want=(np.array(g["Q"]), np.array(g["R"])))
    else:
        cacqr_grid_compute(None, e, c * c * d, c, m, n, variant, a=a, want=(np.array(g["Q"]), np.array(g["R"])))
    for k in ("info", "Q vs the reference's", "R vs the reference's", "A - QR"):
        errors[k] = e[k]


def golden_summa(r_unused, errors, g, uid=[0]):
    """matmult::summa's GEMM / TRMM / SYRK overloads against the REAL reference (oracle/ref/drv_summa.cpp) on its cube: every rank gets the
    very pieces the reference's rank at the same (x, y, z) held (the triangular operand as upstream's packed piece and, a second time, as
    the rect piece; util::transpose first where upstream's call site has it) and must end with the pieces the reference ended with"""
    uid[0] += 1
    if isinstance(g, str):                                     # a committed dump (tests/golden/make_golden.py: summa_dump_case)
        z = np.load(os.path.join(GOLD, g))
        g = {key: (float(z[key]) if key in ("alpha", "beta") else int(z[key])) for key in ("op", "m", "n", "k", "c", "chunks", "alpha", "beta")}
        g["ranks"] = []
        for q, co in enumerate(z["coords"]):
            arrs, i = [], 0
            while "shape_%d_%d" % (q, i) in z:
                rows, cols, packed = (int(v) for v in z["shape_%d_%d" % (q, i)])
                arrs.append((rows, cols, packed, np.array(z["data_%d_%d" % (q, i)]))); i += 1
            g["ranks"].append((tuple(int(v) for v in co), arrs))
    op, m, n, k, c, chunks, alpha, beta = (g[key] for key in ("op", "m", "n", "k", "c", "chunks", "alpha", "beta"))
    by_coord = {co[1:4]: arrs for co, arrs in g["ranks"]}
    size = len(g["ranks"])
    LEFT, RIGHT, UPPER, NT, TR, NONUNIT = 0, 1, 1, 0, 1, 0

    def unpacked(cols, v):
        a = np.zeros((cols, cols))
        for j in range(cols):
            a[:j + 1, j] = v[j * (j + 1) // 2:j * (j + 1) // 2 + j + 1]
        return a

    def rank(q):
        t = TTopo("gsumma%d" % uid[0], 0, q, size, c, chunks)
        arrs = by_coord[(t.x, t.y, t.z)]
        out = {}

        def dev(vec):
            p = rs.dmalloc(8 * max(vec.size, 1))
            np.ctypeslib.as_array((C.c_double * max(vec.size, 1)).from_address(p.value))[:vec.size] = vec
            return p

        def host(p, cnt):
            return np.ctypeslib.as_array((C.c_double * max(cnt, 1)).from_address(p.value))[:cnt].copy()
        plan = C.c_void_p()
        if op == 0:
            (ml, kl, _, va), (_, nl, _, vb), (_, _, _, vc), (_, _, _, want) = arrs
            rs.ok(L.cap_summa_plan_create(C.byref(plan), t.handle, m, n, k, chunks), "cap_summa_plan_create")
            A, B, Cc = dev(va), dev(vb), dev(vc)
            rs.ok(L.cap_summa_dgemm(plan, C.c_double(alpha), A, ml, B, kl, C.c_double(beta), Cc, ml, None), "cap_summa_dgemm")
            out["C pieces vs the reference's"] = rel(host(Cc, ml * nl), want)
            out["operands untouched"] = rel(host(A, va.size), va) + rel(host(B, vb.size), vb)
            bufs = [A, B, Cc]
        elif op <= 4:
            (_, tl, _, vt), (ml, nl, _, vb), (_, _, _, want) = arrs
            side, trans, td = (LEFT if op <= 2 else RIGHT), (TR if op in (2, 4) else NT), (m if op <= 2 else n)
            rs.ok(L.cap_summa_plan_create(C.byref(plan), t.handle, m, n, td, chunks), "cap_summa_plan_create")
            bufs = []
            for packed, vec in ((1, vt), (0, unpacked(tl, vt).T.reshape(-1))):      # (column-major image of the rect piece)
                T, B, tmp = dev(vec), dev(vb), rs.dmalloc(8 * max(vec.size, 1))
                if trans == TR:
                    rs.ok(L.cap_util_transpose(t.handle, T, tmp, vec.size, None), "cap_util_transpose")
                rs.ok(L.cap_summa_dtrmm(plan, side, UPPER, trans, NONUNIT, C.c_double(alpha), T, tl, packed, B, ml, None), "cap_summa_dtrmm")
                out["B pieces vs the reference's (T %s)" % ("packed" if packed else "rect")] = rel(host(B, ml * nl), want)
                bufs += [T, B, tmp]
        else:
            (al, ac, _, va), (_, nl, cpk, vc), (_, _, _, want) = arrs
            rs.ok(L.cap_summa_plan_create(C.byref(plan), t.handle, n, n, k, chunks), "cap_summa_plan_create")
            A, Cc = dev(va), dev(vc)
            rs.ok(L.cap_summa_dsyrk(plan, UPPER, TR if op in (5, 7) else NT, C.c_double(alpha), A, al, C.c_double(beta), Cc, nl, cpk, None), "cap_summa_dsyrk")
            out["C pieces vs the reference's"] = rel(host(Cc, want.size), want)
            out["A untouched"] = rel(host(A, va.size), va)
            bufs = [A, Cc]
        rs.ok(L.cap_summa_plan_destroy(plan), "cap_summa_plan_destroy")
        t.close()
        for b in bufs:
            shim.hipFree(b)
        return out
    for o in run_ranks(size, rank):
        for key, v in o.items():
            errors[key] = max(errors.get(key, 0.0), v)


def mp_case(name):
    """a multi-rank case: no trace of its own (the ranks' threads interleave in it) - the structural checks are run_scenarios.py's"""
    def deco(fn):
        if rs.FILTER and rs.FILTER not in name:
            return fn
        shim.shim_reset()
        before = int(shim.shim_unmodelled())
        errors, findings = {}, []
        try:
            fn(None, errors)
        except Exception as e:
            findings.append("exception: %r" % (e,))
        if int(shim.shim_unmodelled()) > before:
            findings.append("kernels without a CPU model were launched")
        for k, v in errors.items():
            if not (v == v) or v > TOL:
                findings.append("%s off by %.3e (tolerance %.1e)" % (k, v, TOL))
        RESULTS.append({"name": name, "errors": errors, "findings": findings, "stats": {}})
        shim.shim_reset()
        return fn
    return deco


def main(out_path):
    shim.shim_set_compute(1)
    for us in (0, 1):
        tag = " [user stream]" if us else " [NULL stream]"
        for (m, n, k) in [(256, 256, 256), (300, 200, 150), (1024, 8, 1024), (64, 64, 64), (640, 384, 128), (130, 70, 33)]:
            case("operators m=%d n=%d k=%d%s" % (m, n, k, tag), us)(lambda r, e, a=(m, n, k): operators_compute(r, e, *a))
        for (n, ci, split, bc, opts) in [
            (1024, -1, 1, 0, ()), (1024, -1, 1, -3, ()), (1000, -1, 1, -2, ()), (64, -1, 1, 0, ()), (200, 1, 1, -2, ()),
            (1024, 1, 1, 0, ()), (1024, 0, 1, 0, ()), (1024, 0, 2, -2, ()), (768, 1, 1, -2, (("inv_overlap", 0),)), (1000, 1, 1, -2, ()),
            (1024, 1, 1, 0, (("inv_fast", 0),)), (1024, -1, 1, 0, (("nb", 128),)), (1024, -1, 1, 0, (("nb", 64), ("outer", 128), ("tail", 0))),
            (2048, -1, 1, 0, (("nb", 128), ("outer", 256), ("tail", 256), ("depth2", 1))),
            (2048, -1, 1, 0, (("nb", 128), ("outer", 256), ("tail", 0), ("depth2", 1), ("use_sb", 0))),
            (2048, -1, 1, 0, (("nb", 128), ("outer", 256), ("tail", 0), ("depth2", 1), ("pair_rest", 0))),
            (2048, -1, 1, 0, (("nb", 128), ("outer", 128), ("tail", 0), ("depth2", 0))),
            (2048, 1, 1, 0, (("nb", 128), ("outer", 256), ("tail", 0), ("depth2", 1), ("inv_start_m", 1 << 30))),
            (2048, 0, 1, 0, (("nb", 128), ("outer", 256), ("tail", 512), ("depth2", 1))),
            (1536, 1, 1, 0, (("nb", 128), ("use_sb", 0))), (1536, -1, 1, 0, (("nb", 128), ("lookahead", 0))), (1536, -1, 1, 0, (("nb", 128), ("fastdiag", 0))),
            (1536, -1, 1, 0, (("nb", 128), ("chain_coop", 0))), (1536, -1, 1, 0, (("nb", 128), ("fuse_copy", 0), ("depth2", 1))),
            (2048, -1, 1, 0, (("nb", 128), ("outer", 256), ("inner_la", 1), ("tail", 0))), (1536, -1, 1, 0, (("nb", 128), ("serial_m", 512))),
            (1536, -1, 1, 0, (("nb", 128), ("reserve", 8))), (1100, -1, 1, 0, (("nb", 128), ("depth2", 1), ("outer", 256), ("tail", 0))),
            (1100, 1, 1, 0, (("nb", 128), ("depth2", 1), ("outer", 256), ("tail", 0))), (1024, 1, 1, 0, (("nb", 256), ("leaf", 32))),
        ]:
            if us and n >= 1536 and not (dict(opts).get("inner_la") or dict(opts).get("reserve") or ci >= 0 and n == 2048):
                continue                                   # (the big cases once; the caller's own stream on the schedules with most streams)
            case("cholinv n=%d ci=%d split=%d bc=%d %s%s" % (n, ci, split, bc, dict(opts) or "", tag), us)(
                lambda r, e, a=(n, ci, split, bc, opts): cholinv_compute(r, e, *a))
    for (n, nb, P, opts, ci) in [(1024, 128, 1, (), -1), (1024, 128, 2, (), -1), (1024, 128, 4, (), -1), (1024, 128, 4, (("safe", 1),), -1),
                                 (1000, 128, 3, (), -1), (1536, 128, 4, (("strip", 2), ("depth2", 1)), -1), (2048, 128, 8, (), -1), (1024, 128, 4, (), 1),
                                 (1024, 128, 4, (), 0), (1000, 128, 3, (("safe", 1),), 1), (1024, 128, 4, (("ipc", 1),), -1), (1000, 128, 3, (("ipc", 1),), 1),
                                 (2048, 128, 8, (("ipc", 1), ("strip", 2)), -1), (1536, 256, 5, (("strip", 1),), 0), (1152, 128, 8, (), -1)]:
        mp_case("dist n=%d nb=%d P=%d %s ci=%d" % (n, nb, P, dict(opts) or "", ci))(lambda r, e, a=(n, nb, P, opts, ci): dist_compute(r, e, *a))
    for (n, nb, Pr, Pc, opts) in [(1024, 128, 1, 1, ()), (1024, 128, 2, 2, ()), (1024, 128, 1, 4, ()), (1536, 128, 2, 4, ()), (1000, 128, 2, 2, ()),
                                  (1024, 128, 2, 2, (("complete_inv", 1),)), (1000, 128, 2, 4, (("complete_inv", 0),)), (1024, 128, 2, 2, (("safe", 1), ("complete_inv", 1))),
                                  (1024, 128, 2, 2, (("ipc", 1),)), (1536, 128, 2, 4, (("ipc", 1), ("complete_inv", 1))), (2048, 128, 4, 4, ()), (1024, 128, 4, 4, (("ipc", 1),)),
                                  (1536, 128, 2, 4, (("strip", 1), ("depth2", 0))), (1152, 128, 4, 8, ())]:
        mp_case("dist2d n=%d nb=%d %dx%d %s" % (n, nb, Pr, Pc, dict(opts) or ""))(lambda r, e, a=(n, nb, Pr, Pc, opts): dist2d_compute(r, e, *a))
    for (size, c, M, N, K, chunks) in [(1, 1, 256, 256, 256, 0), (4, 1, 512, 256, 384, 0), (8, 2, 300, 300, 300, 2), (9, 1, 300, 270, 330, 3), (27, 3, 270, 270, 270, 0),
                                       (4, 1, 301, 200, 257, 2), (8, 2, 256, 128, 512, 4)]:
        mp_case("summa gemm size=%d c=%d %dx%dx%d chunks=%d" % (size, c, M, N, K, chunks))(lambda r, e, a=(size, c, M, N, K, chunks): summa_compute(r, e, *a))
    for (size, c, M, N, K, chunks) in [(4, 1, 256, 192, 130, 0), (8, 2, 300, 200, 153, 2), (9, 1, 270, 180, 99, 0), (1, 1, 256, 128, 64, 0)]:
        mp_case("summa trmm / syrk / transpose size=%d c=%d m=%d n=%d k=%d chunks=%d" % (size, c, M, N, K, chunks))(
            lambda r, e, a=(size, c, M, N, K, chunks): summa_tri_compute(r, e, *a))
    for us in (0, 1):
        for (m, n, nb, Pr, Pc, kind) in [(1000, 1000, 128, 2, 2, 1), (300, 520, 128, 2, 4, 1), (2048, 2048, 256, 1, 4, 1), (1000, 1000, 0, 2, 2, 0), (301, 203, 0, 3, 2, 0),
                                         (100, 100, 128, 2, 4, 1), (3100, 2900, 512, 1, 1, 1)]:
            case("desc %s %dx%d nb=%d grid %dx%d%s" % ("block-cyclic" if kind else "element-cyclic", m, n, nb, Pr, Pc, " [user stream]" if us else " [NULL stream]"), us)(
                lambda r, e, a=(m, n, nb, Pr, Pc, kind): desc_compute(r, e, *a))
    for (n, nb, size, c, Pr) in [(1024, 128, 8, 2, 2), (1000, 128, 4, 1, 2), (768, 128, 8, 2, 1), (1024, 128, 4, 1, 1)]:
        mp_case("reference pieces -> %dx%d plan through descriptors -> pieces, n=%d nb=%d grid c=%d" % (Pr, size // Pr, n, nb, c))(
            lambda r, e, a=(n, nb, size, c, Pr): cyclic2d_compute(r, e, *a))
    for us in (0, 1):
        case("matrix utilities n=500%s" % (" [user stream]" if us else " [NULL stream]"), us)(lambda r, e: utils_compute(r, e, 500))
    # ---- against the REAL reference's dumps (tests/golden, produced by oracle/_ref): no oracle, no NumPy factorization in between
    for f in sorted(os.listdir(GOLD)):
        if f.startswith("cholinv_n") and f.endswith(".npz"):
            case("golden %s [NULL stream]" % f, 0)(lambda r, e, f=f: golden_cholinv_1rank(r, e, f))
        elif (f.startswith("cholinv_p8_n") or f.startswith("cholinv_grid8_n")) and f.endswith(".npz"):
            mp_case("golden %s (8 ranks, pieces)" % f)(lambda r, e, f=f: golden_cholinv_8ranks(r, e, f))
        elif f.startswith("cacqr") and ("_p8_" in f or "_p16_" in f or "_p27_" in f) and f.endswith(".npz"):
            mp_case("golden %s (%s ranks)" % (f, f.split("_p")[1].split("_")[0]))(lambda r, e, f=f: golden_cacqr(r, e, f))
        elif f.startswith("summa_c") and f.endswith(".npz"):
            mp_case("golden %s (the cube's ranks, pieces)" % f)(lambda r, e, f=f: golden_summa(r, e, f))
    for (size, c, m, n) in [(8, 2, 4096, 128), (4, 1, 4096, 64), (16, 2, 8192, 256), (8, 2, 1000, 64), (27, 3, 2700, 96)]:
        mp_case("cacqr grid size=%d c=%d m=%d n=%d" % (size, c, m, n))(lambda r, e, a=(size, c, m, n): cacqr_grid_compute(r, e, *a))
    for (m, n, iters, P) in [(4096, 256, 2, 1), (8192, 256, 2, 4), (4096, 128, 2, 4), (4096, 64, 1, 2), (6144, 256, 2, 3), (2048, 96, 2, 8)]:
        mp_case("cacqr m=%d n=%d iter=%d P=%d" % (m, n, iters, P))(lambda r, e, a=(m, n, iters, P): cacqr_compute(r, e, *a))
    for (n, ci, c, d, bc) in [(1024, 1, 2, 2, -2), (1000, 1, 2, 2, -2), (1024, 0, 1, 2, -2), (1024, 0, 2, 2, 0), (768, -1, 2, 1, -2), (1536, 0, 2, 2, -3),
                              (1003, 0, 2, 2, -2), (1001, 0, 1, 2, -3), (515, 1, 2, 2, -1)]:      # ragged: upstream cuts the LOCAL dimension
        mp_case("cholinv over the reference's layout n=%d ci=%d bc=%d grid %dx%dx%d" % (n, ci, bc, d, d, c))(lambda r, e, a=(n, ci, c, d, bc): cholinv_cyclic_compute(r, e, *a))
    # (the single-GPU plan's panel width is 1024 = K of its bf16 updates: at least 3 panels for an update to happen, 8 for the paired far update)
    # (N = 8192, where the paired far update starts, takes a minute on 8 host cores: SHIM_BIG=1)
    big = [(8192, 8, ()), (6144, 5, (("pair_rest", 0),))] if os.environ.get("SHIM_BIG") else []
    for (n, nrhs, opts) in [(4096, 8, ()), (4096, 8, (("strip", 1),)), (4096, 8, (("split", 0),)), (4096, 8, (("solve3", 0),)), (3072, 200, ()),
                            (4096, 5, (("reserve", 8),))] + big:
        case("mpchol n=%d nrhs=%d %s [NULL stream]" % (n, nrhs, dict(opts) or ""), 0)(lambda r, e, a=(n, nrhs, opts): mixed_compute(r, e, *a))
    case("mpchol n=3072 nrhs=8 twice [user stream]", 1)(lambda r, e: mixed_compute(r, e, 3072, 8, reps=2))
    for (n, nb, P) in [(1024, 256, 1), (1024, 256, 4), (1280, 256, 4), (2048, 256, 8), (1152, 128, 3), (1024, 128, 2)]:
        mp_case("dmp n=%d nb=%d P=%d" % (n, nb, P))(lambda r, e, a=(n, nb, P): dmp_compute(r, e, *a))
    json.dump({"results": RESULTS}, open(out_path, "w"), indent=1)
    bad = [x for x in RESULTS if x["findings"]]
    print("%d cases, %d with findings" % (len(RESULTS), len(bad)))
    for x in bad[:40]:
        print(" *", x["name"], {k: "%.2e" % v for k, v in x["errors"].items()})
        for f in x["findings"][:6]:
            print("     ", f[:300])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(rs.build_shim.OUT, "compute.json"))
