"""TEST INFRASTRUCTURE: drives the library's host side (the product's object files linked against the recording HIP / RCCL stand-in,
tests/hipshim/) through its C ABI on a machine without a GPU and checks every trace with trace_check.py.

Runs in its OWN process (no torch: torch brings a real libamdhip64 into the process):
    python tests/hipshim/run_scenarios.py out.json
One entry per scenario: {"name", "findings": [...], "stats": {...}}.  A multi-rank plan is driven once per simulated rank with a
callback communicator (cap_comm_create_callbacks) whose collectives are traced operations on the stream they are handed: the
schedule a rank enqueues only depends on (rank, size), not on what its peers send."""
import ctypes as C
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import build_shim       # noqa: E402
import trace_check      # noqa: E402


def signatures():
    src = open(os.path.join(ROOT, "capital_amd", "_lib.py")).read().replace("import torch", "pass")
    ns = {"__file__": os.path.join(ROOT, "capital_amd", "_lib.py"), "__name__": "sig"}
    exec(compile(src, "_lib.py", "exec"), ns)
    return ns["SIGNATURES"]


LIBP, SHIMP = build_shim.build()
shim = C.CDLL(SHIMP, mode=C.RTLD_GLOBAL)
L = C.CDLL(LIBP, mode=os.RTLD_LOCAL | os.RTLD_DEEPBIND)
for name, (res, args) in signatures().items():
    f = getattr(L, name)
    f.restype, f.argtypes = res, args
shim.shim_oob.restype = C.c_longlong
shim.shim_live_allocations.restype = C.c_longlong
shim.shim_note_op.argtypes = [C.c_char_p, C.c_void_p]
shim.shim_mark.argtypes = [C.c_char_p]
shim.shim_dump.argtypes = [C.c_char_p]
shim.shim_live_report.argtypes = [C.c_char_p]
shim.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
shim.hipFree.argtypes = [C.c_void_p]
shim.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
shim.hipStreamDestroy.argtypes = [C.c_void_p]
shim.shim_access_note.argtypes = [C.c_int, C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong, C.c_int, C.c_int]
shim.shim_access_note.restype = None
# the library's access hook (capital_amd/csrc/common.h): every launch now carries the windows it reads and writes into the trace
C.c_void_p.in_dll(L, "cap_access_hook").value = C.cast(shim.shim_access_note, C.c_void_p).value


def acc(mode, ptr, nbytes):
    if ptr and nbytes > 0:
        shim.shim_access_note(mode, ptr, 0, int(nbytes), 1, 0, 1)


_AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
_BC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)
_AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
_A2A = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64),
                   C.POINTER(C.c_int64), C.c_void_p)
_KEEP = []


def ok(st, what):
    if st != 0:
        raise RuntimeError("%s returned %d (%s)" % (what, st, (L.cap_status_string(st) or b"?").decode()))


ALIASING = []                # violations of the collectives' buffer-aliasing rules seen by the callbacks of the running scenario
IPC_MAGIC = (0x4c444e4148435049).to_bytes(8, "little")      # hipshim.cpp: bytes 8 .. 15 of a handle the stand-in exported


class Comm:
    """callback communicator of one simulated rank: every collective is one traced operation on its stream, labelled with the
    communicator it belongs to ("OP stream kind label size me count root") so that the joint replay of all ranks' traces can match
    the k-th collective of a communicator across its ranks (trace_check.check_joint)"""

    def __init__(self, rank, size, label="world", members=None):
        """members: the WORLD ranks of this communicator's ranks, in its own order (default: it is the world)"""
        self.rank, self.size, self.label = rank, size, label
        members = list(members) if members is not None else list(range(size))

        def op(kind, count, root, st):
            shim.shim_note_op(("%s %s %d %d %d %d" % (kind, label, size, rank, count, root)).encode(), st)

        def ag_(ctx, s, r, n, st):
            # small payloads (IPC handles, status words) are really gathered - every slot gets MY piece; a handle the stand-in
            # exported is tagged with the slot it lands in, so a rank maps "the same buffer of peer q" (a range of its own in the
            # stand-in, resolved to peer q's allocation by the joint replay) and the IPC schedules run their real course.
            # Big payloads are only traced.
            # ncclAllGather's aliasing rule: the send buffer is either disjoint from the receive buffer or exactly my slot of it
            if s and r and n > 0 and s < r + n * 8 * size and r < s + n * 8 and s != r + rank * n * 8:
                ALIASING.append("allgather on %s: send buffer %#x overlaps the receive buffer %#x (%d x %d bytes) but is not slot %d of it" % (label, s, r, size, n * 8, rank))
            acc(1, s, n * 8); acc(2, r, n * 8 * size)
            op("allgather", n, -1, st)
            if 0 < n * 8 <= 4096 and s and r:
                for q in range(size):
                    C.memmove(r + q * n * 8, s, n * 8)
                    for h in range(0, n * 8 - 63, 64):
                        if C.string_at(r + q * n * 8 + h + 8, 8) == IPC_MAGIC:
                            C.memmove(r + q * n * 8 + h + 16, (members[q] + 1).to_bytes(4, "little"), 4)
            return 0
        ag = _AG(ag_)
        bc = _BC(lambda ctx, b, n, root, st: (acc(1 if root == rank else 2, b, n * 8), op("bcast", n, root, st), 0)[2])
        ar = _AR(lambda ctx, b, n, st: (acc(3, b, n * 8), op("allreduce", n, -1, st), 0)[2])

        def a2a_(ctx, s, sc, sd, r, rc, rd, st):
            for q in range(size):
                acc(1, (s or 0) + 8 * sd[q] if s else None, 8 * sc[q]); acc(2, (r or 0) + 8 * rd[q] if r else None, 8 * rc[q])
                for q2 in range(size):      # no piece that is sent may overlap a piece that is received (ncclSend / ncclRecv pairs in one group)
                    a0, a1, b0, b1 = (s or 0) + 8 * sd[q], (s or 0) + 8 * (sd[q] + sc[q]), (r or 0) + 8 * rd[q2], (r or 0) + 8 * (rd[q2] + rc[q2])
                    if s and r and sc[q] > 0 and rc[q2] > 0 and a0 < b1 and b0 < a1:
                        ALIASING.append("alltoallv on %s: the piece sent to %d overlaps the piece received from %d" % (label, q, q2))
            # pairwise exchanges (ncclSend / ncclRecv pairs in the library; ranks without a partner do not call): a local operation
            shim.shim_note_op(b"alltoallv", st)
            return 0
        a2a = _A2A(a2a_)
        _KEEP.extend([ag, bc, ar, a2a])
        self.handle = C.c_void_p()
        ok(L.cap_comm_create_callbacks(C.byref(self.handle), rank, size, ag, bc, ar, None), "cap_comm_create_callbacks")
        ok(L.cap_comm_set_alltoallv_callback(self.handle, a2a), "cap_comm_set_alltoallv_callback")

    def close(self):
        L.cap_comm_destroy(self.handle)


def dmalloc(nbytes):
    p = C.c_void_p()
    assert shim.hipMalloc(C.byref(p), max(int(nbytes), 8)) == 0
    return p


class Run:
    """one scenario: a fresh trace, a user stream (0 = the NULL stream, else a non-blocking stream of the test's own), marked calls"""

    def __init__(self, name, user_stream):
        self.name = name
        shim.shim_reset()
        shim.shim_mark(("scenario %s" % name).encode())
        self.stream = C.c_void_p(0)
        if user_stream:
            ok(shim.hipStreamCreateWithFlags(C.byref(self.stream), 1), "hipStreamCreateWithFlags")
        self.uid = 1 if user_stream else 0          # (the stand-in numbers streams from 1 after every process start - see finish())

    def call(self, what, fn, *args):
        shim.shim_mark(("begin %s" % what).encode())
        st = fn(*args)
        shim.shim_mark(("end %s user=@" % what).encode())
        return st

    def finish(self):
        path = os.path.join(build_shim.OUT, "trace_%d.txt" % os.getpid())
        shim.shim_dump(path.encode())
        lines = open(path).read().splitlines()
        os.unlink(path)
        # the user's stream is the first STREAM line of the trace when the scenario made one
        uid = 0
        if self.stream.value:
            uid = int([l for l in lines if l.startswith("STREAM ")][0].split()[1])
        lines = [l.replace("user=@", "user=%d" % uid) for l in lines]
        if KEEP_TRACE:
            os.makedirs(KEEP_TRACE, exist_ok=True)
            open(os.path.join(KEEP_TRACE, "".join(ch if ch.isalnum() else "_" for ch in self.name)[:150] + ".txt"), "w").write("\n".join(lines) + "\n")
        self.lines = lines
        findings, stats = trace_check.check(lines)
        findings += ALIASING
        del ALIASING[:]
        # kernel launches of the LAST marked "...factor" call, by (mangled) kernel name: compared with rocprofv3's counts of the same
        # schedule on the GPU by tests/test_schedule_structure.py
        hist, cur = {}, None
        for l in lines:
            if l.startswith("MARK begin") and "factor" in l:
                cur = {}
            elif l.startswith("MARK end") and cur is not None:
                hist, cur = cur, None
            elif cur is not None and l.startswith("K "):
                k = l.split()[2]; cur[k] = cur.get(k, 0) + 1
        stats["factor_kernels"] = hist
        if self.stream.value:
            shim.hipStreamDestroy(self.stream)
        stats["oob"] = int(shim.shim_oob())
        return {"name": self.name, "findings": findings, "stats": stats}


RESULTS = []
FILTER = os.environ.get("SHIM_FILTER", "")          # run only the scenarios whose name contains this (debugging)
KEEP_TRACE = os.environ.get("SHIM_KEEP_TRACE", "")  # directory: keep every scenario's trace there


GROUPS = {}                  # (group name, user stream) -> {rank: trace}: the ranks of one multi-rank configuration


def scenario(name, user_stream, group=None, rank=0, nranks=1):
    """group / rank / nranks: this scenario is rank `rank` of the `nranks` simulated ranks of configuration `group`; once all of them
    have run, their traces are replayed TOGETHER (trace_check.check_joint: collectives matched across the ranks, peer copies of the IPC
    exchanges checked against the peer's own kernels) and the outcome is one more result, "<group> [joint replay of N ranks]"."""
    def deco(fn):
        if FILTER and FILTER not in name:
            return fn
        r = Run(name + (" [user stream]" if user_stream else " [NULL stream]"), user_stream)
        try:
            fn(r)
            out = r.finish()
        except Exception as e:      # a refused configuration or a crash of the host side is a finding too
            out = {"name": r.name, "findings": ["exception: %r" % (e,)], "stats": {}}
            r.lines = None
        RESULTS.append(out)
        if group is not None and nranks > 1:
            g = GROUPS.setdefault((group, user_stream), {})
            g[rank] = r.lines
            if len(g) == nranks:
                nm = "%s [joint replay of %d ranks, %s]" % (group, nranks, "user stream" if user_stream else "NULL stream")
                if any(v is None for v in g.values()):
                    RESULTS.append({"name": nm, "findings": ["a rank's scenario did not run"], "stats": {}})
                else:
                    f, st = trace_check.check_joint([g[q] for q in range(nranks)])
                    RESULTS.append({"name": nm, "findings": f, "stats": st})
                del GROUPS[(group, user_stream)]
        return fn
    return deco


def cholinv_case(r, n, ci, split, bc, opts=(), reps=2, comm=None, local_cols=None):
    plan = C.c_void_p()
    ok(L.cap_cholinv_plan_create(C.byref(plan), n, ci, split, bc, b"U", comm), "cap_cholinv_plan_create")
    for k, v in opts:
        ok(L.cap_cholinv_set_option(plan, k.encode(), v), "set_option " + k)
    cols = n if local_cols is None else max(local_cols, 1)
    A = dmalloc(8 * n * cols); out = dmalloc(8 * n * cols)
    info = C.c_int64(0)
    for _ in range(reps):
        ok(r.call("cholinv_factor", L.cap_cholinv_factor, plan, A, n, r.stream), "cap_cholinv_factor")
    r.call("cholinv_info", L.cap_cholinv_info, plan, r.stream, C.byref(info))
    ok(r.call("cholinv_get_R", L.cap_cholinv_get_R, plan, out, n, r.stream), "cap_cholinv_get_R")
    if ci >= 0:
        ok(r.call("cholinv_get_Rinv", L.cap_cholinv_get_Rinv, plan, out, n, r.stream), "cap_cholinv_get_Rinv")
    ok(L.cap_cholinv_plan_destroy(plan), "cap_cholinv_plan_destroy")
    shim.hipFree(A); shim.hipFree(out)


def local_cols_1d(n, nb, P, p):
    nblk = (n + nb - 1) // nb
    return sum(min(nb, n - j * nb) for j in range(p, nblk, P))


def dist_case(r, n, nb, P, p, opts=(), ci=-1):
    comm = Comm(p, P)
    plan = C.c_void_p()
    ok(L.cap_dist_plan_create(C.byref(plan), n, nb, comm.handle), "cap_dist_plan_create")
    for k, v in opts:
        ok(L.cap_dist_set_option(plan, k.encode(), v), "dist set_option " + k)
    if ci >= 0:
        ok(L.cap_dist_set_option(plan, b"complete_inv", ci), "dist complete_inv")
    lc = int(L.cap_dist_local_cols(plan))
    A = dmalloc(8 * n * max(lc, 1)); out = dmalloc(8 * n * max(lc, 1))
    info = C.c_int64(0)
    for _ in range(2):
        ok(r.call("dist_factor", L.cap_dist_factor, plan, A, n, r.stream), "cap_dist_factor")
    r.call("dist_info", L.cap_dist_info, plan, r.stream, C.byref(info))
    if dict(opts).get("ipc") and P > 1 and int(L.cap_dist_get_option(plan, b"ipc_active")) != 1:
        raise RuntimeError("the IPC strip exchange was asked for but is not active (the stand-in's peer mapping failed)")
    ok(r.call("dist_get_R", L.cap_dist_get_R, plan, out, n, r.stream), "cap_dist_get_R")
    if ci >= 0:
        ok(r.call("dist_get_Rinv", L.cap_dist_get_Rinv, plan, out, n, r.stream), "cap_dist_get_Rinv")
    ok(L.cap_dist_plan_destroy(plan), "cap_dist_plan_destroy")
    comm.close(); shim.hipFree(A); shim.hipFree(out)


def dist2d_case(r, n, nb, Pr, Pc, pr, pc, opts=()):
    world = Comm(pr * Pc + pc, Pr * Pc); row = Comm(pc, Pc, "row%d" % pr, [pr * Pc + q for q in range(Pc)]); col = Comm(pr, Pr, "col%d" % pc, [q * Pc + pc for q in range(Pr)])
    plan = C.c_void_p()
    ok(L.cap_dist2d_plan_create(C.byref(plan), n, nb, world.handle, Pr, row.handle, col.handle), "cap_dist2d_plan_create")
    for k, v in opts:
        ok(L.cap_dist2d_set_option(plan, k.encode(), v), "dist2d set_option " + k)
    lr, lc = int(L.cap_dist2d_get(plan, 0)), int(L.cap_dist2d_get(plan, 1))
    A = dmalloc(8 * max(lr, 1) * max(lc, 1)); out = dmalloc(8 * max(lr, 1) * max(lc, 1))
    info = C.c_int64(0)
    for _ in range(2):
        ok(r.call("dist2d_factor", L.cap_dist2d_factor, plan, A, max(lr, 1), r.stream), "cap_dist2d_factor")
    r.call("dist2d_info", L.cap_dist2d_info, plan, r.stream, C.byref(info))
    if dict(opts).get("ipc") and int(L.cap_dist2d_get(plan, 12)) != 1:
        raise RuntimeError("the IPC operand moves were asked for but are not active (the stand-in's peer mapping failed)")
    ok(r.call("dist2d_get_R", L.cap_dist2d_get_R, plan, out, max(lr, 1), r.stream), "cap_dist2d_get_R")
    if dict(opts).get("complete_inv", -1) >= 0:
        ok(r.call("dist2d_get_Rinv", L.cap_dist2d_get_Rinv, plan, out, max(lr, 1), r.stream), "cap_dist2d_get_Rinv")
    ok(L.cap_dist2d_plan_destroy(plan), "cap_dist2d_plan_destroy")
    for c in (world, row, col):
        c.close()
    shim.hipFree(A); shim.hipFree(out)


def mpchol_case(r, n, nrhs, opts=()):
    plan = C.c_void_p()
    ok(L.cap_mpchol_plan_create(C.byref(plan), n, nrhs), "cap_mpchol_plan_create")
    for k, v in opts:
        ok(L.cap_mpchol_set_option(plan, k.encode(), v), "mpchol set_option " + k)
    A = dmalloc(8 * n * n); B = dmalloc(8 * n * nrhs); X = dmalloc(8 * n * nrhs)
    info = C.c_int64(0); it = C.c_int(0); rr = C.c_double(0)
    for _ in range(2):
        ok(r.call("mpchol_factor", L.cap_mpchol_factor, plan, A, n, r.stream), "cap_mpchol_factor")
    r.call("mpchol_info", L.cap_mpchol_info, plan, r.stream, C.byref(info))
    ok(r.call("mpchol_solve", L.cap_mpchol_solve, plan, A, n, B, n, X, n, nrhs, 3, 1e-15, C.byref(it), C.byref(rr), r.stream), "cap_mpchol_solve")
    ok(L.cap_mpchol_plan_destroy(plan), "cap_mpchol_plan_destroy")
    for q in (A, B, X):
        shim.hipFree(q)


def dmp_case(r, n, nb, P, p, nrhs=5):
    comm = Comm(p, P)
    plan = C.c_void_p()
    ok(L.cap_dmp_plan_create(C.byref(plan), n, nb, nrhs, comm.handle), "cap_dmp_plan_create")
    lc = int(L.cap_dmp_local_cols(plan))
    A = dmalloc(8 * n * max(lc, 1)); B = dmalloc(8 * n * nrhs); X = dmalloc(8 * n * nrhs)
    info = C.c_int64(0); it = C.c_int(0); rr = C.c_double(0)
    ok(r.call("dmp_factor (first call of the plan)", L.cap_dmp_factor, plan, A, n, r.stream), "cap_dmp_factor")
    r.call("dmp_info", L.cap_dmp_info, plan, r.stream, C.byref(info))
    ok(r.call("dmp_factor", L.cap_dmp_factor, plan, A, n, r.stream), "cap_dmp_factor")
    ok(r.call("dmp_solve", L.cap_dmp_solve, plan, A, n, B, n, X, n, nrhs, 3, 1e-15, C.byref(it), C.byref(rr), r.stream), "cap_dmp_solve")
    ok(L.cap_dmp_plan_destroy(plan), "cap_dmp_plan_destroy")
    comm.close()
    for q in (A, B, X):
        shim.hipFree(q)


def cacqr_case(r, m, n, iters, P, p):
    comm = Comm(p, P)
    plan = C.c_void_p()
    ok(L.cap_cacqr_plan_create(C.byref(plan), m, n, iters, comm.handle), "cap_cacqr_plan_create")
    A = dmalloc(8 * m * n)
    info = C.c_int64(0)
    for _ in range(2):
        ok(r.call("cacqr_factor", L.cap_cacqr_factor, plan, A, m, r.stream), "cap_cacqr_factor")
    r.call("cacqr_info", L.cap_cacqr_info, plan, r.stream, C.byref(info))
    ok(L.cap_cacqr_plan_destroy(plan), "cap_cacqr_plan_destroy")
    comm.close(); shim.hipFree(A)


def group_of(color_of, key_of, size, rank):
    ranks = sorted((q for q in range(size) if color_of(q) == color_of(rank)), key=key_of)
    return ranks.index(rank), len(ranks), ranks


class Topo:
    """topo::square (kind 0) / topo::rect (kind 1) bundle of one simulated rank over callback communicators: the sub-groups are the
    ones capital_amd/topo.py assembles for the host-staged runs (topology.h:16-143)"""

    def __init__(self, kind, rank, size, c, num_chunks=0):
        d, x, y, z = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        ok(L.cap_topo_coords(kind, rank, size, c, C.byref(d), C.byref(x), C.byref(y), C.byref(z)), "cap_topo_coords")

        def co(q):
            dd, xx, yy, zz = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
            L.cap_topo_coords(kind, q, size, c, C.byref(dd), C.byref(xx), C.byref(yy), C.byref(zz))
            return xx.value, yy.value, zz.value
        if kind == 0:
            splits = [(lambda q: (co(q)[1], co(q)[2]), lambda q: co(q)[0]), (lambda q: (co(q)[0], co(q)[2]), lambda q: co(q)[1]),
                      (lambda q: q // c, lambda q: q), (lambda q: co(q)[2], lambda q: q), None, None, None]
        else:
            cube, sl = c * c * c, c * c
            splits = [(lambda q: (q // cube, ((q % cube) % c) + c * ((q % cube) // sl)), lambda q: q % cube), None,
                      (lambda q: (q // cube, (q % cube) // c), lambda q: q % cube), (lambda q: q % c, lambda q: q),
                      (lambda q: (q % sl, (q // sl) // c), lambda q: q // sl), (lambda q: (q % sl, (q // sl) % c), lambda q: q // sl),
                      (lambda q: q // cube, lambda q: q)]
        self.world = Comm(rank, size)
        self.subs = []
        arr = (C.c_void_p * 7)()
        for i, sp in enumerate(splits):
            if sp is None:
                continue
            me, n, members = group_of(sp[0], sp[1], size, rank)
            cm = Comm(me, n, "sub%d:%s" % (i, str(sp[0](rank)).replace(" ", "")), members)
            self.subs.append(cm); arr[i] = cm.handle
        self.handle = C.c_void_p()
        ok(L.cap_topo_create_from(C.byref(self.handle), kind, self.world.handle, c, 0, num_chunks, arr, 7), "cap_topo_create_from")

    def close(self):
        L.cap_topo_destroy(self.handle)
        for cm in self.subs + [self.world]:
            cm.close()


def summa_case(r, size, c, rank, M, N, K, chunks):
    t = Topo(0, rank, size, c, chunks)
    plan = C.c_void_p()
    ok(L.cap_summa_plan_create(C.byref(plan), t.handle, M, N, K, chunks), "cap_summa_plan_create")
    ml, nl, kl = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    L.cap_summa_local_dims(plan, C.byref(ml), C.byref(nl), C.byref(kl))
    ml, nl, kl = max(ml.value, 1), max(nl.value, 1), max(kl.value, 1)
    A = dmalloc(8 * ml * kl); B = dmalloc(8 * kl * nl); Cc = dmalloc(8 * ml * nl)
    for _ in range(2):
        ok(r.call("summa_dgemm", L.cap_summa_dgemm, plan, 1.0, A, ml, B, kl, 0.5, Cc, ml, r.stream), "cap_summa_dgemm")
    ok(L.cap_summa_plan_destroy(plan), "cap_summa_plan_destroy")
    if M == K:      # the TRMM overload (T m x m on the left) and the SYRK overload on plans of their own shapes
        p2 = C.c_void_p()
        ok(L.cap_summa_plan_create(C.byref(p2), t.handle, M, N, M, chunks), "cap_summa_plan_create")
        T = dmalloc(8 * ml * ml); tmp = dmalloc(8 * ml * ml)
        ok(r.call("util_transpose", L.cap_util_transpose, t.handle, T, tmp, ml * ml, r.stream), "cap_util_transpose")
        ok(r.call("summa_dtrmm", L.cap_summa_dtrmm, p2, 0, 1, 1, 0, 1.0, T, ml, 0, B, ml, r.stream), "cap_summa_dtrmm")
        ok(L.cap_summa_plan_destroy(p2), "cap_summa_plan_destroy")
        p3 = C.c_void_p()
        ok(L.cap_summa_plan_create(C.byref(p3), t.handle, N, N, K, chunks), "cap_summa_plan_create")
        Cs = dmalloc(8 * nl * nl)
        ok(r.call("summa_dsyrk", L.cap_summa_dsyrk, p3, 1, 1, -1.0, B, kl, 1.0, Cs, nl, 0, r.stream), "cap_summa_dsyrk")
        ok(L.cap_summa_plan_destroy(p3), "cap_summa_plan_destroy")
        for q in (T, tmp, Cs):
            shim.hipFree(q)
    t.close()
    for q in (A, B, Cc):
        shim.hipFree(q)


def cacqr_grid_case(r, size, c, rank, m, n, iters):
    t = Topo(1, rank, size, c)
    plan = C.c_void_p()
    ok(L.cap_cacqr_plan_create_grid(C.byref(plan), m, n, iters, t.handle), "cap_cacqr_plan_create_grid")
    d = size // (c * c)
    ml, nl = (m + d - 1) // d, max(n // c, 1)
    A = dmalloc(8 * ml * nl); out = dmalloc(8 * n * n)
    info = C.c_int64(0)
    for _ in range(2):
        ok(r.call("cacqr_factor (grid)", L.cap_cacqr_factor, plan, A, ml, r.stream), "cap_cacqr_factor")
    r.call("cacqr_info", L.cap_cacqr_info, plan, r.stream, C.byref(info))
    ok(r.call("cacqr_R_piece", L.cap_cacqr_R_piece, plan, out, max(n // c, 1), r.stream), "cap_cacqr_R_piece")
    ok(L.cap_cacqr_plan_destroy(plan), "cap_cacqr_plan_destroy")
    t.close(); shim.hipFree(A); shim.hipFree(out)


def redist_case(r, n, nb, size, c, Pr, rank):
    world = Comm(rank, size)
    plan = C.c_void_p()
    ok(L.cap_redist_plan_create(C.byref(plan), n, nb, world.handle, c, Pr), "cap_redist_plan_create")
    e, lr, lc = (int(L.cap_redist_get(plan, w)) for w in (0, 1, 2))
    piece = dmalloc(8 * max(e, 1) * max(e, 1)); bc = dmalloc(8 * max(lr, 1) * max(lc, 1))
    ok(r.call("redistribute_cyclic_to_bc", L.cap_redistribute_cyclic_to_bc, plan, piece, max(e, 1), bc, max(lr, 1), r.stream), "cyclic_to_bc")
    ok(r.call("redistribute_bc_to_cyclic", L.cap_redistribute_bc_to_cyclic, plan, bc, max(lr, 1), piece, max(e, 1), r.stream), "bc_to_cyclic")
    ok(L.cap_redist_plan_destroy(plan), "cap_redist_plan_destroy")
    world.close(); shim.hipFree(piece); shim.hipFree(bc)


def desc_case(r, n, nb, Pr, Pc, pr, pc):
    """host matrix -> my block-cyclic piece -> host, through the pinned staging buffers (host memory of the test: a plain bytearray)"""
    d = C.c_void_p()
    ok(L.cap_desc_create_bc(C.byref(d), n, n, nb, Pr, Pc, pr, pc, None, 0), "cap_desc_create_bc")
    host = (C.c_double * (n * n))()
    ok(r.call("desc_import_host_global", L.cap_desc_import_host_global, d, host, n, r.stream), "cap_desc_import_host_global")
    ok(r.call("desc_export_host_global", L.cap_desc_export_host_global, d, host, n, r.stream), "cap_desc_export_host_global")
    lr, lc = int(L.cap_desc_get(d, 3)), int(L.cap_desc_get(d, 2))
    piece = (C.c_double * max(lr * lc, 1))()
    ok(r.call("desc_import_host", L.cap_desc_import_host, d, piece, max(lr, 1), r.stream), "cap_desc_import_host")
    ok(r.call("desc_import_host (again, the first one still in flight)", L.cap_desc_import_host, d, piece, max(lr, 1), r.stream), "cap_desc_import_host")
    ok(r.call("desc_export_host", L.cap_desc_export_host, d, piece, max(lr, 1), r.stream), "cap_desc_export_host")
    ok(r.call("desc_import_host_global (behind an export)", L.cap_desc_import_host_global, d, host, n, r.stream), "cap_desc_import_host_global")
    ok(L.cap_desc_destroy(d), "cap_desc_destroy")


def operators_case(r, m, n, k):
    """the blas / lapack seam on the caller's stream (cap_dgemm .. cap_dtrtri)"""
    A = dmalloc(8 * max(m, k) * max(m, k)); B = dmalloc(8 * max(k, m) * n); Cc = dmalloc(8 * m * max(n, m))
    ok(r.call("dgemm NN", L.cap_dgemm, 0, 0, m, n, k, 1.0, A, m, B, k, 0.0, Cc, m, r.stream), "cap_dgemm")
    ok(r.call("dgemm TN beta=1", L.cap_dgemm, 1, 0, m, n, k, -1.0, A, k, B, k, 1.0, Cc, m, r.stream), "cap_dgemm")
    ok(r.call("dsyrk", L.cap_dsyrk, 1, 1, m, k, -1.0, A, k, 1.0, Cc, m, r.stream), "cap_dsyrk")
    wt = int(L.cap_dtrmm_work_size(0, m, n)); W = dmalloc(8 * max(wt, 1))
    ok(r.call("dtrmm", L.cap_dtrmm, 0, 1, 0, 0, m, n, 1.0, A, m, B, m, W, r.stream), "cap_dtrmm")
    ws = int(L.cap_dtrsm_work_size(0, m, n)); W2 = dmalloc(8 * max(ws, 1))
    ok(r.call("dtrsm", L.cap_dtrsm, 0, 1, 1, m, n, 1.0, A, m, B, m, W2, r.stream), "cap_dtrsm")
    wp = int(L.cap_dpotrf_work_size(m)); W3 = dmalloc(8 * max(wp, 1)); info = dmalloc(8)
    ok(r.call("dpotrf", L.cap_dpotrf, 1, m, A, m, info, W3, r.stream), "cap_dpotrf")
    wi = int(L.cap_dtrtri_work_size(m)); W4 = dmalloc(8 * max(wi, 1))
    ok(r.call("dtrtri", L.cap_dtrtri, 1, m, A, m, W4, r.stream), "cap_dtrtri")
    for q in (A, B, Cc, W, W2, W3, W4, info):
        shim.hipFree(q)


def lifecycle_case(r):
    """plans that are created and destroyed without use, reconfigured between calls, refused, reused with another size of right-hand
    sides: nothing may dangle, leak or be destroyed twice"""
    for (n, ci) in [(2048, -1), (2048, 1), (1000, 0)]:
        plan = C.c_void_p()
        ok(L.cap_cholinv_plan_create(C.byref(plan), n, ci, 1, -2, b"U", None), "cap_cholinv_plan_create")
        ok(L.cap_cholinv_plan_destroy(plan), "destroy unused")
    plan = C.c_void_p()
    ok(L.cap_cholinv_plan_create(C.byref(plan), 8192, 1, 1, 0, b"U", None), "cap_cholinv_plan_create")
    A = dmalloc(8 * 8192 * 8192); out = dmalloc(8 * 8192 * 8192)
    ok(r.call("cholinv_factor", L.cap_cholinv_factor, plan, A, 8192, r.stream), "factor")
    for k, v in [("nb", 256), ("use_sb", 0), ("inv_overlap", 0), ("nb", 1024), ("use_sb", 1), ("reserve", 8), ("reserve", 0), ("inner_la", 1), ("inner_la", 0),
                 ("inv_fast", 0), ("inv_fast", 1), ("chain_coop", 0), ("chain_coop", -1), ("profile", 1)]:
        ok(L.cap_cholinv_set_option(plan, k.encode(), v), "set_option " + k)
        ok(r.call("cholinv_factor after %s=%d" % (k, v), L.cap_cholinv_factor, plan, A, 8192, r.stream), "factor after " + k)
        ok(r.call("cholinv_get_Rinv", L.cap_cholinv_get_Rinv, plan, out, 8192, r.stream), "get_Rinv")
    ok(L.cap_cholinv_plan_destroy(plan), "cap_cholinv_plan_destroy")
    # refused configurations leave nothing behind
    bad = C.c_void_p()
    assert L.cap_cholinv_plan_create(C.byref(bad), 1024, 1, 1, 0, b"L", None) != 0
    assert L.cap_cholinv_plan_create(C.byref(bad), -5, 1, 1, 0, b"U", None) != 0
    assert L.cap_mpchol_plan_create(C.byref(bad), 1000, 8) != 0                   # n % 128
    cm = Comm(0, 4)
    assert L.cap_dmp_plan_create(C.byref(bad), 1000, 256, 5, cm.handle) != 0
    assert L.cap_dist2d_plan_create(C.byref(bad), 1536, 128, cm.handle, 3, None, None) != 0      # 3 does not divide 4
    assert L.cap_dist_factor(None, A, 8192, r.stream) != 0
    cm.close()
    # the 1 x P plan: option changes between factor calls
    cm = Comm(1, 4)
    dp = C.c_void_p()
    ok(L.cap_dist_plan_create(C.byref(dp), 4096, 128, cm.handle), "cap_dist_plan_create")
    lc = int(L.cap_dist_local_cols(dp))
    Al = dmalloc(8 * 4096 * lc)
    for k, v in [("safe", 1), ("safe", 0), ("ipc", 1), ("ipc", 0), ("strip", 1), ("strip", 2), ("depth2", 0), ("complete_inv", 1), ("complete_inv", -1)]:
        ok(L.cap_dist_set_option(dp, k.encode(), v), "dist set_option " + k)
        ok(r.call("dist_factor after %s=%d" % (k, v), L.cap_dist_factor, dp, Al, 4096, r.stream), "dist factor after " + k)
    ok(L.cap_dist_plan_destroy(dp), "cap_dist_plan_destroy")
    cm.close()
    for q in (A, out, Al):
        shim.hipFree(q)


def cblas_case(r, what, m, n, k):
    """include/capital_amd_cblas.h over the stand-in: host memory in, staged through "device" buffers, the operator, the window back - the
    staging copies (synchronous, NULL stream) against every stream the operator uses inside"""
    import numpy as np
    CB = C.CDLL(build_shim.build_cblas())
    d = C.c_double
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    if what == "dgemm":
        a, b, c = (np.asfortranarray(np.ones(sh)) for sh in ((m + 1, k), (k + 3, n), (m, n)))
        for beta in (0.0, 1.0):
            r.call("cblas_dgemm", CB.cblas_dgemm, 102, 111, 111, m, n, k, d(1.0), p(a), m + 1, p(b), k + 3, d(beta), p(c), m)
        r.call("cblas_dgemm", CB.cblas_dgemm, 102, 112, 112, m, n, k, d(1.0), p(np.asfortranarray(np.ones((k, m)))), k, p(np.asfortranarray(np.ones((n, k)))), n, d(0.0), p(c), m)
    elif what == "dtrmm":
        for side, t in ((141, m), (142, n)):
            tm, b = np.asfortranarray(np.triu(np.ones((t, t)))), np.asfortranarray(np.ones((m + 2, n)))
            for tr in (111, 112):
                r.call("cblas_dtrmm", CB.cblas_dtrmm, 102, side, 121, tr, 131, m, n, d(1.0), p(tm), t, p(b), m + 2)
    elif what == "dsyrk":
        a, c = np.asfortranarray(np.ones((k, n))), np.asfortranarray(np.ones((n, n)))
        r.call("cblas_dsyrk", CB.cblas_dsyrk, 102, 121, 112, n, k, d(-1.0), p(a), k, d(1.0), p(c), n)
    else:
        a = np.asfortranarray(np.eye(n) * n + 1.0)
        CB.LAPACKE_dpotrf.restype = C.c_int; CB.LAPACKE_dtrtri.restype = C.c_int
        for _ in range(2):
            r.call("LAPACKE_dpotrf", CB.LAPACKE_dpotrf, 102, C.c_char(b"U"), n, p(a), n)
        r.call("LAPACKE_dtrtri", CB.LAPACKE_dtrtri, 102, C.c_char(b"U"), C.c_char(b"N"), min(n, 3000), p(a), n)
    CB.capcb_release()


def main(out_path, user_streams=(0, 1)):
    for us in user_streams:
        # ---- single-GPU Cholesky plan: the headline schedule, reference semantics, ragged sizes, schedule options
        for (n, ci, split, bc, opts) in [
            (4096, -1, 1, -3, ()), (8192, -1, 1, 0, ()), (4097, -1, 1, -3, ()), (1000, -1, 1, -2, ()), (64, -1, 1, 0, ()),
            (65536, -1, 1, 0, ()),                                             # BASELINE's headline size: plan + schedule only, nothing computed
            (65536, -1, 1, -5, ()), (32768, -1, 1, -5, ()),                    # ... with bench.py's knobs (bcMult -5): launch counts vs rocprofv3
            (32768, 0, 1, 0, ()), (16384, 1, 1, 0, ()), (4096, 0, 2, -2, ()), (4096, 1, 1, -2, (("inv_overlap", 0),)), (3000, 1, 1, -2, ()),
            (8192, -1, 1, 0, (("use_sb", 0),)), (8192, -1, 1, 0, (("pair_rest", 0),)), (8192, -1, 1, 0, (("depth2", 0),)),
            (8192, -1, 1, 0, (("chain_coop", 0),)), (8192, -1, 1, 0, (("inner_la", 1),)), (8192, -1, 1, 0, (("serial_m", 4096),)),
            (8192, -1, 1, 0, (("reserve", 8),)), (16384, -1, 1, 0, (("reserve", 8), ("reserve_m", 8192))), (8192, 1, 1, 0, (("inv_fast", 0),)),
            # round 5, last session: option mixes no GPU test runs - reference semantics without strip buffers / with the tree started at
            # once / on masked streams, the paired far update at sizes where pairs and single strips alternate, ragged sizes with R^-1
            (16384, 1, 1, 0, (("use_sb", 0),)), (16384, 0, 2, 0, (("inv_start_m", 1 << 30),)), (16384, 1, 1, 0, (("reserve", 8),)),
            (24576, -1, 1, 0, ()), (24576, -1, 1, 0, (("use_sb", 0),)), (28672, -1, 1, 0, (("outer", 512), ("tail", 0))), (12345, 1, 1, 0, ()),
            (16384, -1, 1, 0, (("fuse_copy", 0),)), (16384, 1, 1, 0, (("depth2", 1), ("pair_rest", 1))), (8192, 0, 1, 0, (("nb", 1024),)),
            (8192, -1, 1, 0, (("lookahead", 0),)), (8192, -1, 1, 0, (("fastdiag", 0),)), (8192, -1, 1, 0, (("inner_la", 1), ("depth2", 1))),
        ]:
            scenario("cholinv n=%d ci=%d split=%d bc=%d %s" % (n, ci, split, bc, dict(opts) or ""), us)(
                lambda r, a=(n, ci, split, bc, opts): cholinv_case(r, *a))
        # ---- 1 x P plan, every simulated rank
        for (n, nb, P, opts, ci) in [(4096, 128, 4, (), -1), (4096, 128, 4, (("safe", 1),), -1), (4096, 128, 4, (("strip", 2), ("depth2", 1)), -1),
                                     (2049, 128, 3, (), -1), (8192, 512, 8, (), -1), (2048, 128, 4, (), 1), (2048, 128, 4, (), 0), (1024, 128, 1, (), 1),
                                     (65536, 512, 8, (), -1),
                                     (4096, 128, 2, (), -1), (3000, 128, 8, (), 1), (4096, 256, 5, (("strip", 1),), 0), (2048, 128, 4, (("safe", 1),), 1),
                                     (4096, 128, 7, (("depth2", 0),), -1), (1152, 128, 8, (), -1)]:
            for p in range(P):
                scenario("dist n=%d nb=%d P=%d rank=%d %s ci=%d" % (n, nb, P, p, dict(opts) or "", ci), us,
                         "dist n=%d nb=%d P=%d %s ci=%d" % (n, nb, P, dict(opts) or "", ci), p, P)(
                    lambda r, a=(n, nb, P, p, opts, ci): dist_case(r, *a))
        # ---- the same behind the cholinv handle (comm of size P) with the reference's element-cyclic layout
        for p in range(8):
            def cyc(r, p=p):
                comm = Comm(p, 8)
                cholinv_case(r, 1024, 1, 1, -2, (("nb", 128), ("cyclic_c", 2)), reps=1, comm=comm.handle, local_cols=512)
                comm.close()
            scenario("cholinv over 8 ranks, cyclic_c=2, rank=%d" % p, us, "cholinv over 8 ranks, cyclic_c=2", p, 8)(cyc)
        # ---- Pr x Pc plan
        for (n, nb, Pr, Pc, opts) in [(4096, 128, 2, 2, ()), (4096, 128, 2, 4, (("strip", 1),)), (2048, 128, 2, 2, (("complete_inv", 1),)),
                                      (1000, 128, 2, 2, ()), (4096, 128, 1, 4, (("strip", 2),)), (4096, 128, 2, 2, (("safe", 1),)),
                                      (65536, 512, 2, 4, ()),
                                      (4096, 128, 4, 4, ()), (4096, 128, 1, 8, ()), (3000, 128, 2, 4, (("complete_inv", 0),)), (4096, 128, 4, 4, (("complete_inv", 1),)),
                                      (2048, 128, 2, 2, (("complete_inv", 1), ("safe", 1))), (4096, 128, 2, 4, (("depth2", 0),)), (1152, 128, 4, 8, ())]:
            for pr in range(Pr):
                for pc in range(Pc):
                    scenario("dist2d n=%d nb=%d %dx%d at (%d,%d) %s" % (n, nb, Pr, Pc, pr, pc, dict(opts) or ""), us,
                             "dist2d n=%d nb=%d %dx%d %s" % (n, nb, Pr, Pc, dict(opts) or ""), pr * Pc + pc, Pr * Pc)(
                        lambda r, a=(n, nb, Pr, Pc, pr, pc, opts): dist2d_case(r, *a))
        # ---- strip exchange / operand moves as IPC peer copies (the stand-in "maps" a rank's own buffers as its peers')
        for (n, nb, P, opts) in [(4096, 128, 4, (("ipc", 1),)), (4096, 128, 4, (("ipc", 1), ("safe", 1))), (8192, 512, 8, (("ipc", 1),)),
                                 (3000, 128, 3, (("ipc", 1),)), (4096, 128, 2, (("ipc", 1), ("strip", 1))), (2048, 128, 8, (("ipc", 1), ("complete_inv", 1)))]:
            for p in range(P):
                scenario("dist n=%d nb=%d P=%d rank=%d %s" % (n, nb, P, p, dict(opts)), us, "dist n=%d nb=%d P=%d %s" % (n, nb, P, dict(opts)), p, P)(
                    lambda r, a=(n, nb, P, p, opts, -1): dist_case(r, *a))
        for (n, nb, Pr, Pc, opts) in [(2048, 128, 2, 2, (("ipc", 1),)), (4096, 128, 2, 4, (("ipc", 1),)), (2049, 256, 2, 4, (("ipc", 1), ("complete_inv", 1))),
                                      (4096, 128, 4, 4, (("ipc", 1),)), (4096, 128, 1, 4, (("ipc", 1),)), (3000, 128, 2, 2, (("ipc", 1), ("strip", 1), ("safe", 1)))]:
            for pr in range(Pr):
                for pc in range(Pc):
                    scenario("dist2d n=%d nb=%d %dx%d at (%d,%d) %s" % (n, nb, Pr, Pc, pr, pc, dict(opts)), us,
                             "dist2d n=%d nb=%d %dx%d %s" % (n, nb, Pr, Pc, dict(opts)), pr * Pc + pc, Pr * Pc)(
                        lambda r, a=(n, nb, Pr, Pc, pr, pc, opts): dist2d_case(r, *a))
        # ---- SUMMA (GEMM, TRMM, SYRK overloads, util::transpose) on d x d x c grids; CholeskyQR on the c x d x c grid
        for (size, c, M, N, K, chunks) in [(8, 2, 300, 300, 300, 2), (4, 1, 512, 256, 512, 0), (9, 1, 300, 300, 300, 3), (1, 1, 256, 256, 256, 0), (27, 3, 270, 270, 270, 0)]:
            for rank in range(size):
                scenario("summa size=%d c=%d rank=%d %dx%dx%d chunks=%d" % (size, c, rank, M, N, K, chunks), us,
                         "summa size=%d c=%d %dx%dx%d chunks=%d" % (size, c, M, N, K, chunks), rank, size)(
                    lambda r, a=(size, c, rank, M, N, K, chunks): summa_case(r, *a))
        for (size, c, m, n, iters) in [(8, 2, 4096, 128, 2), (4, 1, 4096, 64, 2), (16, 2, 8192, 256, 1)]:
            for rank in range(size):
                scenario("cacqr grid size=%d c=%d rank=%d m=%d n=%d iter=%d" % (size, c, rank, m, n, iters), us,
                         "cacqr grid size=%d c=%d m=%d n=%d iter=%d" % (size, c, m, n, iters), rank, size)(
                    lambda r, a=(size, c, rank, m, n, iters): cacqr_grid_case(r, *a))
        # ---- redistribution element-cyclic <-> block-cyclic, descriptors with pinned staging, the operator seam
        for (n, nb, size, c, Pr) in [(1024, 128, 8, 2, 1), (1000, 128, 8, 2, 2), (512, 64, 4, 1, 2)]:
            for rank in range(size):
                scenario("redist n=%d nb=%d size=%d c=%d Pr=%d rank=%d" % (n, nb, size, c, Pr, rank), us,
                         "redist n=%d nb=%d size=%d c=%d Pr=%d" % (n, nb, size, c, Pr), rank, size)(
                    lambda r, a=(n, nb, size, c, Pr, rank): redist_case(r, *a))
        # (6144 on one process: a 288 MiB piece = five 64 MiB chunks through the two pinned buffers - the host refills a buffer the
        #  copy engine may still be reading, and reads one it may still be writing: its accesses are in the trace as "HA" lines)
        for (n, nb, Pr, Pc) in [(1000, 128, 2, 2), (2048, 256, 1, 4), (300, 128, 2, 4), (6144, 512, 1, 1)]:
            for pr in range(Pr):
                for pc in range(Pc):
                    scenario("desc n=%d nb=%d %dx%d at (%d,%d)" % (n, nb, Pr, Pc, pr, pc), us)(lambda r, a=(n, nb, Pr, Pc, pr, pc): desc_case(r, *a))
        for (m, n, k) in [(1024, 1024, 1024), (1000, 777, 515), (4096, 8, 4096), (64, 64, 64), (2048, 2048, 128)]:
            scenario("operators m=%d n=%d k=%d" % (m, n, k), us)(lambda r, a=(m, n, k): operators_case(r, *a))
        scenario("plan life cycles: unused, reconfigured between calls, refused", us)(lifecycle_case)
        # ---- mixed precision, one GPU and P ranks
        for (n, nrhs, opts) in [(4096, 8, ()), (8192, 8, (("strip", 1),)), (16384, 8, (("pair_rest", 0),)), (8192, 8, (("split", 0),)), (65536, 8, ()),
                                (16384, 8, (("reserve", 8),)), (16384, 8, (("solve3", 0),)), (16384, 200, (("update_kernel", 1),)), (8192, 8, (("reserve", 16), ("split", 0))),
                                (24576, 8, ()), (16384, 8, (("update_kernel", 0),))]:
            scenario("mpchol n=%d %s" % (n, dict(opts) or ""), us)(lambda r, a=(n, nrhs, opts): mpchol_case(r, *a))
        for (n, nb, P) in [(2048, 256, 1), (2048, 256, 4), (1280, 256, 4), (8192, 512, 8), (1152, 256, 2), (4096, 128, 3), (8192, 1024, 8), (2048, 256, 7)]:
            for p in range(P):
                scenario("dmp n=%d nb=%d P=%d rank=%d" % (n, nb, P, p), us, "dmp n=%d nb=%d P=%d" % (n, nb, P), p, P)(lambda r, a=(n, nb, P, p): dmp_case(r, *a))
        # ---- CholeskyQR
        for (m, n, iters, P) in [(16384, 256, 2, 1), (16384, 128, 2, 4), (4096, 64, 1, 2), (1 << 21, 256, 2, 8)]:
            for p in range(P):
                scenario("cacqr m=%d n=%d iter=%d P=%d rank=%d" % (m, n, iters, P, p), us, "cacqr m=%d n=%d iter=%d P=%d" % (m, n, iters, P), p, P)(
                    lambda r, a=(m, n, iters, P, p): cacqr_case(r, *a))
    if 0 in user_streams:
        # ---- the CBLAS / LAPACKE offload library (always on the NULL stream): potrf at a size with look-ahead on helper streams, ragged products
        for (what, m, n, k) in [("dpotrf", 0, 5000, 0), ("dpotrf", 0, 1000, 0), ("dgemm", 3000, 2000, 1000), ("dgemm", 129, 77, 33), ("dgemm", 4096, 4, 4096),
                                ("dtrmm", 2048, 1000, 0), ("dsyrk", 0, 3000, 700)]:
            scenario("cblas offload %s m=%d n=%d k=%d" % (what, m, n, k), 0)(lambda r, a=(what, m, n, k): cblas_case(r, *a))
    live = os.path.join(build_shim.OUT, "live_%d.txt" % os.getpid())
    shim.shim_live_report(live.encode())
    leaks = open(live).read().splitlines()
    os.unlink(live)
    json.dump({"results": RESULTS, "live_allocations_at_exit": leaks}, open(out_path, "w"), indent=1)
    bad = [x for x in RESULTS if x["findings"]]
    print("%d scenarios, %d with findings" % (len(RESULTS), len(bad)))
    for x in bad[:40]:
        print(" *", x["name"])
        for f in x["findings"][:6]:
            print("     ", f)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(build_shim.OUT, "scenarios.json"),
         tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (0, 1))
