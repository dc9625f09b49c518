"""Worker of tests/test_gpu_late_read.py: fresh one-rank plans of three kinds (cap_dmp = distributed mixed precision, cap_dist = 1 x P fp64,
cap_mpchol = single-GPU mixed precision), each created, factored ONCE and read AT ONCE (the plan's info query, then the factor), next to idle
peer processes that hold contexts on the same GPU - the situation of round 5's one late read.  Any read that differs
from the kind's first, carefully synchronised read is a failure: exit code 1.

    python tests/late_read_worker.py <iterations per kind> <idle peers>"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

if len(sys.argv) > 1 and sys.argv[1] == "--peer":
    torch.cuda.set_device(0)
    x = torch.ones(1 << 20, device="cuda"); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        y = x * 2
    torch.cuda.synchronize()
    print("peer up", flush=True)
    sys.stdin.read()
    sys.exit(0)

from capital_amd import _lib, mixed, dist_cholesky
from capital_amd.matrix import matrix

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 700
npeers = int(sys.argv[2]) if len(sys.argv) > 2 else 3
peers = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--peer"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for _ in range(npeers)]
for q in peers:
    q.stdout.readline()
n, nb = 1024, 256
torch.cuda.set_device(0)
g = np.random.default_rng(17).standard_normal((n, n))
a = g @ g.T / n + 0.5 * np.eye(n); a = 0.5 * (a + a.T)
A1 = torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
Am = matrix(n, n, 1, 1); Am.from_numpy(a)
L = _lib.lib()


class SelfComm:
    def __init__(self):
        self.handle = C.c_void_p(); self.rank, self.size = 0, 1
        _lib.check(L.cap_comm_create_self(C.byref(self.handle)), "cap_comm_create_self")


def one_dmp(careful):
    sc = SelfComm()
    p = mixed.dist_plan(n, sc, nb=nb, nrhs_max=5)
    p.factor(A1)
    info = p.last_info()
    if careful: time.sleep(0.2); torch.cuda.synchronize()
    R = p.R32_local()
    p.close(); L.cap_comm_destroy(sc.handle)
    return info, np.asarray(R)


def one_dist(careful):
    sc = SelfComm()
    ctx = dist_cholesky.Context(n, nb, sc)
    ctx.set_local(a)
    ctx.factor()
    info = ctx.last_info()
    if careful: time.sleep(0.2); torch.cuda.synchronize()
    R = ctx.local_R()
    L.cap_dist_plan_destroy(ctx.plan); ctx.plan = None
    L.cap_comm_destroy(sc.handle)
    return info, np.asarray(R)


def one_mpchol(careful):
    p = mixed.plan(n, 4)
    p.factor(Am)
    info = p.last_info()
    if careful: time.sleep(0.2); torch.cuda.synchronize()
    R = p.R32()
    R = R.cpu().numpy() if hasattr(R, "cpu") else np.asarray(R)
    p.close()
    return info, R


bad = 0
for name, fn in (("cap_dmp", one_dmp), ("cap_dist", one_dist), ("cap_mpchol", one_mpchol)):
    _, ref = fn(True)
    _, ref2 = fn(True)
    assert np.array_equal(ref, ref2), name + ": two careful reads differ"
    t0 = time.time(); late = 0
    for it in range(iters):
        info, R = fn(False)
        if info != 0 or not np.array_equal(R, ref):
            late += 1
            print("%s iteration %d: info %d, distance %.4f from the careful read" % (name, it, info, np.linalg.norm(R.astype(np.float64) - ref) / np.linalg.norm(ref)), flush=True)
    print("%s: %d fresh plans factored once and read at once beside %d idle peers: %d late or wrong reads (%.1f s)" % (name, iters, npeers, late, time.time() - t0), flush=True)
    bad += late
for q in peers:
    q.stdin.close(); q.wait()
sys.exit(1 if bad else 0)
