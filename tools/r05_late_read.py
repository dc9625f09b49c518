"""Tries to reproduce the late read of DESIGN.md section 7 in ONE process: many one-rank cap_dmp plans, each created, factored ONCE and
read at once (cap_dmp_info + device synchronisation + copy, exactly what tests/dist_worker.py does), with some stream churn in between.
A read that differs from the first plan's factor is a late (or wrong) read.  CAP_DMP_PRIME=0 switches the plan's stream priming off.
Measured: priming off 400 iterations / on 200 (no peers, no second plan): 0 late reads (profiles/r05_late_read_prime{0,1}.log); priming off,
three idle peer processes, a second live plan, 500 iterations: 0 (profiles/r05_late_read_prime0_peers3.log).  The event was not reproduced.

    python tools/r05_late_read.py [iterations] [churn streams per iteration] [idle peer processes holding a context on the GPU]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from capital_amd import _lib, mixed

if len(sys.argv) > 1 and sys.argv[1] == "--peer":
    # an idle peer: a context on the GPU, one kernel, one helper stream - then nothing until the parent goes away
    import torch
    torch.cuda.set_device(0)
    x = torch.ones(1 << 20, device="cuda"); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        y = x * 2
    torch.cuda.synchronize()
    print("peer up", flush=True)
    sys.stdin.read()
    sys.exit(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
churn = int(sys.argv[2]) if len(sys.argv) > 2 else 3
npeers = int(sys.argv[3]) if len(sys.argv) > 3 else 0
import subprocess
peers = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--peer"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for _ in range(npeers)]
for q in peers:
    q.stdout.readline()
n, nb = 2048, 256
torch.cuda.set_device(0)
g = np.random.default_rng(17).standard_normal((n, n))
a = g @ g.T / n + 0.5 * np.eye(n); a = 0.5 * (a + a.T)
A1 = torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
L = _lib.lib()
hip = C.CDLL("libamdhip64.so")


class SelfComm:
    def __init__(self):
        self.handle = C.c_void_p(); self.rank, self.size = 0, 1
        _lib.check(L.cap_comm_create_self(C.byref(self.handle)), "cap_comm_create_self")


live = int(sys.argv[4]) if len(sys.argv) > 4 else 1


def one(sync_more=False):
    other = None
    if live:                                     # like the test: another plan of the process is alive and has just been used
        so = SelfComm()
        other = mixed.dist_plan(n, so, nb=nb, nrhs_max=5)
        other.factor(A1); other.factor(A1); other.last_info(); other.R32_local()
    sc = SelfComm()
    p = mixed.dist_plan(n, sc, nb=nb, nrhs_max=5)
    p.factor(A1)
    info = p.last_info()
    if sync_more:
        time.sleep(0.3); torch.cuda.synchronize()
    R = p.R32_local()
    p.close(); L.cap_comm_destroy(sc.handle)
    if other is not None:
        other.close(); L.cap_comm_destroy(so.handle)
    return info, R


_, ref = one(sync_more=True)
_, ref2 = one(sync_more=True)
print("two careful reads: bitwise equal %s, distance %.2e" % (np.array_equal(ref, ref2), np.linalg.norm(ref.astype(np.float64) - ref2) / np.linalg.norm(ref)), flush=True)
bad = 0
t0 = time.time()
for it in range(iters):
    ss = []
    for _ in range(churn):                       # stream churn: created, used once, destroyed
        s = C.c_void_p()
        hip.hipStreamCreateWithPriority(C.byref(s), C.c_uint(1), C.c_int(-1 if it & 1 else 0))
        ss.append(s)
    x = torch.zeros(1024, device="cuda")
    for s in ss:
        with torch.cuda.stream(torch.cuda.ExternalStream(s.value)):
            x += 1
    torch.cuda.synchronize()
    for s in ss:
        hip.hipStreamDestroy(s)
    info, R = one()
    d = np.linalg.norm(R.astype(np.float64) - ref) / np.linalg.norm(ref)
    if info != 0 or not d < 1e-5:
        bad += 1
        print("iteration %d: info %d, read differs from the reference factor by %.4e" % (it, info, d), flush=True)
print("prime=%s peers=%d iterations=%d late_or_wrong_reads=%d (%.1f s)" % (os.environ.get("CAP_DMP_PRIME", "1"), npeers, iters, bad, time.time() - t0), flush=True)
for q in peers:
    q.stdin.close(); q.wait()
