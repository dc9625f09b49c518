"""The shader clock the chip holds under each workload (DVFS), sampled by a one-wave monitor kernel (tools/micro/clockmon.hip) while the
workload runs: s_memtime (shader clock) against s_memrealtime (100 MHz).  Prints, per workload, the mean / min clock inside its window and the
fp64 / bf16 matrix-pipe peak AT that clock (256 CUs x 4 SIMDs x 32 fp64 resp. 1024 bf16 flop per cycle) next to the nominal 78.6 TF / 2.5 PF.
    python tools/clock_probe.py        (GPU box; builds tools/micro/libclockmon.so with hipcc if it is missing)"""
import ctypes as C, os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
so = os.path.join(R, "tools", "micro", "libclockmon.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(R, "tools", "micro", "clockmon.hip")])
import numpy as np, torch
from capital_amd import cholinv, cacqr, mixed, _lib
from capital_amd.matrix import matrix
M = C.CDLL(so)
M.cm_mark.argtypes = [C.c_void_p, C.c_int]
NS = 600000
assert M.cm_start(NS, 27) == 0
st = lambda: torch.cuda.current_stream().cuda_stream
work = []
def window(name, f, flops=None, kind="fp64"):
    f(); torch.cuda.current_stream().synchronize()
    i = len(work) * 2
    M.cm_mark(st(), i); f(); f(); M.cm_mark(st(), i + 1); torch.cuda.current_stream().synchronize()
    work.append((name, i, flops, kind, 2))
# idle
time.sleep(0.3)
work.append(("idle (0.3 s before any work)", None, None, None, 0))
L = _lib.lib()
h = 8192
a = torch.randn(h, h, dtype=torch.float64, device="cuda"); c = torch.empty(h, h, dtype=torch.float64, device="cuda")
L.cap_dgemm.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_int64, C.c_void_p]
def gemm():
    for _ in range(20): L.cap_dgemm(1, 0, h, h, h, 1.0, a.data_ptr(), h, a.data_ptr(), h, 0.0, c.data_ptr(), h, None)
window("fp64 TN GEMM 8192^3 x 20", gemm, 20 * 2.0 * h ** 3)
n = 65536
A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
pack = cholinv.info(-1, 1, -5, 'U')
window("fp64 Cholesky N=65536 (headline)", lambda: cholinv.factor(A, pack, None), n ** 3 / 3.0)
p = mixed.plan(n, 8)
window("mixed-precision factor N=65536 (bf16 updates)", lambda: p.factor(A), None, "bf16")
mb = 32768
a16 = torch.randn(mb, 4096, device="cuda").to(torch.bfloat16); c32 = torch.zeros(mb, mb, device="cuda")
def upd():
    for _ in range(10): L.cap_bf16_update(-1, mb, mb, 4096, C.c_float(-1.0), a16.data_ptr(), 4096, a16.data_ptr(), 4096, c32.data_ptr(), mb, 1, 0, None)
window("bf16 update m=32768 K=4096 x 10 (alone)", upd, 10 * 2.0 * 4096 * (mb * (mb + 1) / 2), "bf16")
m, k = 1 << 21, 256
Q = matrix(k, m, 1, 1); Q.distribute_random(0, 0, 1, 1, 0)
qp = cacqr.info(2, cholinv.info(1, 1, 0, 'U'))
window("CholeskyQR2 2^21 x 256 (gram256 + qrapply256: 4.46 M matrix-pipe cycles per CU and kernel)", lambda: cacqr.factor(Q, qp, None), None)
buf = np.zeros(2 * NS, dtype=np.uint64); marks = np.zeros(256, dtype=np.uint64)
ns = M.cm_stop(buf.ctypes.data_as(C.c_void_p), marks.ctypes.data_as(C.c_void_p))
assert ns > 10, ns
S = buf[:2 * ns].reshape(ns, 2).astype(np.float64)
rt, mt = S[:, 0], S[:, 1]
ghz = np.diff(mt) / np.diff(rt) * 0.1            # cycles per 10 ns tick -> GHz
tm = 0.5 * (rt[1:] + rt[:-1])
print("%d samples over %.2f s" % (ns, (rt[-1] - rt[0]) / 1e8))
if os.environ.get("CLOCK_PROBE_DEBUG"): print("rt0 %.0f rt1 %.0f marks %s" % (rt[0], rt[-1], [int(x) for x in marks[:12]]))
for name, i, flops, kind, reps in work:
    if i is None:
        sel = tm < rt[0] + 0.25e8
    else:
        sel = (tm >= float(marks[i])) & (tm <= float(marks[i + 1]))
    if sel.sum() < 2: print("%-48s (no samples)" % name); continue
    g = ghz[sel]
    line = "%-48s clock mean %.3f GHz (min %.3f, max %.3f, %d samples)" % (name, g.mean(), g.min(), g.max(), sel.sum())
    if kind == "fp64":
        pk = 256 * 4 * 32 * g.mean() / 1e3
        line += " | fp64 MFMA peak at this clock %.1f TF (nominal 78.6)" % pk
        if flops:
            t = (float(marks[i + 1]) - float(marks[i])) / 1e8 / reps
            line += " | achieved %.1f TF = %.3f of it" % (flops / t / 1e12, flops / t / 1e12 / pk)
    elif kind == "bf16":
        pk = 256 * 4 * 1024 * g.mean() / 1e3
        line += " | bf16 MFMA peak at this clock %.0f TF (nominal 2500)" % pk
        if flops:
            t = (float(marks[i + 1]) - float(marks[i])) / 1e8 / reps
            line += " | achieved %.0f TF = %.3f of it" % (flops / t / 1e12, flops / t / 1e12 / pk)
    print(line)
