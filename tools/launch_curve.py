"""The trailing-update launches of one fp64 factorization, one by one, IN the run (plan option "profile": HIP events around every launch on its
stream): duration, algorithmic flops, TFLOP/s - next to the same SYRK shape launched alone.   python tools/launch_curve.py [N]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import _lib, cholinv
from capital_amd.matrix import matrix
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
L = _lib.lib()
A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
pack = cholinv.info(-1, 1, -5, 'U')
pack.set_option("profile", 1)
cholinv.factor(A, pack, None); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); cholinv.factor(A, pack, None); e1.record(); torch.cuda.synchronize()
cap = 1024
ms = (C.c_double * cap)(); fl = (C.c_double * cap)(); cnt = C.c_int64(0)
_lib.check(L.cap_cholinv_profile_launches(pack._plan, ms, fl, cap, C.byref(cnt)), "cap_cholinv_profile_launches")
k = min(cnt.value, cap)
tot_ms = sum(ms[i] for i in range(k)); tot_fl = sum(fl[i] for i in range(k))
print("N=%d: factor %.1f ms (%.1f TF on N^3/3); %d update launches, %.1f ms, %.1f TF in the run (%.3f of 78.6)" % (
    n, e0.elapsed_time(e1), n ** 3 / 3 / e0.elapsed_time(e1) / 1e9, cnt.value, tot_ms, tot_fl / tot_ms / 1e9, tot_fl / tot_ms / 1e9 / 78.6))
# the same flops as a plain upper SYRK (K from the launch's flops: m(m+1)K = flops) launched alone, for a few sizes
L.cap_dsyrk.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_double, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_int64, C.c_void_p]
print("launch : ms in the run, GFLOP, TF in the run")
bins = {}
for i in range(k):
    print("%4d : %8.3f ms %9.1f GFLOP %6.1f TF" % (i, ms[i], fl[i] / 1e9, fl[i] / ms[i] / 1e9))
