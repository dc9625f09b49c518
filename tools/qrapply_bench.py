"""Times the n = 256 CholeskyQR kernels alone (internal launchers, C++-mangled; a tool, not API).  CAP_CQR_DIAG=1/2/3 strips
stores / MFMAs / both from qrapply256 for timing surgery."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import _lib
L = _lib.lib()
m, n = 1 << 21, 256
Q = torch.randn(n, m, dtype=torch.float64, device="cuda")        # column-major m x n
Qo = torch.empty_like(Q)
Ri = torch.triu(torch.randn(n, n, dtype=torch.float64, device="cuda")).t().contiguous()   # column-major upper
f = getattr(L, "_Z21cap_qrapply256_launchPKdlS0_PdllP12ihipStream_t")
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
st = torch.cuda.current_stream().cuda_stream
def run(out):
    rc = f(Q.data_ptr(), m, Ri.data_ptr(), out.data_ptr(), m, m, st)
    assert rc == 0, rc
for name, out in (("out-of-place", Qo), ("in-place", Q)):
    run(out); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): run(out)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 5
    print("qrapply256 %s diag=%s: %.3f ms  (%.0f GB/s r+w, %.1f TF useful)" % (name, os.environ.get("CAP_CQR_DIAG", "0"), t, 16.0 * m * n / t / 1e6, m * n * (n + 16.0) / t / 1e9))
