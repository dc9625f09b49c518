"""qrapply256 against a plain fp64 product, several row counts, out of place and in place; then timing at 2^21 rows."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import _lib
L = _lib.lib()
n = 256
f = getattr(L, "_Z21cap_qrapply256_launchPKdlS0_PdllP12ihipStream_t")
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(1)
Ri = torch.triu(torch.randn(n, n, dtype=torch.float64, device="cuda")).t().contiguous()   # column-major upper
bad = 0
for m in (128, 256, 128 * 7, 128 * 256, 128 * 257, 128 * 300, 128 * 1029, 1 << 19):
    ld = m + 2 * (m % 256 == 0)
    Qs = torch.randn(n, ld, dtype=torch.float64, device="cuda")
    ref = (Qs[:, :m].t() @ Ri.t()).t().contiguous()          # (m x n) = Q R, stored column-major = ref[c, r]
    for name in ("out", "inplace"):
        Q = Qs.clone()
        out = torch.full_like(Q, 7.0) if name == "out" else Q
        assert f(Q.data_ptr(), ld, Ri.data_ptr(), out.data_ptr(), ld, m, st) == 0
        torch.cuda.synchronize()
        err = (out[:, :m] - ref).abs().max().item() / ref.abs().max().item()
        pad_ok = True if name == "inplace" else bool((out[:, m:] == 7.0).all().item())
        ok = err < 1e-13 and pad_ok
        bad += not ok
        print("m=%8d %-8s rel err %.2e pad untouched %s %s" % (m, name, err, pad_ok, "ok" if ok else "FAIL"))
m = 1 << 21
Q = torch.randn(n, m, dtype=torch.float64, device="cuda"); Qo = torch.empty_like(Q)
for name, out in (("out-of-place", Qo), ("in-place", Q)):
    Q.normal_()
    f(Q.data_ptr(), m, Ri.data_ptr(), out.data_ptr(), m, m, st); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        Q.normal_()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); f(Q.data_ptr(), m, Ri.data_ptr(), out.data_ptr(), m, m, st); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    t = sorted(ts)[2]
    print("qrapply256 %s diag=%s: %.3f ms (min %.3f)  (%.0f GB/s r+w, %.1f TF useful)" % (name, os.environ.get("CAP_CQR_DIAG", "0"), t, min(ts), 16.0 * m * n / t / 1e6, m * n * (n + 16.0) / t / 1e9))
print("FAILURES", bad)
