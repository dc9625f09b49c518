"""Quick GPU check of cap_dgemm/cap_dsyrk: correctness vs torch fp64 matmul + timing."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
h = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "capital_amd/lib/libcapital_amd.so"))
i64, dbl, ptr, cint = C.c_int64, C.c_double, C.c_void_p, C.c_int
h.cap_dgemm.argtypes = [cint, cint, i64, i64, i64, dbl, ptr, i64, ptr, i64, dbl, ptr, i64, ptr]
h.cap_dsyrk.argtypes = [cint, cint, i64, i64, dbl, ptr, i64, dbl, ptr, i64, ptr]
dev = "cuda"
def colmajor(r, c, ld=None):
    ld = ld or r
    buf = torch.randn(c, ld, dtype=torch.float64, device=dev)
    return buf, buf[:, :r].t()   # view is r x c "column-major with leading dim ld"
def gemm(ta, tb, m, n, k, alpha, beta, pad=0):
    Ab, A = colmajor(*( (k, m) if ta else (m, k)), ld=((k if ta else m) + pad))
    Bb, B = colmajor(*( (n, k) if tb else (k, n)), ld=((n if tb else k) + pad))
    Cb, Cm = colmajor(m, n, ld=m + pad)
    ref = alpha * ((A.t() if ta else A) @ (B.t() if tb else B)) + beta * Cm
    st = h.cap_dgemm(ta, tb, m, n, k, alpha, Ab.data_ptr(), Ab.shape[1], Bb.data_ptr(), Bb.shape[1], beta, Cb.data_ptr(), Cb.shape[1], None)
    torch.cuda.synchronize()
    err = (Cm - ref).abs().max().item() / max(ref.abs().max().item(), 1e-300)
    return st, err
ok = True
for (ta, tb) in [(1, 0), (0, 0), (1, 1), (0, 1)]:
    for (m, n, k, pad) in [(128, 128, 16, 0), (256, 384, 64, 0), (100, 37, 23, 3), (129, 257, 130, 1), (512, 256, 1024, 0), (64, 64, 8, 0), (1, 1, 1, 0)]:
        st, err = gemm(ta, tb, m, n, k, -1.0, 1.0, pad)
        flag = "OK " if (st == 0 and err < 1e-13) else "BAD"
        ok &= flag == "OK "
        print(f"{flag} ta={ta} tb={tb} m={m} n={n} k={k} pad={pad} status={st} relerr={err:.2e}")
# syrk upper trans
for (n, k) in [(256, 128), (1000, 77), (1024, 512)]:
    Ab, A = colmajor(k, n)
    Cb, Cm = colmajor(n, n)
    C0 = Cm.clone()
    st = h.cap_dsyrk(1, 1, n, k, -1.0, Ab.data_ptr(), Ab.shape[1], 1.0, Cb.data_ptr(), Cb.shape[1], None)
    torch.cuda.synchronize()
    ref = C0 - A.t() @ A
    up = torch.triu(torch.ones(n, n, device=dev, dtype=torch.bool))
    e1 = ((Cm - ref)[up]).abs().max().item() / ref.abs().max().item()
    e2 = ((Cm - C0)[~up]).abs().max().item()
    flag = "OK " if (st == 0 and e1 < 1e-13 and e2 == 0) else "BAD"
    ok &= flag == "OK "
    print(f"{flag} syrk n={n} k={k} upper_err={e1:.2e} lower_touched={e2:.1e}")
print("ALL OK" if ok else "FAILURES")
# timing
def bench(m, n, k, syrk=False, reps=5):
    Ab, A = colmajor(k, m); Bb, B = colmajor(k, n); Cb, Cm = colmajor(m, n)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    def run():
        if syrk: h.cap_dsyrk(1, 1, n, k, -1.0, Ab.data_ptr(), k, 1.0, Cb.data_ptr(), m, None)
        else: h.cap_dgemm(1, 0, m, n, k, -1.0, Ab.data_ptr(), k, Bb.data_ptr(), k, 1.0, Cb.data_ptr(), m, None)
    run(); torch.cuda.synchronize()
    s.record()
    for _ in range(reps): run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    fl = (m * (n + 1) * k) if syrk else 2.0 * m * n * k
    print(f"{'syrk' if syrk else 'gemm'} TN m={m} n={n} k={k}: {ms:.3f} ms  {fl / ms * 1e-9:.2f} TFLOP/s")
for (m, n, k) in [(4096, 4096, 4096), (8192, 8192, 512), (8192, 8192, 1024), (8192, 8192, 8192), (16384, 16384, 512), (16384, 16384, 2048), (32768, 32768, 512)]:
    bench(m, n, k)
for (n, k) in [(8192, 512), (16384, 1024), (32768, 512), (32768, 1024)]:
    bench(n, n, k, syrk=True)
# torch (rocBLAS) comparator
a = torch.randn(8192, 8192, dtype=torch.float64, device=dev); b = torch.randn(8192, 8192, dtype=torch.float64, device=dev)
torch.matmul(a.t(), b); torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record(); 
for _ in range(3): torch.matmul(a.t(), b)
e.record(); torch.cuda.synchronize()
print(f"torch/rocBLAS dgemm TN 8192^3: {2*8192**3/(s.elapsed_time(e)/3)*1e-9:.2f} TFLOP/s (comparator only)")
