"""The bulk update's shape launched alone through the C ABI: C (n x n upper) -= A^T A, A: K x n.   python tools/syrk_alone.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import _lib
L = _lib.lib()
shapes = ((57344, 2048), (57344, 1024), (32768, 2048), (16384, 2048), (8192, 2048), (8192, 8192))
if os.environ.get("SYRK_ONLY"): shapes = shapes[:1]
for n, K in shapes:
    a = torch.randn(n, K, dtype=torch.float64, device="cuda")       # column-major K x n (ld = K)
    c = torch.zeros(n, n, dtype=torch.float64, device="cuda")
    def run(): 
        rc = L.cap_dsyrk(1, 1, n, K, -1.0, a.data_ptr(), K, 1.0, c.data_ptr(), n, None)
        assert rc == 0, rc
    run(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    fl = float(K) * n * (n + 1)
    print("upper SYRK n=%d K=%d alone: %.3f ms = %.1f TF (%.3f of 78.6)" % (n, K, ms, fl / ms / 1e9, fl / ms / 1e9 / 78.6), flush=True)
    del a, c
