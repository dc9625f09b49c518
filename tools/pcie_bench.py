"""PCIe-inclusive rate of the Cholesky hot path: host matrix -> pinned staging -> HBM -> factor -> R back to the host.
Never the headline `value` (inputs resident in HBM); reported next to it in DESIGN.md."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from capital_amd import _lib, cholinv
from capital_amd.matrix import matrix

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
L = _lib.lib()
s = torch.cuda.current_stream().cuda_stream
A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
pack = cholinv.info(-1, 1, -5, 'U')
cholinv.factor(A, pack, None); torch.cuda.synchronize()
t0 = time.perf_counter(); cholinv.factor(A, pack, None); torch.cuda.synchronize(); t_factor = time.perf_counter() - t0
for kind in ("pageable", "pinned"):
    if kind == "pageable":
        host = np.empty((n, n)); host[:] = 1.0
        hp = host.ctypes.data
    else:
        hostt = torch.empty(n, n, dtype=torch.float64).pin_memory(); hostt.fill_(1.0)
        hp = hostt.data_ptr()
    _lib.check(L.cap_desc_export_host(A._desc(), hp, n, s))          # generated matrix -> host (warm-up of the staging buffers)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.check(L.cap_desc_import_host(A._desc(), hp, n, s)); torch.cuda.synchronize()
    t_in = time.perf_counter() - t0
    t0 = time.perf_counter()
    cholinv.factor(A, pack, None)
    R = cholinv.construct_R(pack)
    _lib.check(L.cap_desc_export_host(R._desc(), hp, n, s)); torch.cuda.synchronize()
    t_fx = time.perf_counter() - t0
    gb = n * n * 8 / 1e9
    print("N=%d %s host memory: import %.3f s (%.1f GB/s), factor %.3f s (%.1f TF), factor+export %.3f s; end to end %.3f s = %.1f TF PCIe-inclusive"
          % (n, kind, t_in, gb / t_in, t_factor, n ** 3 / 3 / t_factor / 1e12, t_fx, t_in + t_fx, n ** 3 / 3 / (t_in + t_fx) / 1e12), flush=True)
    del R
