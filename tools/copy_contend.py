"""What does a concurrent 512 MiB strip exchange cost the bulk SYRK?  One GPU, so the peer is emulated:

  alone            the trailing-update SYRK (m = 32768, K = 1024, upper tiles) back to back
  + blit copies    a second stream looping 512 MiB device-to-device copies with the default kind (a CU blit kernel: the
                   stand-in for a collective's kernel competing for CU slots)
  + NoCU copies    the same copies with hipMemcpyDeviceToDeviceNoCU (SDMA engines: what the IPC peer pushes of csrc/dist.hip use)
  + D2H copies     512 MiB device -> pinned host (SDMA over PCIe; HBM read traffic only)

    python tools/copy_contend.py
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import _lib

L = _lib.lib()
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
D2D, D2D_NOCU, D2H = 3, 1024, 2

m, k = 32768, 1024
Apan = torch.randn(m, k, dtype=torch.float64, device="cuda")          # k x m column-major (K-contiguous)
Cm = torch.zeros(m, m, dtype=torch.float64, device="cuda")
src = torch.empty(512 << 17, dtype=torch.float64, device="cuda")      # 512 MiB
dst = torch.empty_like(src)
host = torch.empty(512 << 17, dtype=torch.float64).pin_memory()
s_copy = torch.cuda.Stream()
flops = float(m) * (m + 1) * k


def syrk_loop(reps):
    for _ in range(reps):
        L.cap_dsyrk(1, 1, m, k, -1.0, Apan.data_ptr(), k, 1.0, Cm.data_ptr(), m, None)


def run(label, kind=None, to_host=False, reps=8):
    syrk_loop(1); torch.cuda.synchronize()
    stop = [False]
    ncopies = 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    syrk_loop(reps)
    e1.record()
    if kind is not None:
        # keep the copy stream busy for the whole SYRK loop: enqueue more copies than can finish
        t_est = reps * flops / 70e12
        ncopies = int(t_est * 1.0e12 / (512 << 20)) + 4 if not to_host else int(t_est * 50e9 / (512 << 20)) + 2
        for _ in range(ncopies):
            hip.hipMemcpyAsync(host.data_ptr() if to_host else dst.data_ptr(), src.data_ptr(), 512 << 20, kind, s_copy.cuda_stream)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    torch.cuda.synchronize()
    print("%-34s SYRK %.2f ms = %.1f TF   (%d x 512 MiB copies enqueued beside it)" % (label, ms, flops / ms / 1e9, ncopies), flush=True)


run("alone")
run("+ device-to-device (blit kernel)", D2D)
run("+ device-to-device NoCU (SDMA)", D2D_NOCU)
run("+ device-to-pinned-host (SDMA)", D2H, to_host=True)
run("alone (again)")
