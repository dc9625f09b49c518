"""Round 5: where the one-launch diagonal-block chain loses its time INSIDE the fp64 factorization (N = 32768: the verdict's 1.41 ms per
launch against 0.58 ms alone): per-step stamps of single launches early, in the middle and in the chain-bound tail, plus the factor time with
the counter fences on / off (CAP_CHAIN_FENCE is read once per process: run this file twice).   python tools/r05_chain_fp64.py [n]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from capital_amd import _lib, cholinv
from capital_amd.matrix import matrix
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
pack = cholinv.info(-1, 1, -5, 'U')
cholinv.factor(A, pack, None); torch.cuda.synchronize()
def t(reps=3):
    cholinv.factor(A, pack, None); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): cholinv.factor(A, pack, None)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for rnd in range(2):
    for G in (32, 0):
        pack.set_option("chain_coop", G)
        tf = t()
        print("fence=%s N=%d chain_coop=%d: %.2f ms = %.2f TF | info %d fallbacks %d" % (os.environ.get("CAP_CHAIN_FENCE", "default(1)"), n, G, tf * 1e3,
              n ** 3 / 3 / tf / 1e12, pack.last_info(), pack.get_option("chain_fallbacks")), flush=True)
pack.set_option("chain_coop", 32)
nblk = int(pack.get_option("nb")) // 64
nch = n // int(pack.get_option("nb"))
for k in (2, nch // 4, nch // 2, (3 * nch) // 4, nch - 8, nch - 2):
    assert L.cap_chain_trace_arm(C.c_int64(k)) == 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cholinv.factor(A, pack, None); torch.cuda.synchronize()
    buf = np.zeros((64, 32, 8), dtype=np.int64)
    assert L.cap_chain_trace_read(buf.ctypes.data_as(C.c_void_p)) == 0
    live = [w for w in range(62) if buf[w, 0, 0] != 0]
    if not live:
        print("chain %d: no trace" % k); continue
    tt = buf[live][:, :nblk, :].astype(np.float64) / 100.0
    t0s = tt[:, 0, 0]
    steps = [tt[:, s, 5].max() - tt[:, s, 0].min() for s in range(nblk)]
    w0 = tt[0]
    print("chain %2d of %d (columns left %5d): %d workgroups entered over %.1f us; sweep %.1f us; steps (us) %s | wg0 leaf mean %.1f us, wg0 wait-at-end mean %.1f us, workers update max mean %.1f us"
          % (k, nch, n - k * nblk * 64, len(live), t0s.max() - t0s.min(), tt[:, nblk - 1, 5].max() - t0s.min(), " ".join("%.0f" % x for x in steps),
             float(np.mean([w0[s, 4] - w0[s, 3] for s in range(1, nblk)])), float(np.mean([w0[s, 5] - w0[s, 4] for s in range(1, nblk)])),
             float(np.mean([(tt[1:, s, 3] - tt[1:, s, 2]).max() for s in range(1, nblk)]))), flush=True)
