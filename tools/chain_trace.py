"""Per-step timeline of ONE launch of the one-launch diagonal-block chain (cap_chain_trace_arm / _read): on an idle GPU and in the
middle of a mixed-precision factorization.     python tools/chain_trace.py [G] [which]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from capital_amd import _lib, cholinv, mixed
from capital_amd.matrix import matrix
G = int(sys.argv[1]) if len(sys.argv) > 1 else 32
which = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L = _lib.lib()
def arm(k): assert L.cap_chain_trace_arm(C.c_int64(k)) == 0
def read(tag, nblk):
    buf = np.zeros((64, 32, 8), dtype=np.int64)
    assert L.cap_chain_trace_read(buf.ctypes.data_as(C.c_void_p)) == 0
    live = [w for w in range(62) if buf[w, 0, 0] != 0]
    if not live:
        print(tag, "no trace recorded"); return
    t = buf[live][:, :nblk, :].astype(np.float64) / 100.0          # us
    t0 = t[:, 0, 0].min()
    print("%s: %d workgroups, first step entered at +%.1f .. +%.1f us (residency skew), whole chain %.1f us"
          % (tag, len(live), t[:, 0, 0].min() - t0, t[:, 0, 0].max() - t0, t[:, nblk - 1, 5].max() - t0))
    print("  step | wg0: solve  wait  update  leaf  wait | workers: solve max   update max / mean   idle mean | step total")
    m = buf[live][:, nblk, :].astype(np.float64) / 100.0
    if m[:, 1].max() > 0:
        lv = [j for j in range(1, 8) if m[:, j].max() > 0]
        print("  inverse levels: start +%.1f us after the last step; level ends (max over workgroups, us after start): %s; whole launch %.1f us"
              % (m[:, 0].min() - t[:, nblk - 1, 5].max(), " ".join("%.1f" % (m[:, j].max() - m[:, 0].min()) for j in lv), m[:, lv[-1]].max() - t0))
    lf = buf[63, :nblk, :4].astype(np.float64) / 100.0
    if lf[:, 3].max() > 0:
        sel = lf[1:]                                                  # (step 0 has no update in front of it)
        print("  workgroup 0's leaf, mean over steps (us): zero-fill %.1f | potrf %.1f | R out %.1f | trtri %.1f | Dinv out %.1f"
              % ((sel[:, 0] - t[0, 1:, 3]).mean(), (sel[:, 1] - sel[:, 0]).mean(), (sel[:, 2] - sel[:, 1]).mean(), (sel[:, 3] - sel[:, 2]).mean(),
                 (t[0, 1:, 4] - sel[:, 3]).mean()))
    pp = buf[62].reshape(-1)[:16 * 16].reshape(16, 16).astype(np.float64) / 100.0      # potrf_lds phases of steps 1..15: 4 panels x 3 stamps
    if pp[1:, 0].max() > 0:
        sel = pp[1:nblk]; st = lf[1:, 0]
        names = ["potrf16", "row", "update"]; prev = st; out = []
        for pnl in range(4):
            for j in range(3):
                if pnl == 3 and j > 0: break
                cur = sel[:, pnl * 3 + j]; out.append("%s%d %.2f" % (names[j], pnl, (cur - prev).mean())); prev = cur
        print("  potrf_lds phases, mean over steps (us): " + " | ".join(out))
    for s in range(nblk):
        w0 = t[0, s]; wk = t[1:, s]
        if s == 0:
            print("  %4d | %10s %5s %7s %5.1f %5.1f | %51s | %9.1f" % (s, "", "", "", w0[4] - w0[0], w0[5] - w0[4], "", t[:, s, 5].max() - t[:, s, 0].min()))
            continue
        print("  %4d | %10.1f %5.1f %7.1f %5.1f %5.1f | %18.1f %12.1f / %5.1f %11.1f | %9.1f"
              % (s, w0[1] - w0[0], w0[2] - w0[1], w0[3] - w0[2], w0[4] - w0[3], w0[5] - w0[4], (wk[:, 1] - wk[:, 0]).max(),
                 (wk[:, 3] - wk[:, 2]).max(), (wk[:, 3] - wk[:, 2]).mean(), ((wk[:, 2] - wk[:, 1]) + (wk[:, 5] - wk[:, 3])).mean(),
                 t[:, s, 5].max() - t[:, s, 0].min()))
# idle GPU: fp64 plan, two 1024 panels
n = 2048
A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
pack = cholinv.info(-1, 1, -2, 'U'); pack.set_option("nb", 1024); pack.set_option("chain_coop", G)
cholinv.factor(A, pack, None); torch.cuda.synchronize()
import time
for g in (0, G):
    pack.set_option("chain_coop", g)
    cholinv.factor(A, pack, None); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): cholinv.factor(A, pack, None)
    torch.cuda.synchronize(); print("idle GPU, n=2048 nb=1024 (2 chains), chain_coop=%d: %.1f us per factor" % (g, (time.perf_counter() - t0) / 20 * 1e6))
arm(0); cholinv.factor(A, pack, None); torch.cuda.synchronize(); read("idle", 16)
pack._release()
# inside the mixed factorization
n = int(os.environ.get("CHAIN_NMP", "65536"))
A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
p = mixed.plan(n, 8); p.set_option("chain_coop", G)
p.factor(A); torch.cuda.synchronize()
for k in (which, which + 20):
    arm(k); p.factor(A); torch.cuda.synchronize(); read("mixed N=%d, chain %d" % (n, k), 16)
p.set_option("chain_coop", 0); p.close()
