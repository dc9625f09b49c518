"""Reference-semantics modes (complete_inv = 0 / 1: R and R^-1) on one GPU: the blocked factorization + inverse tree against
the plain recursion, with the tree's overlap knobs; plus the NN product the tree is made of, stand-alone.

    python tools/modes_bench.py [N ...]      (default 32768)
True flops: N^3/3 (factor) + N^3/12 (complete_inv = 0, split = 1: two half-size triangular inverses) or + N^3/3 (complete_inv = 1)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import _lib, cholinv, validate
from capital_amd.matrix import matrix


def timeit(f, reps=3):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


def nn_product(h):
    """The two tree products at size h: W = -Ri11 R12 (A upper triangular) and Ri12 = W Ri22 (B upper triangular), NN forms."""
    L = _lib.lib()
    L.cap_dgemm.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                            C.c_double, C.c_void_p, C.c_int64, C.c_void_p]
    a = torch.randn(h, h, dtype=torch.float64, device="cuda"); b = torch.randn(h, h, dtype=torch.float64, device="cuda")
    c = torch.empty(h, h, dtype=torch.float64, device="cuda")
    t_nn = timeit(lambda: L.cap_dgemm(0, 0, h, h, h, 1.0, a.data_ptr(), h, b.data_ptr(), h, 0.0, c.data_ptr(), h, None), 5)
    t_tn = timeit(lambda: L.cap_dgemm(1, 0, h, h, h, 1.0, a.data_ptr(), h, b.data_ptr(), h, 0.0, c.data_ptr(), h, None), 5)
    print("dense product %d^3: NN (A M-contiguous) %.2f ms = %.1f TF | TN %.2f ms = %.1f TF" % (h, t_nn * 1e3, 2 * h ** 3 / t_nn / 1e12, t_tn * 1e3, 2 * h ** 3 / t_tn / 1e12), flush=True)


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [32768]
    for h in (2048, 8192):
        nn_product(h)
    for n in sizes:
        A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
        pack = cholinv.info(-1, 1, -5, 'U')
        t = timeit(lambda: cholinv.factor(A, pack, None))
        print("N=%d complete_inv=-1: %.1f ms  %.2f TF" % (n, t * 1e3, n ** 3 / 3 / t / 1e12), flush=True)
        del pack
        for ci in (0, 1):
            true_fl = n ** 3 / 3.0 + (n ** 3 / 12.0 if ci == 0 else n ** 3 / 3.0)
            for label, opts in (("recursion (inv_fast=0)", {"inv_fast": 0}),
                                ("tree after the sweep (inv_overlap=0)", {"inv_overlap": 0}),
                                ("tree overlapped, default start", {}),
                                ("tree overlapped from panel 0", {"inv_start_m": 1 << 40}),
                                ("tree overlapped, start at n/4 left", {"inv_start_m": n // 4})):
                pack = cholinv.info(ci, 1, -5, 'U')
                for k, v in opts.items():
                    pack.set_option(k, v)
                t = timeit(lambda: cholinv.factor(A, pack, None))
                res = validate.cholesky.residual(A, pack)
                print("N=%d complete_inv=%d %-40s %.1f ms  %.2f TF on N^3/3  %.2f TF true (%.0f%% of 78.6)  residual %.2e"
                      % (n, ci, label + ":", t * 1e3, n ** 3 / 3 / t / 1e12, true_fl / t / 1e12, 100 * true_fl / t / 1e12 / 78.6, res), flush=True)
                del pack
                torch.cuda.empty_cache()
        del A
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
