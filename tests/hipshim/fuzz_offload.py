"""TEST INFRASTRUCTURE: the REAL reference twice on the same random configuration - linked with MKL (oracle/_ref/*_ref) and linked with
libcapital_amd_cblas.so over the CPU stand-in (oracle/_ref/*_cap, compute mode): every BLAS / LAPACK call upstream's recursion issues
(ragged, tiny, odd leading dimensions, triangular operands with junk below the diagonal) goes through the library's operator seam, and every
rank's dump must come out the same.  python tests/hipshim/fuzz_offload.py SEED COUNT"""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(seed, count):
    import build_shim
    import test_reference_offload as tro
    build_shim.build_cblas()
    env = tro.cap_env([os.path.join(build_shim.OUT, "cblas"), build_shim.OUT]); env["SHIM_COMPUTE"] = "1"
    # every third configuration also through the third build: INTEGRATION.md section A pasted over upstream's engine specialisations
    env_engine = tro.cap_env([build_shim.build_engine_dir(), build_shim.OUT]); env_engine["SHIM_COMPUTE"] = "1"
    have_engine = os.path.exists(os.path.join(tro.REFDIR, "cholinv_engine"))
    rng = random.Random(seed)
    bad = 0
    for i in range(count):
        u = rng.random()
        if u < 0.5:
            ranks = rng.choice([1, 8])
            n = rng.choice([rng.randint(1, 40), rng.randint(40, 500), 64 * rng.randint(1, 10)])
            pol = rng.choice([1, 2]) if ranks > 1 else rng.choice([0, 3])
            exe, argv = "cholinv", (n, rng.choice([0, 1]), rng.choice([1, 1, 2, 3]), rng.choice([1, 0, -1, -2, -3, -5]), 0, 0, pol)
        elif u < 0.8:
            c, ranks = rng.choice([(1, 1), (1, 2), (1, 3), (1, 8), (2, 8), (2, 16)])
            n = c * rng.randint(1, 64 // c)
            m = n + rng.choice([0, 3, rng.randint(10, 800)])
            exe, argv = "cacqr", (2, m, n, c, rng.choice([0, 1]), rng.choice([1, 2]), rng.choice([0, -1, -2]))
        else:
            c = rng.choice([1, 2]); ranks = c ** 3
            op = rng.randint(0, 7)
            m, n, k = (rng.choice([rng.randint(1, 20), rng.randint(20, 300)]) for _ in range(3))
            exe, argv = "summa", (op, m, n, k, c, 0, rng.choice([0, 2, 3]), rng.choice([1.0, -0.5]), rng.choice([0.0, 1.0]) if op in (0, 5, 6, 7) else 0.0)
        try:
            # tolerance: the two builds sum in different orders; ill-conditioned random squares (M = N CholeskyQR) amplify that
            w = tro.dumps_equal(exe, ranks, argv, env, 1e-9 if exe == "cacqr" else 1e-11)
            if have_engine and i % 3 == 0 and w is not None:
                w = max(w, tro.dumps_equal(exe, ranks, argv, env_engine, 1e-9 if exe == "cacqr" else 1e-11, build="engine"))
            print("ok   %s ranks=%d %s  %s" % (exe, ranks, argv, "upstream's own run is not a valid one here (illegal BLAS argument / singular input)" if w is None else "%.1e" % w), flush=True)
        except AssertionError as e:
            bad += 1
            print("BAD  %s ranks=%d %s  %s" % (exe, ranks, argv, str(e)[:400]), flush=True)
    print("%d configurations, %d with findings" % (count, bad))
    return bad


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    sys.exit(1 if main(int(sys.argv[1]), int(sys.argv[2])) else 0)
