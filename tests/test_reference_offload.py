"""libcapital_amd_cblas.so (include/capital_amd_cblas.h): the seven CBLAS / LAPACKE symbols the reference imports, on host pointers, served
by the library's operators.  On the CPU (this file, no GPU): the product's object file of it linked against the recording stand-in in
compute mode - (1) every entry point against NumPy with BLAS / LAPACK's own conventions (leading dimensions, the untouched triangle, C
unread when beta = 0, info), (2) the REAL reference - oracle/_ref/*_cap: its unmodified sources with this library in MKL's place - running
cholinv, CholeskyQR2 and SUMMA on 1 ... 8 MPI ranks: its own validators' residuals, and its dumps against the MKL-linked build's.
tests/test_zz_reference_offload_gpu.py runs the same on the device."""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
MPIEXEC = "/opt/conda/bin/mpiexec"
SYMBOLS = ("cblas_dgemm", "cblas_dtrmm", "cblas_dsyrk", "LAPACKE_dpotrf", "LAPACKE_dtrtri", "LAPACKE_dgeqrf", "LAPACKE_dorgqr", "capcb_counters", "capcb_release")
COL, NT, TR, UP, LO, NONUNIT, LEFT, RIGHT = 102, 111, 112, 121, 122, 131, 141, 142


def test_the_library_builds_and_exports_what_its_header_declares():
    from capital_amd import build
    lib = build.build_cblas(verbose=False)
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert exported == set(SYMBOLS), exported ^ set(SYMBOLS)
    header = open(os.path.join(ROOT, "include", "capital_amd_cblas.h")).read()
    code = "import ctypes as C; L = C.CDLL(%r); [getattr(L, s) for s in %r]; print('loaded')" % (lib, SYMBOLS)     # loads (no compute call: no GPU here)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "loaded" in r.stdout, r.stderr[-2000:]
    for s in SYMBOLS:
        assert s + "(" in header, s
    compat = open(os.path.join(ROOT, "include", "for_upstream", "mkl.h")).read()      # the stand-in for "mkl.h" the reference is compiled against
    for s in SYMBOLS[:7]:
        assert s + "(" in compat, s
    # the main library does not export BLAS names: a process that also holds a CPU BLAS keeps it
    main = subprocess.check_output(["nm", "-D", "--defined-only", build.LIB], text=True)
    assert "cblas_" not in main and "LAPACKE_" not in main
    # no CPU arithmetic behind the seam: the library needs libcapital_amd.so and the HIP runtime, nothing else that computes
    needed = subprocess.check_output(["readelf", "-d", lib], text=True)
    assert "libcapital_amd.so" in needed and "libamdhip64" in needed and "mkl" not in needed and "blas" not in needed.replace("cblas.so", "")


@pytest.fixture(scope="module")
def standin():
    """the offload library over the CPU stand-in (compute mode), loaded into this process"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipshim"))
    import build_shim
    dst = build_shim.build_cblas()
    shim = C.CDLL(os.path.join(build_shim.OUT, "libhipshim.so"), mode=C.RTLD_GLOBAL)
    C.CDLL(os.path.join(build_shim.OUT, "libcapital_amd_shim.so"), mode=C.RTLD_GLOBAL)
    L = C.CDLL(dst)
    shim.shim_set_compute(1)
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f(a):
    return np.asfortranarray(a)


def exercise(L, rng, sizes, every_form=True):
    """every entry point against NumPy, on host memory with leading dimensions larger than the windows - shared with the GPU test
    (every_form = False: SYRK only as Upper / Trans and TRMM without Right / Trans - the forms the reference issues and the -m gpu
    operator tests run)"""
    worst = {}

    def rel(x, ref):
        return float(np.linalg.norm(x - ref) / max(np.linalg.norm(ref), 1e-300))
    d = C.c_double
    for (m, n, k) in sizes:
        for ta in (NT, TR):
            for tb in (NT, TR):
                a = _f(np.full((m + 3, k + 2) if ta == NT else (k + 3, m + 2), np.nan)); b = _f(np.full((k + 1, n + 2) if tb == NT else (n + 1, k + 2), np.nan))
                av = a[:m, :k] if ta == NT else a[:k, :m]; bv = b[:k, :n] if tb == NT else b[:n, :k]
                av[:] = rng.standard_normal(av.shape); bv[:] = rng.standard_normal(bv.shape)
                opa = av if ta == NT else av.T; opb = bv if tb == NT else bv.T
                for beta in ((0.0, -0.5, 1.0) if every_form else (0.0, 1.0)):
                    c = _f(np.full((m + 5, n + 1), 7.25)); c0 = rng.standard_normal((m, n))
                    c[:m, :n] = np.nan if beta == 0.0 else c0                        # beta = 0: C must not be read
                    L.cblas_dgemm(COL, ta, tb, m, n, k, d(1.5), _p(a), a.shape[0], _p(b), b.shape[0], d(beta), _p(c), c.shape[0])
                    worst["dgemm"] = max(worst.get("dgemm", 0), rel(c[:m, :n], 1.5 * opa @ opb + (beta * c0 if beta else 0)))
                    assert np.all(c[m:, :] == 7.25) and np.all(c[:, n:] == 7.25)       # nothing outside the window
        for tr in ((NT, TR) if every_form else (TR,)):
            a = _f(np.full((n + 2, k + 1) if tr == NT else (k + 2, n + 1), np.nan)); av = a[:n, :k] if tr == NT else a[:k, :n]
            av[:] = rng.standard_normal(av.shape)
            g = av @ av.T if tr == NT else av.T @ av
            for uplo in ((UP, LO) if every_form else (UP,)):
                for beta in (0.0, 1.0):
                    c0 = rng.standard_normal((n, n)); c = _f(np.full((n + 4, n), 3.5)); c[:n] = c0
                    if beta == 0.0 and every_form:                               # beta = 0: the triangle that is written is not read first
                        c[:n][np.triu_indices(n) if uplo == UP else np.tril_indices(n)] = np.nan
                    L.cblas_dsyrk(COL, uplo, tr, n, k, d(-1.0), _p(a), a.shape[0], d(beta), _p(c), c.shape[0])
                    tri = np.triu if uplo == UP else np.tril
                    other = (lambda x: np.tril(x, -1)) if uplo == UP else (lambda x: np.triu(x, 1))
                    worst["dsyrk"] = max(worst.get("dsyrk", 0), rel(tri(c[:n]), tri(-g + beta * c0)))
                    assert np.array_equal(other(c[:n]), other(c0)) and np.all(c[n:] == 3.5)   # the other triangle comes back untouched
        for side in (LEFT, RIGHT):
            t = m if side == LEFT else n
            tm = _f(np.full((t + 2, t), np.nan)); tv = np.linalg.cholesky(np.atleast_2d(np.cov(rng.standard_normal((t, 2 * t + 8))))).T * 3.0
            tm[:t] = tv + np.tril(np.full((t, t), np.nan), -1)                          # NaNs below the diagonal: never referenced
            for tr in (NT, TR):
                if not every_form and side == RIGHT and tr == TR:
                    continue                                                         # (the reference issues the other three: cholinv.hpp:114-154)
                b = _f(np.full((m + 1, n + 3), 9.0)); b0 = rng.standard_normal((m, n)); b[:m, :n] = b0
                L.cblas_dtrmm(COL, side, UP, tr, NONUNIT, m, n, d(0.75), _p(tm), tm.shape[0], _p(b), b.shape[0])
                op = tv.T if tr == TR else tv
                worst["dtrmm"] = max(worst.get("dtrmm", 0), rel(b[:m, :n], 0.75 * (op @ b0 if side == LEFT else b0 @ op)))
                assert np.all(b[m:] == 9.0) and np.all(b[:, n:] == 9.0)
        s = np.atleast_2d(np.cov(rng.standard_normal((n, 2 * n + 8)))) + 0.1 * np.eye(n)
        a = _f(np.full((n + 3, n), 1.25)); a[:n] = np.triu(s) + np.tril(np.full((n, n), -77.0), -1)
        L.LAPACKE_dpotrf.restype = C.c_int; L.LAPACKE_dtrtri.restype = C.c_int
        assert L.LAPACKE_dpotrf(COL, C.c_char(b"U"), n, _p(a), a.shape[0]) == 0
        r = np.linalg.cholesky(s).T
        worst["dpotrf"] = max(worst.get("dpotrf", 0), rel(np.triu(a[:n]), r))
        assert np.all(np.tril(a[:n], -1) == np.tril(np.full((n, n), -77.0), -1)) and np.all(a[n:] == 1.25)
        assert L.LAPACKE_dtrtri(COL, C.c_char(b"U"), C.c_char(b"N"), n, _p(a), a.shape[0]) == 0
        worst["dtrtri"] = max(worst.get("dtrtri", 0), rel(np.triu(a[:n]) @ r, np.eye(n)))
        assert np.all(np.tril(a[:n], -1) == np.tril(np.full((n, n), -77.0), -1))
    # LAPACK's info: the first non-positive pivot; a zero on the diagonal of a triangle; arguments this library does not take
    n = sizes[0][1]
    s = np.cov(rng.standard_normal((n, 2 * n + 8))) + 0.1 * np.eye(n); s[n // 2, n // 2] = -1.0
    a = _f(np.triu(s))
    assert L.LAPACKE_dpotrf(COL, C.c_char(b"U"), n, _p(a), n) == n // 2 + 1
    t = _f(np.triu(rng.standard_normal((n, n))) + 4 * np.eye(n)); t[3, 3] = 0.0
    assert L.LAPACKE_dtrtri(COL, C.c_char(b"U"), C.c_char(b"N"), n, _p(t), n) == 4
    assert L.LAPACKE_dpotrf(COL, C.c_char(b"L"), n, _p(a), n) == -2 and L.LAPACKE_dpotrf(101, C.c_char(b"U"), n, _p(a), n) == -1
    assert L.LAPACKE_dtrtri(COL, C.c_char(b"U"), C.c_char(b"U"), n, _p(t), n) == -3
    L.LAPACKE_dgeqrf.restype = C.c_int
    assert L.LAPACKE_dgeqrf(COL, n, n, _p(a), n, _p(a)) == -1010
    if every_form:
        # BLAS's degenerate scalars: k = 0 or alpha = 0 scale C (A and B are not referenced - NaNs in them must not matter); TRMM with alpha = 0 zeroes B
        m, n, k = 50, 40, 30
        c0 = _f(rng.standard_normal((m, n))); nan_a, nan_b = _f(np.full((m, k), np.nan)), _f(np.full((k, n), np.nan))
        for kk, alpha, beta in ((0, 1.0, -0.5), (0, 1.0, 0.0), (k, 0.0, 2.0)):
            c = c0.copy(order="F")
            L.cblas_dgemm(COL, NT, NT, m, n, kk, d(alpha), _p(nan_a), m, _p(nan_b), k, d(beta), _p(c), m)
            assert np.array_equal(c, beta * c0), (kk, alpha, beta)
        cs = _f(rng.standard_normal((n, n))); c = cs.copy(order="F")
        L.cblas_dsyrk(COL, UP, TR, n, k, d(0.0), _p(nan_b), k, d(0.5), _p(c), n)
        assert np.array_equal(np.triu(c), np.triu(0.5 * cs)) and np.array_equal(np.tril(c, -1), np.tril(cs, -1))
        b = c0.copy(order="F")
        L.cblas_dtrmm(COL, LEFT, UP, NT, NONUNIT, m, n, d(0.0), _p(_f(np.full((m, m), np.nan))), m, _p(b), m)
        assert not b.any()
        # an illegal argument: a line on stderr, the call ignored (xerbla's way) - the window stays as it was
        c = c0.copy(order="F")
        L.cblas_dgemm(COL, NT, NT, m, n, k, d(1.0), _p(nan_a), m - 1, _p(nan_b), k, d(0.0), _p(c), m)
        L.cblas_dtrmm(COL, LEFT, 99, NT, NONUNIT, m, n, d(1.0), _p(nan_a), m, _p(c), m)        # uplo = 99 is no BLAS value at all
        assert np.array_equal(c, c0)
        # ConjTrans is Trans in real arithmetic
        a2, b2 = _f(rng.standard_normal((k, m))), _f(rng.standard_normal((k, n))); c = c0.copy(order="F")
        L.cblas_dgemm(COL, 113, NT, m, n, k, d(1.0), _p(a2), k, _p(b2), k, d(0.0), _p(c), m)
        assert np.linalg.norm(c - a2.T @ b2) <= 1e-13 * np.linalg.norm(c)
    calls, bi, bo = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
    L.capcb_counters(C.byref(calls), C.byref(bi), C.byref(bo))
    assert calls.value > 0 and bi.value > 0 and bo.value > 0
    L.capcb_release()                                                   # buffers back; the next call allocates again
    one = _f(np.array([[2.0]])); L.cblas_dgemm(COL, NT, NT, 1, 1, 1, d(1.0), _p(one), 1, _p(one), 1, d(0.0), _p(one), 1)
    assert one[0, 0] == 4.0
    return worst


def test_every_entry_point_against_numpy_with_blas_conventions(standin):
    worst = exercise(standin, np.random.default_rng(3), [(70, 40, 33), (130, 129, 64), (257, 96, 300), (1, 1, 1), (5, 300, 2)])
    assert set(worst) == {"dgemm", "dsyrk", "dtrmm", "dpotrf", "dtrtri"}
    assert max(worst.values()) < 5e-14, worst


def test_a_legal_blas_form_the_library_does_not_implement_ends_the_process_loudly(standin):
    """a Lower / Unit TRMM or a row-major call is LEGAL BLAS; the offload library used to print one line and return with the
    output untouched - a program linked with it in MKL's place would have computed on with wrong numbers.  Now it aborts with the reason.
    (Run in a child process under the same CPU stand-in the fixture installs.)"""
    import subprocess, sys
    code = ("import ctypes as C, numpy as np, os, sys\n"
            "L = C.CDLL(os.environ['CAPCB_TEST_LIB'], mode=C.RTLD_GLOBAL)\n"
            "a = np.asfortranarray(np.eye(8)); b = np.asfortranarray(np.ones((8, 8)))\n"
            "p = lambda x: x.ctypes.data_as(C.c_void_p)\n"
            "form = sys.argv[1]\n"
            "if form == 'lower': L.cblas_dtrmm(102, 141, 122, 111, 131, 8, 8, C.c_double(1.0), p(a), 8, p(b), 8)\n"
            "if form == 'unit': L.cblas_dtrmm(102, 141, 121, 111, 132, 8, 8, C.c_double(1.0), p(a), 8, p(b), 8)\n"
            "if form == 'rowmajor': L.cblas_dgemm(101, 111, 111, 8, 8, 8, C.c_double(1.0), p(a), 8, p(b), 8, C.c_double(0.0), p(b), 8)\n"
            "print('returned')\n")
    env = dict(os.environ, CAPCB_TEST_LIB=standin._name)
    for form in ("lower", "unit", "rowmajor"):
        r = subprocess.run([sys.executable, "-c", code, form], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "returned" not in r.stdout, (form, r.returncode, r.stdout)
        assert "does not implement" in r.stderr, (form, r.stderr[-500:])


def build_and_run_demo(tmp_path, libdir, run_dirs, extra_env, n):
    """examples/cblas_offload_demo.c - a plain CBLAS / LAPACKE program (host arrays, MKL's argument lists) linked with -lcapital_amd_cblas"""
    import re
    exe = str(tmp_path / "cblas_offload_demo")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-O1", os.path.join(ROOT, "examples", "cblas_offload_demo.c"), "-I" + os.path.join(ROOT, "include", "for_upstream"),
           "-L" + libdir, "-lcapital_amd_cblas", "-Wl,--allow-shlib-undefined", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ); env.pop("LD_PRELOAD", None); env["LD_LIBRARY_PATH"] = ":".join(run_dirs); env.update(extra_env)
    run = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
    m = re.search(r"residual = (\S+) inverse = (\S+) calls = (\d+)", run.stdout)
    assert m and float(m.group(1)) < 1e-14 and float(m.group(2)) < 1e-13 and int(m.group(3)) == 5, run.stdout


def test_a_plain_cblas_program_runs_on_the_library(tmp_path):
    import shutil
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipshim"))
    import build_shim
    build_shim.build_cblas()
    d = os.path.join(build_shim.OUT, "cblas")
    build_and_run_demo(tmp_path, d, [d, build_shim.OUT], {"SHIM_COMPUTE": "1"}, 901)


def cap_env(libdirs):
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1", CAPCB_REPORT="1")
    env.pop("LD_PRELOAD", None)
    # (the system's libstdc++ in front of conda's older one, which the MPI launcher's rpath would otherwise pick for the whole process)
    env["LD_LIBRARY_PATH"] = ":".join(["/usr/lib/x86_64-linux-gnu"] + list(libdirs))
    return env


def reference_available():
    return all(os.path.exists(os.path.join(REFDIR, b)) for b in ("cholinv_cap", "cacqr_cap", "summa_cap", "cholinv_ref")) and os.path.exists(MPIEXEC)


def run_reference(env, exe, ranks, argv, timeout=600):
    import signal
    cmd = [MPIEXEC, "-n", str(ranks), os.path.join(REFDIR, exe)] + [str(a) for a in argv]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    try:
        so, se = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)             # the launcher AND its ranks: nothing of a stuck run stays on the device
        except OSError:
            pass
        so, se = p.communicate()
        raise AssertionError((exe, argv, "no result within %d s" % timeout, so[-1000:], se[-1000:]))
    r = subprocess.CompletedProcess(cmd, p.returncode, so, se)
    assert r.returncode == 0, (exe, argv, r.stdout[-1500:], r.stderr[-1500:])
    run_reference.last_output = r.stdout + r.stderr
    import re
    line = [l for l in r.stdout.splitlines() if "ranks=" in l]
    kv = {}
    for k, v in (re.findall(r"(\w+)=(\S+)", line[-1]) if line else []):
        try:
            kv[k] = float(v)
        except ValueError:
            pass
    served = [int(x) for x in re.findall(r"capital_amd_cblas: (\d+) calls served", r.stderr)]
    return kv, served


# (exe, ranks, argv without the dump file, what the reference's own validator must print)
REFERENCE_RUNS = [
    ("cholinv", 1, (512, 1, 1, -2, 0, 0, 0), {"residual": 1e-14}),
    ("cholinv", 1, (1000, 0, 2, -3, 0, 0, 0), {"residual": 1e-14}),
    ("cholinv", 8, (512, 1, 1, -2, 0, 0, 1), {"residual": 1e-14}),
    ("cholinv", 8, (1001, 0, 1, 0, 0, 0, 1), {"residual": 1e-14}),
    ("cholinv", 8, (768, 0, 2, -2, 0, 0, 2), {"residual": 1e-14}),          # ReplicateComp base case
    ("cacqr", 1, (2, 4096, 96, 1, 1, 1, 0), {"residual": 1e-13, "orthogonality": 1e-14}),
    ("cacqr", 4, (2, 4096, 64, 1, 1, 1, 0), {"residual": 1e-13, "orthogonality": 1e-14}),
    ("cacqr", 8, (2, 2050, 64, 2, 1, 1, 0), {"residual": 1e-13, "orthogonality": 1e-14}),
    ("cacqr", 8, (2, 2048, 64, 2, 0, 1, -1), {"residual": 1e-13, "orthogonality": 1e-14}),     # the Gram matrix's Cholesky with complete_inv = 0
]


def dump_arrays(exe, ranks, argv, raw):
    """the arrays of one rank's dump file (oracle/ref/drv_*.cpp): cholinv A, R, R^-1; cacqr A, Q, R; summa operands and result"""
    f8 = lambda b: np.frombuffer(b, dtype=np.float64)
    i8 = lambda b: [int(v) for v in np.frombuffer(b, dtype=np.int64)]
    if exe == "cholinv":
        body = f8(raw if ranks == 1 else raw[64:])
        return [body[i * (body.size // 3):(i + 1) * (body.size // 3)] for i in range(3)]
    if exe == "cacqr":
        if ranks == 1:
            m, n = int(argv[1]), int(argv[2])
            body = f8(raw); return [body[:m * n], body[m * n:2 * m * n], body[2 * m * n:]]
        h = i8(raw[:80]); body = f8(raw[80:]); a = h[6] * h[7]
        return [body[:a], body[a:2 * a], body[2 * a:]]
    h = i8(raw[:64]); off = 64; out = []
    for _ in range(h[6]):
        rows, cols, packed = i8(raw[off:off + 24]); off += 24
        cnt = cols * (cols + 1) // 2 if packed else rows * cols
        out.append(f8(raw[off:off + 8 * cnt])); off += 8 * cnt
    return out


def dumps_equal(exe, ranks, argv, env_cap, tol, build="cap"):
    """the same run by the MKL-linked build and by the build on this library: every rank's dump file, array by array (the worst relative
    difference over the arrays: inputs must be identical, R / R^-1 / Q / the products equal to rounding)"""
    with tempfile.TemporaryDirectory() as td:
        out = {}
        for tag, env in (("ref", dict(os.environ, MKL_NUM_THREADS="1")), (build, env_cap)):
            dump = os.path.join(td, tag + ".bin")
            a = list(argv) + [dump] + ([1] if exe != "summa" else [])
            kv, _ = run_reference(env, exe + "_" + tag, ranks, a)
            if tag == "ref" and ("MKL ERROR" in run_reference.last_output or not all(kv.get(k, 0.0) < 1e-8 for k in ("residual", "orthogonality"))):
                return None                 # upstream itself handed BLAS an illegal argument (a 0-column split piece), or its own validator
                                            # says the MKL run is no factorization (a rank-deficient random input): nothing to compare
            files = [dump] if os.path.exists(dump) else ["%s.%d" % (dump, q) for q in range(ranks)]
            out[tag] = [open(f, "rb").read() for f in files]
        worst = 0.0
        for x, y in zip(out["ref"], out[build]):
            assert len(x) == len(y)
            ax, ay = dump_arrays(exe, ranks, argv, x), dump_arrays(exe, ranks, argv, y)
            assert np.array_equal(ax[0], ay[0])                                 # the input
            for a, b in zip(ax, ay):
                if a.size:
                    worst = max(worst, float(np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-300)))
        assert worst < tol, (exe, ranks, argv, worst)
        return worst


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/*_cap (the reference built from /root/reference on this library) or mpiexec is not here")
def test_the_real_reference_runs_on_the_library_and_passes_its_own_validators():
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipshim"))
    import build_shim
    build_shim.build_cblas()
    env = cap_env([os.path.join(build_shim.OUT, "cblas"), build_shim.OUT]); env["SHIM_COMPUTE"] = "1"
    for exe, ranks, argv, checks in REFERENCE_RUNS:
        kv, served = run_reference(env, exe + "_cap", ranks, list(argv) + ["-", 1])
        for k, tol in checks.items():
            assert kv[k] < tol, (exe, ranks, argv, kv)
        assert len(served) == ranks and min(served) > 0, (exe, ranks, served)      # every rank's BLAS / LAPACK calls went through the library
    # the same numbers as with MKL behind the seam, array by array (R, R^-1 / Q, R / the products)
    assert dumps_equal("cholinv", 1, (300, 1, 1, -2, 0, 0, 0), env, 1e-12) > 0
    dumps_equal("cholinv", 8, (300, 0, 1, -2, 0, 0, 1), env, 1e-12)
    dumps_equal("cacqr", 8, (2, 600, 48, 2, 1, 1, 0), env, 1e-12)
    for op, m, n, k in ((0, 150, 130, 170), (2, 140, 90, 0), (3, 140, 90, 0), (5, 0, 100, 160)):
        dumps_equal("summa", 8, (op, m, n, k, 2, 0, 2, 1.5, -0.5 if op in (0, 5) else 0.0), env, 1e-13)


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/*_cap or mpiexec is not here")
def test_random_configurations_of_the_reference_on_mkl_and_on_the_library():
    """tests/hipshim/fuzz_offload.py: the real reference twice per random configuration (cholinv with every working base-case policy, CholeskyQR2
    1D / 3D with random options for its inner Cholesky, SUMMA's overloads; 1 ... 16 ranks) - once on MKL, once on this library over the CPU
    stand-in: every rank's dump equal array by array (1000 configurations agreed when this was written; 30 with a fixed seed here)"""
    env = dict(os.environ); env.pop("LD_PRELOAD", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipshim", "fuzz_offload.py"), "11", "30"], capture_output=True, text=True, timeout=1200, env=env)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0 and lines[-1] == "30 configurations, 0 with findings", "\n".join(l for l in lines if not l.startswith("ok"))[-3000:] + r.stderr[-2000:]
    assert {l.split()[1] for l in lines if l.startswith("ok")} == {"cholinv", "cacqr", "summa"}


def test_integration_section_A_shows_the_files_that_are_compiled():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for f in ("blas_interface_double.inc", "lapack_interface_double.inc"):
        text = open(os.path.join(ROOT, "examples", "engine_binding", f)).read()
        assert "```cpp\n" + text + "```" in doc, f


@pytest.mark.skipif(not (reference_available() and os.path.exists(os.path.join(REFDIR, "cholinv_engine"))),
                    reason="oracle/_ref/*_engine (INTEGRATION.md section A pasted into the reference) or mpiexec is not here")
def test_integration_section_A_pasted_into_the_reference_compiles_and_runs():
    """oracle/ref/build_ref.py cut upstream's double specialisations of blas::engine / lapack::engine out of its copy of the reference and
    pasted examples/engine_binding/*.inc (= INTEGRATION.md section A) in their place: the reference runs on the stand-in (host memory is
    device memory there), its validators pass, and no call went through the CBLAS library that is linked beside it"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipshim"))
    import build_shim
    env = cap_env([build_shim.build_engine_dir(), build_shim.OUT]); env["SHIM_COMPUTE"] = "1"
    for exe, ranks, argv, checks in REFERENCE_RUNS:
        kv, served = run_reference(env, exe + "_engine", ranks, list(argv) + ["-", 1])
        for k, tol in checks.items():
            assert kv[k] < tol, (exe, ranks, argv, kv)
        assert served == [0] * ranks, (exe, ranks, served)


@pytest.mark.skipif(not os.path.exists("/opt/conda/lib/libmkl_rt.so"), reason="MKL (the BLAS the reference is linked with) is not here")
def test_random_blas_and_lapack_calls_behave_like_mkl():
    """tests/hipshim/fuzz_cblas.py: the same random call (ragged shapes, padded leading dimensions, every transpose / side / triangle,
    alpha / beta incl. 0 and 1, NaNs wherever BLAS promises not to look, non-SPD matrices, now and then an illegal argument) through MKL and
    through this library over the stand-in: same output window, same NaN pattern, same info (5500 calls agreed when this was written)"""
    env = dict(os.environ, MKL_NUM_THREADS="1"); env.pop("LD_PRELOAD", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipshim", "fuzz_cblas.py"), "9", "250"], capture_output=True, text=True, timeout=900, env=env)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0 and lines[-1] == "250 calls, 0 with findings", "\n".join(l for l in lines if l.startswith("BAD"))[-3000:] + r.stderr[-1500:]
    assert {l.split()[1] for l in lines if l.startswith("ok")} >= {"dgemm", "dsyrk", "dtrmm", "potrf", "trtri"}


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/*_cap or mpiexec is not here")
def test_every_mpi_rank_of_the_reference_gets_a_gpu_of_its_own():
    """one process per GPU without a line of code in the MPI program: with several devices visible (SHIM_DEVICES = 8 on the stand-in) the offload
    library takes the launcher's local rank modulo the device count, CAPCB_DEVICE names a device, one visible device is left alone"""
    import re
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipshim"))
    import build_shim
    build_shim.build_cblas()
    base = cap_env([os.path.join(build_shim.OUT, "cblas"), build_shim.OUT]); base["SHIM_COMPUTE"] = "1"
    for extra, want in (({"SHIM_DEVICES": "8"}, list(range(8))), ({"SHIM_DEVICES": "4"}, [0, 0, 1, 1, 2, 2, 3, 3]),
                        ({"SHIM_DEVICES": "4", "CAPCB_DEVICE": "3"}, [3] * 8), ({}, [0] * 8)):
        kv, served = run_reference(dict(base, **extra), "cholinv_cap", 8, [256, 1, 1, -2, 0, 0, 1, "-", 1])
        assert kv["residual"] < 1e-14 and len(served) == 8
        got = sorted(int(x) for x in re.findall(r"bytes device -> host, device (\d+)", run_reference.last_output))
        assert got == want, (extra, got)


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref or mpiexec is not here")
def test_the_mkl_linked_reference_runs_on_the_library_when_it_is_preloaded():
    """not even a relink: LD_PRELOAD=libcapital_amd_cblas.so in front of the build that is linked with MKL - the seven symbols resolve to this
    library, every rank's BLAS / LAPACK calls are served by it, the reference's validator passes"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipshim"))
    import build_shim
    lib = build_shim.build_cblas()
    env = cap_env([os.path.join(build_shim.OUT, "cblas"), build_shim.OUT]); env["SHIM_COMPUTE"] = "1"; env["LD_PRELOAD"] = lib
    for exe, ranks, argv, checks in (("cholinv", 8, (512, 0, 1, -2, 0, 0, 1), {"residual": 1e-14}),
                                     ("cacqr", 8, (2, 1024, 64, 2, 1, 1, 0), {"residual": 1e-13, "orthogonality": 1e-14})):
        kv, served = run_reference(env, exe + "_ref", ranks, list(argv) + ["-", 1])
        for k, tol in checks.items():
            assert kv[k] < tol, (exe, kv)
        assert sorted(served)[-ranks] > 0, served            # (the launcher's own processes inherit the preload and report 0 calls)
