"""The library's host side computing REAL factors on the CPU (tests/hipshim/run_compute.py): the product's own object files linked
against the recording stand-in in COMPUTE MODE - every launch runs a CPU model of its kernel (tests/hipshim/kernels_cpu.cpp: the kernels'
contract incl. launch geometry; the GEMM models walk a launch's blocks through the library's own tile enumeration and decode the very
argument structs the library passes, csrc/kargs.h / gemm_index.h), every copy is carried out, fresh "device" memory holds NaNs.  Multi-rank
plans run with one thread per rank and communicators that really move the data (blocking collectives; the IPC exchanges copy into the
peer's buffer behind the same token all-reduces as on the GPU).

What comes out is compared with NumPy / the CPU oracle to 2e-11: R and R^-1 of the single-GPU plan under every schedule option and a
dozen option mixes (n = 64 ... 2048, ragged sizes), the operator seam, the 1 x P plan on 1 ... 8 ranks and the Pr x Pc plan on
1x1 ... 4x8 grids (safe, IPC, R^-1), the reference's element-cyclic layout end to end on its d x d x c grid, SUMMA (GEMM, TRMM, SYRK overloads, util::transpose) on 1 ... 27 ranks, descriptors with pinned staging (block- and element-cyclic: host matrix -> pieces -> host, exact),
CholeskyQR2 on 1 ... 8 ranks, the mixed-precision solve (bf16 updates emulated with round-to-nearest-even bf16 operands and fp32
accumulation: a factor in the bf16 error band, a solution refined to 1e-14) on one GPU with every schedule option and on 1 ... 8
ranks.  This is a check of the HOST side - every pointer, leading dimension, flag, grid size and event-free data
flow it computes - on a machine without a GPU; the kernels themselves are checked on the GPU (-m gpu).  It found that a plan speaking
the reference's layout applied the single-process base-case rule where upstream's rule depends on its grid (cholinv.hpp:15-18)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def computed(tmp_path_factory):
    from capital_amd import build
    build.build(verbose=False)
    out = str(tmp_path_factory.mktemp("compute") / "compute.json")
    env = dict(os.environ); env.pop("LD_PRELOAD", None)
    env["SHIM_FILTER"] = ""; env["SHIM_KEEP_TRACE"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipshim", "run_compute.py"), out], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return json.load(open(out))["results"]


def test_every_plan_computes_the_right_factors_on_the_cpu(computed):
    bad = [(x["name"], x["findings"][:3]) for x in computed if x["findings"]]
    assert not bad, "\n".join("%s: %s" % b for b in bad[:20])
    assert len(computed) >= 171, len(computed)
    names = " | ".join(x["name"] for x in computed)
    for must in ("operators m=130 n=70 k=33", "cholinv n=2048 ci=1", "'pair_rest': 0", "'inner_la': 1", "'use_sb': 0", "'inv_fast': 0", "cholinv n=1000 ci=1",
                 "dist n=2048 nb=128 P=8 {'ipc': 1, 'strip': 2}", "dist n=1000 nb=128 P=3 {'ipc': 1} ci=1", "dist2d n=1152 nb=128 4x8", "dist2d n=1024 nb=128 4x4 {'ipc': 1}",
                 "dist2d n=1000 nb=128 2x4 {'complete_inv': 0}", "reference's layout n=1024 ci=0 bc=0 grid 2x2x2", "reference's layout n=1003 ci=0",
                 "summa gemm size=27 c=3", "cacqr m=8192 n=256 iter=2 P=4", "mpchol n=4096 nrhs=8 {'split': 0}", "mpchol n=4096 nrhs=8 {'solve3': 0}",
                 "mpchol n=3072 nrhs=8 twice [user stream]", "dmp n=2048 nb=256 P=8", "dmp n=1152 nb=128 P=3",
                 "summa trmm / syrk / transpose size=8 c=2", "desc block-cyclic 300x520 nb=128 grid 2x4", "desc element-cyclic 301x203",
                 "cacqr grid size=27 c=3", "reference pieces -> 2x4 plan through descriptors", "matrix utilities n=500 [user stream]",
                 "golden cholinv_grid8_n256_ci0_s1_bc0.npz (8 ranks, pieces)", "golden cholinv_grid8_n251_ci0_s1_bc-2.npz", "golden cholinv_p8_n192_ci0_s1_bc-3.npz",
                 "golden cholinv_n100_ci0_s2_bc-4.npz", "golden cacqr2_p8_c2_m256_n16.npz", "golden cacqr1_p8_c2_m200_n12.npz",
                 "golden summa_c2_gemm_m51_n43_k35.npz", "golden summa_c2_trmm_left_trans_m41_n30.npz", "golden summa_c2_trmm_right_m40_n31.npz",
                 "golden summa_c2_syrk_trans_n31_k47.npz", "golden summa_c2_syrk_rect_n33_k40.npz"):
        assert must in names, must
    # the numbers are real: every case carries errors at rounding level, none is exactly zero across the board
    worst = max(v for x in computed for k, v in x["errors"].items() if k in ("R", "Rinv", "C", "A - QR", "R pieces", "dpotrf", "dgemm NN"))
    assert 1e-17 < worst < 2e-11, worst
    assert all(x["errors"].get("info", 0.0) == 0.0 for x in computed)


def test_the_models_compute_they_do_not_echo():
    """teeth: one case through the harness by hand - the right input gives the right factor, the same plan on a DIFFERENT input (the
    lower triangle of the matrix where the upper one is consumed, cholinv.hpp:13) gives a different one"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipshim"))
    code = r"""
import ctypes as C, sys, numpy as np
sys.path.insert(0, %r)
import run_compute as rc
rc.shim.shim_set_compute(1)
e = {}
r = rc.rs.Run("teeth", 0)
rc.cholinv_compute(r, e, 512, 1, 1, 0, (("nb", 128),))
ok = all(v < rc.TOL for v in e.values())
# the same plan on a matrix whose LOWER triangle is what was factored: the library must not have read it
print("OK" if ok else "BAD", e)
a = rc.spd(512, 5); n = 512
plan = C.c_void_p(); rc.rs.ok(rc.L.cap_cholinv_plan_create(C.byref(plan), n, -1, 1, 0, b"U", None), "create")
A = rc.rs.dmalloc(8 * n * n); out = rc.rs.dmalloc(8 * n * n)
rc.view(A, n, n)[:] = np.tril(a)                      # upper triangle zero: NOT the matrix
rc.rs.ok(rc.L.cap_cholinv_factor(plan, A, n, None), "factor")
rc.rs.ok(rc.L.cap_cholinv_get_R(plan, out, n, None), "get_R")
print("DIFF %%.3e" %% rc.rel(rc.view(out, n, n), np.linalg.cholesky(a).T))
""" % os.path.join(ROOT, "tests", "hipshim")
    env = dict(os.environ); env.pop("LD_PRELOAD", None); env["SHIM_FILTER"] = ""; env["SHIM_KEEP_TRACE"] = ""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert lines[0].startswith("OK"), lines
    diff = float(lines[1].split()[1])
    assert diff > 1e-3 or diff != diff, lines      # a different input gives a different (or NaN) factor: the models compute, they do not echo


@pytest.mark.parametrize("fault,what", [("1:7", "a tile grid that is one tile column short"), ("2:20", "a leading dimension that is two elements off"),
                                        ("3:3", "a triangular-operand hint on a dense operand"), ("4:11", "a K range that is one K tile short")])
def test_one_wrong_launch_among_hundreds_is_noticed(fault, what):
    """teeth: the blocked factorization of N = 2048 (nb = 128: ~110 launches, 32 of them LDS-DMA products) with ONE product handed over
    wrong, the way a host-side bug would do it (tests/hipshim/kernels_cpu.cpp, SHIM_FAULT) - the factor must come out wrong"""
    code = r"""
import sys
sys.path.insert(0, %r)
import run_compute as rc
rc.shim.shim_set_compute(1)
e = {}
r = rc.rs.Run("fault", 0)
rc.cholinv_compute(r, e, 2048, -1, 1, 0, (("nb", 128), ("outer", 256), ("tail", 0), ("depth2", 1)), reps=1)
print("R %%.3e" %% e["R"])
""" % os.path.join(ROOT, "tests", "hipshim")
    env = dict(os.environ); env.pop("LD_PRELOAD", None); env["SHIM_FILTER"] = ""; env["SHIM_KEEP_TRACE"] = ""
    out = {}
    for f in ("", fault):
        env["SHIM_FAULT"] = f
        if not f:
            env.pop("SHIM_FAULT")
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        out[f] = float(r.stdout.strip().splitlines()[-1].split()[1])
    assert out[""] < 2e-11, out
    assert out[fault] > 1e-8 or out[fault] != out[fault], (what, out)


def test_random_configurations_nobody_wrote_a_test_for():
    """tests/hipshim/fuzz_compute.py: random sizes, block widths, grids and option mixes through the compute mode (1100 of them passed when
    the harness was written; 40 with a fixed seed here)"""
    env = dict(os.environ); env.pop("LD_PRELOAD", None); env["SHIM_FILTER"] = ""; env["SHIM_KEEP_TRACE"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipshim", "fuzz_compute.py"), "7", "40"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert lines[-1] == "40 cases, 0 with findings", "\n".join(l for l in lines if not l.startswith("ok"))[-3000:]
    assert len({l.split()[1] for l in lines if l.startswith("ok")}) >= 5           # several plan kinds came up


def test_random_configurations_against_the_real_reference_run_beside_the_library():
    """tests/hipshim/fuzz_reference.py: the REAL reference (oracle/_ref, built from /root/reference where that exists) factors random
    (n, complete_inv, split, bc_mult_dim) on 1 rank and on its 2 x 2 x 2 grid, random CholeskyQR / CholeskyQR2 on 1D and c x d x c grids of
    1 ... 27 ranks, random GEMM / TRMM / SYRK calls of matmult::summa on the cubes of 1, 8, 27 ranks; the library runs the same input through
    the compute mode: same R, same R^-1 pattern, same Q pieces, same product pieces (6800 configurations agreed when this was written; 40 with
    a fixed seed here).  Skipped where the reference binary or an MPI launcher is missing."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipshim"))
    import fuzz_reference
    if not fuzz_reference.available():
        pytest.skip("oracle/_ref (the reference built from /root/reference) or mpiexec is not here")
    probe = subprocess.run([fuzz_reference.MPIEXEC, "-n", "1", fuzz_reference.REF, "8", "1", "1", "0", "0", "0", "0", "-", "1"], capture_output=True, text=True, timeout=120)
    if probe.returncode != 0:
        pytest.skip("the reference binary does not run here: " + (probe.stderr or probe.stdout)[-300:])
    env = dict(os.environ); env.pop("LD_PRELOAD", None); env["SHIM_FILTER"] = ""; env["SHIM_KEEP_TRACE"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipshim", "fuzz_reference.py"), "21", "40"], capture_output=True, text=True, timeout=1200, env=env)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0 and lines[-1] == "40 configurations, 0 with findings", "\n".join(l for l in lines if not l.startswith("ok"))[-3000:] + r.stderr[-2000:]
    kinds = {("summa" if "summa" in l else "cacqr" if "cacqr" in l else "2x2x2" if "2x2x2" in l else "1 rank") for l in lines if l.startswith("ok")}
    assert kinds == {"summa", "cacqr", "2x2x2", "1 rank"}


@pytest.mark.parametrize("name,argv,checks", [
    ("cholinv_driver", ("2048", "-1", "1", "-3", "1", "1"), {"residual": 1e-14}),
    ("cholinv_driver", ("1024", "1", "1", "-2", "1", "1"), {"residual": 1e-14}),
    ("summa_driver", ("768", "512", "640", "1", "0", "2", "2", "1"), {"gemm": 1e-13, "syrk": 1e-13, "trmm": 1e-13}),
    ("cacqr_driver", ("2", "16384", "128", "1", "1", "1", "1", "0", "0", "0", "0", "1", "1"), {"residual": 1e-13, "orthogonality": 1e-14}),
    ("cacqr_driver", ("1", "4096", "256", "1", "1", "1", "1", "0", "0", "0", "0", "1", "1"), {"residual": 1e-13, "orthogonality": 1e-12}),
])
def test_the_plain_c_drivers_run_on_the_cpu(tmp_path, name, argv, checks):
    """examples/*.c - the C forms of the reference's three bench drivers, no Python in the process - linked against the stand-in in
    compute mode instead of libamdhip64: they run their own validation blocks (test/cholesky/validate.hpp:33-46, the three
    summa::invoke overloads against the local operators, test/qr/validate.hpp:24-51) on a machine without a GPU.  The -m gpu suite
    runs the same programs with the same arguments on the device."""
    import re
    import shutil
    if not shutil.which("gcc") or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("gcc / ROCm headers not available")
    from capital_amd import build
    build.build(verbose=False)
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipshim"))
    import build_shim
    lib, shim = build_shim.build()
    exe = str(tmp_path / (name + ".cpu"))
    cmd = ["gcc", "-std=c99", "-D_POSIX_C_SOURCE=199309L", "-Wall", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", name + ".c"), "-L" + os.path.dirname(lib), "-lcapital_amd_shim", "-lhipshim", "-lm", "-Wl,-rpath," + os.path.dirname(lib), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ); env.pop("LD_PRELOAD", None); env["SHIM_COMPUTE"] = "1"
    run = subprocess.run([exe] + list(argv), capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    for word, bound in checks.items():
        m = re.search(r"\b%s\b[^=\n]*?=?\s*([0-9.]+e[+-][0-9]+)" % word, run.stdout)
        assert m, (word, run.stdout[-1500:])
        assert float(m.group(1)) <= bound, (word, m.group(1), run.stdout[-1500:])
