"""libcapital_amd_cblas.so on the device (see tests/test_reference_offload.py for the CPU form of the same checks): every entry point on
host memory against NumPy, and the REAL reference - oracle/_ref/*_cap, its unmodified sources linked with this library in MKL's place -
running cholinv / CholeskyQR2 / SUMMA on 1 ... 8 MPI ranks that share cuda:0: its own validators, its dumps against the MKL-linked build's.
Runs in child processes without torch (the library and the HIP runtime are the only GPU code in them).  Last file of the suite by name."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_reference_offload as tro  # noqa: E402

LIBDIR = os.path.join(ROOT, "capital_amd", "lib")

CHILD = r"""
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, %r)
import test_reference_offload as tro
L = C.CDLL(%r)                       # fails loudly when the library (or libcapital_amd.so / the HIP runtime behind it) is missing
worst = tro.exercise(L, np.random.default_rng(5), [(70, 40, 33), (257, 96, 300), (1, 1, 1), (5, 300, 2), (1024, 512, 768), (2048, 1024, 256)], every_form=False)
print("WORST " + json.dumps(worst))
"""


@pytest.mark.gpu
def test_every_entry_point_on_the_device_against_numpy():
    import json
    lib = os.path.join(LIBDIR, "libcapital_amd_cblas.so")
    assert os.path.exists(lib), "libcapital_amd_cblas.so is not built (python -m capital_amd.build)"
    env = dict(os.environ); env.pop("LD_PRELOAD", None)
    r = subprocess.run([sys.executable, "-c", CHILD % (os.path.join(ROOT, "tests"), lib)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    worst = json.loads([l for l in r.stdout.splitlines() if l.startswith("WORST ")][-1][6:])
    assert set(worst) == {"dgemm", "dsyrk", "dtrmm", "dpotrf", "dtrtri"}
    assert max(worst.values()) < 1e-12, worst


@pytest.mark.gpu
def test_a_plain_cblas_program_runs_on_the_device(tmp_path):
    tro.build_and_run_demo(tmp_path, LIBDIR, [LIBDIR, "/opt/rocm/lib"], {}, 3001)


@pytest.mark.gpu
@pytest.mark.skipif(not tro.reference_available(), reason="oracle/_ref/*_cap (the reference built from /root/reference on this library) or mpiexec is not here")
def test_the_real_reference_runs_on_the_device_and_passes_its_own_validators():
    probe = subprocess.run([tro.MPIEXEC, "-n", "8", os.path.join(tro.REFDIR, "cholinv_ref"), "16", "1", "1", "0", "0", "0", "1", "-", "1"],
                           capture_output=True, text=True, timeout=300, env=dict(os.environ, MKL_NUM_THREADS="1"))
    if probe.returncode != 0 and "ranks=" not in probe.stdout:
        pytest.skip("the MPI launcher does not start the MKL-linked reference on this box: " + (probe.stderr or probe.stdout)[-300:])
    env = tro.cap_env([LIBDIR, "/opt/rocm/lib"])
    # (a subset of the CPU test's runs: every launch is up to 8 processes that each open the device - about 5 s apiece)
    for exe, ranks, argv, checks in [tro.REFERENCE_RUNS[i] for i in (0, 2, 3, 6, 7)]:
        kv, served = tro.run_reference(env, exe + "_cap", ranks, list(argv) + ["-", 1], timeout=300)
        for k, tol in checks.items():
            assert kv[k] < tol, (exe, ranks, argv, kv)
        assert len(served) == ranks and min(served) > 0, (exe, ranks, served)
    tro.dumps_equal("cholinv", 8, (300, 0, 1, -2, 0, 0, 1), env, 1e-12)
    tro.dumps_equal("cacqr", 8, (2, 600, 48, 2, 1, 1, 0), env, 1e-12)
    for op, m, n, k in ((0, 150, 130, 170), (2, 140, 90, 0)):
        tro.dumps_equal("summa", 8, (op, m, n, k, 2, 0, 2, 1.5, -0.5 if op in (0, 5) else 0.0), env, 1e-13)
