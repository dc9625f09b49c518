"""CPU-only checks of bench.py's plumbing: argument defaults, the cpu_baseline leg (the REAL reference binary when
oracle/_ref exists, else the NumPy port) and the JSON field contract of the pieces that do not need a GPU."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_defaults_match_the_baseline_metric():
    b = _bench()
    old = sys.argv
    sys.argv = ["bench.py"]
    try:
        a = b.parse()
    finally:
        sys.argv = old
    assert (a.gpus, a.n, a.complete_inv) == (1, 65536, -1)      # BASELINE.json: fp64 Cholesky, N = 65536
    assert a.steps >= 1 and a.warmup >= 1
    assert b.FP64_MFMA_PEAK_TF == 78.6


def test_cpu_baseline_leg_returns_a_reported_comparator():
    b = _bench()
    r = b.cpu_baseline(1024, budget_s=40)                          # tiny bounded sample: seconds on any host
    assert set(["value", "unit", "cores", "kind", "sample"]) <= set(r)
    assert r["unit"] == "TFLOP/s" and r["value"] > 0 and r["kind"] in ("reference", "port")
    assert r["cores"] >= 1
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "cholinv_ref")) and os.path.exists("/opt/conda/bin/mpiexec"):
        assert r["kind"] == "reference" and 1 <= r["cores"] <= (os.cpu_count() or 1)    # largest cube of ranks the host holds / all-cores MKL
        assert all(x["residual"] < 1e-14 for x in r["runs"]) and len(r["runs"]) >= 1
