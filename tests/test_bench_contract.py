"""CPU-only checks of bench.py's plumbing: argument defaults, the cpu_baseline leg (the REAL reference binary when
oracle/_ref exists, else the NumPy port) and the JSON field contract of the pieces that do not need a GPU."""
import importlib.util
import os
import sys

import pytest

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


def test_gpus_n_without_a_launcher_never_leaves_without_a_json_line():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts its own ranks (bench.self_launch); on a box with fewer GPUs than
    ranks it must still print ONE JSON line (value null + the reason) and exit non-zero - never a bare usage message."""
    import json
    import subprocess
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return                                           # (a multi-GPU box would really run it: covered by the -m gpu tests)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CAPITAL_BENCH_EMULATE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       timeout=300, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode != 0 and len(lines) == 1, r.stdout + r.stderr
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and "GPU" in d["error"]


def test_self_launch_hands_the_arguments_on(monkeypatch):
    """The relaunch is `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    <same arguments>` with --n rewritten to --size (torchrun's parser trips over --n)."""
    b = _bench()
    seen = {}
    monkeypatch.setenv("CAPITAL_BENCH_EMULATE", "1")
    monkeypatch.setattr(b.subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)

    class A:
        gpus, steps, warmup, workload = 4, 1, 1, "cholesky"
    assert b.self_launch(A, ["--gpus", "4", "--n", "4096", "--workload", "cacqr", "--n=512"]) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--size", "4096", "--workload", "cacqr", "--size=512"]
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_the_traffic_figure_follows_the_kernels_machine_code():
    """roofline.traffic comes from committed PMC passes and is only reported for the kernel build it was measured on.  The identity is
    the MACHINE CODE of gemm.hip's kernels (capital_amd/build.py device_text_md5), not the source text: host-only edits of gemm.hip twice
    turned the figure into null without changing one instruction (round 5: a scratch-buffer release, the schedule checker's access notes)."""
    import types
    from capital_amd import build
    build.build(verbose=False)
    text = build.device_text_md5("gemm.hip")
    if text is None:
        pytest.skip("LLVM object tools not available")
    import bench
    args = types.SimpleNamespace(complete_inv=-1, nb=0, outer=0, tail=-1)
    t = bench.traffic_from_profile(65536, args)
    assert t is not None and 1e10 < t < 4e10, t                                    # the committed passes belong to THIS build's kernels
    assert bench.traffic_from_profile(32768, args) is None                         # ... and to this configuration only
    assert bench.traffic_from_profile(65536, types.SimpleNamespace(complete_inv=-1, nb=256, outer=0, tail=-1)) is None


def test_the_reference_on_the_operators_extra_is_bounded_and_cannot_raise():
    """bench.reference_on_operators (cpu_baseline.reference_with_offloaded_blas: the unmodified reference on libcapital_amd_cblas.so, PCIe-inclusive): on the CPU stand-in
    it returns the reference's own residual and the staging counters; past its limit or without a device it returns an error entry"""
    import bench
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "cholinv_cap")):
        e = bench.reference_on_operators(512, 1, 30)
        assert e["value"] is None and "not built" in e["error"]
        return
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipshim"))
    import build_shim
    build_shim.build_cblas()
    dirs = [os.path.join(build_shim.OUT, "cblas"), build_shim.OUT]
    e = bench.reference_on_operators(1024, 2, 120, dirs, {"SHIM_COMPUTE": "1"})
    assert e["value"] > 0 and e["residual"] < 1e-14 and e["blas_lapack_calls_served"] > 10 and e["bytes_host_to_device"] > 8 * 1024 * 1024, e
    late = bench.reference_on_operators(2048, 2, 0.05, dirs, {"SHIM_COMPUTE": "1"})
    assert late["value"] is None and "no result within" in late["error"]
    nodev = bench.reference_on_operators(256, 1, 60)              # the product build of the library: no device here -> a loud, contained failure
    import torch
    if not torch.cuda.is_available():
        assert nodev["value"] is None and "exit code" in nodev["error"]
