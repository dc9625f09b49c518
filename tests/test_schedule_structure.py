"""The C++ host side of the library - plan creation, index arithmetic, the multi-stream schedules with their event edges - run on
the CPU: the product's own object files linked against a recording stand-in for the HIP runtime and RCCL (tests/hipshim/), driven
through the C ABI (tests/hipshim/run_scenarios.py, its own process: no torch, no real HIP runtime), every trace replayed with
vector clocks (tests/hipshim/trace_check.py).

What this can and cannot show: kernels are names in a trace, so nothing is computed and a kernel's reads and writes are unknown -
data races between kernels are the business of the -m gpu tests (results against the oracle under random per-stream delays).
What it does show, for every rank of every simulated grid and for the sizes of BASELINE.json (N = 65536 on 8 ranks included):
  * every stream a call puts work on is joined into the caller's stream (or waited for by the host) before the call returns -
    with the caller on the NULL stream (torch's default) and on a non-blocking stream of its own;
  * no wait names an event that was never recorded, no launch has an empty grid, no stream is used after its destruction;
  * every copy / memset stays inside one allocation;
  * a plan gives back everything it allocated (round 5: this found a 4.25 MiB chain backup and an 8 MiB split-K scratch buffer
    kept per helper-stream handle for the life of the process - 282 live allocations after 286 plans, now 4)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "hipshim"))
import trace_check  # noqa: E402

SO = os.path.join(ROOT, "capital_amd", "lib", "libcapital_amd.so")


def _run(tmp, streams):
    if not os.path.exists(SO) or not os.path.isdir(os.path.join(ROOT, "capital_amd", "lib", "obj")):
        from capital_amd import build
        build.build(verbose=False)
    out = os.path.join(str(tmp), "scenarios_%s.json" % streams.replace(",", "_"))
    env = dict(os.environ); env.pop("LD_PRELOAD", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipshim", "run_scenarios.py"), out, streams], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return json.load(open(out))


@pytest.fixture(scope="module")
def scenarios(tmp_path_factory):
    return _run(tmp_path_factory.mktemp("shim"), "0,1")


def test_every_schedule_joins_its_streams_and_stays_inside_its_buffers(scenarios):
    res = scenarios["results"]
    assert len(res) >= 580, len(res)
    bad = [(x["name"], x["findings"][:4]) for x in res if x["findings"]]
    assert not bad, "\n".join("%s: %s" % b for b in bad[:20])
    tot = {}
    for x in res:
        for k, v in x["stats"].items():
            if not isinstance(v, dict):
                tot[k] = tot.get(k, 0) + v
    # the scenarios really ran the schedules: tens of thousands of launches, event edges and collectives went through the stand-in
    assert tot["kernels"] > 50000 and tot["waits"] > 50000 and tot["records"] > 50000 and tot["ops"] > 10000 and tot["oob"] == 0, tot
    names = " ".join(x["name"] for x in res)
    for must in ("cholinv n=65536", "dist n=65536 nb=512 P=8 rank=7", "dist2d n=65536 nb=512 2x4 at (1,3)", "mpchol n=65536", "dmp n=8192 nb=512 P=8",
                 "cacqr m=2097152 n=256 iter=2 P=8", "cyclic_c=2", "{'ipc': 1}", "summa size=27 c=3 rank=26", "cacqr grid size=16 c=2 rank=15",
                 "redist n=1000 nb=128 size=8 c=2 Pr=2 rank=7", "desc n=300 nb=128 2x4 at (1,3)", "operators m=1000 n=777 k=515", "plan life cycles"):
        assert must in names, must


def test_plans_give_back_what_they_allocate(tmp_path):
    """Caller on the NULL stream: after 290+ plans / bundles / descriptors were created, used and destroyed the process holds what is
    per PROCESS by design - the chain's fall-back counters, the counter words + backup of the NULL stream and of the panel stream
    cap_dpotrf keeps per device, the NULL stream's split-K scratch - and nothing per plan."""
    d = _run(tmp_path, "0")
    assert len(d["results"]) >= 290 and not any(x["findings"] for x in d["results"])
    live = d["live_allocations_at_exit"]
    assert len(live) <= 6, live


def _lines(text):
    return [l.strip() for l in text.strip().splitlines()]


def test_the_checker_catches_seeded_defects():
    sound = """
        STREAM 1 nonblocking flags
        STREAM 2 nonblocking priority
        MARK begin f
        K 1 import 1 1 1 0
        RECORD 1 1
        WAIT 2 1
        K 2 panel 1 1 1 0
        RECORD 2 2
        WAIT 1 2
        K 1 tail 1 1 1 0
        MARK end f user=1
    """
    assert trace_check.check(_lines(sound))[0] == []
    # the join is missing: the caller's stream never waits for the helper stream's last kernel
    f, _ = trace_check.check(_lines(sound.replace("WAIT 1 2\n", "")))
    assert len(f) == 1 and f[0].startswith("dangling: stream 2"), f
    # ... unless the host waits for it
    assert trace_check.check(_lines(sound.replace("WAIT 1 2\n", "HOSTSYNC stream 2\n")))[0] == []
    assert trace_check.check(_lines(sound.replace("WAIT 1 2\n", "HOSTSYNC device\n")))[0] == []
    # work enqueued behind the recorded event is not covered by waiting for the event
    f, _ = trace_check.check(_lines(sound.replace("WAIT 1 2\n", "K 2 late 1 1 1 0\nWAIT 1 2\n")))
    assert len(f) == 1 and "dangling" in f[0], f
    # a wait for an event nobody recorded orders nothing
    f, _ = trace_check.check(_lines(sound.replace("RECORD 2 2\n", "")))
    assert any("unrecorded wait" in x for x in f) and any("dangling" in x for x in f), f
    # NULL-stream semantics: a blocking stream is joined by the next NULL-stream operation, a non-blocking one is not
    null = """
        STREAM 3 %s flags
        MARK begin g
        K 3 helper 1 1 1 0
        MARK end g user=0
    """
    assert trace_check.check(_lines(null % "blocking"))[0] == []
    assert len(trace_check.check(_lines(null % "nonblocking"))[0]) == 1
    # a destroyed stream, an out-of-range copy and a refused launch are findings as they stand
    f, _ = trace_check.check(_lines("STREAM 4 nonblocking flags\nSTREAMDESTROY 4\nMARK begin h\nK 4 x 1 1 1 0\nHOSTSYNC device\nMARK end h user=0\n"
                                    "OOB copy dst 0x10 64\nBADLAUNCH 0 k grid 0 1 1 block 256 1 1"))
    assert sum("destroyed stream" in x for x in f) == 1 and sum(x.startswith("OOB") for x in f) == 1 and sum(x.startswith("BADLAUNCH") for x in f) == 1, f


@pytest.mark.parametrize("n,csv", [(65536, "r05_bench_n65536_kernel_stats.csv"), (32768, "r05_bench_n32768_kernel_stats.csv")])
def test_the_recorded_schedule_is_the_one_rocprof_saw_on_the_gpu(scenarios, n, csv):
    """The stand-in's trace of one factor call at bench.py's knobs has, kernel by kernel, the launch counts of the rocprofv3
    --kernel-trace --stats summary of `python bench.py --size N` on the MI355X (profiles/, committed): what runs under the stand-in IS
    the schedule that was measured."""
    import csv as csvmod
    import shutil
    if not shutil.which("c++filt"):
        pytest.skip("c++filt not available")
    path = os.path.join(ROOT, "profiles", csv)
    rows = {r["Name"]: int(r["Calls"]) for r in csvmod.DictReader(open(path))}
    x = [r for r in scenarios["results"] if r["name"].startswith("cholinv n=%d ci=-1 split=1 bc=-5" % n) and "NULL" in r["name"]][0]
    hist = x["stats"]["factor_kernels"]
    names = sorted(hist)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
    mine = {d: hist[m] for m, d in zip(names, dem)}
    lib = {k: v for k, v in rows.items() if "(anonymous namespace)::" in k and not k.startswith("void at::")}
    # factor calls in the profiled run = calls of the dominant kernel / its launches per factor
    dom = [k for k in mine if "dgemm_tn_dma_kernel<1, false, 0, true, false>" in k][0]
    assert rows[dom] % mine[dom] == 0
    calls = rows[dom] // mine[dom]
    assert calls >= 2
    checked = 0
    for k, v in mine.items():
        if "copy_window_kernel" in k:
            continue                                    # (the bench's own residual check uses it too)
        assert k in lib, (k, sorted(lib)[:8])
        assert lib[k] == calls * v, (k, lib[k], calls, v)
        checked += 1
    assert checked >= 4
