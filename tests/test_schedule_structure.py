"""The C++ host side of the library - plan creation, index arithmetic, the multi-stream schedules with their event edges - run on
the CPU: the product's own object files linked against a recording stand-in for the HIP runtime and RCCL (tests/hipshim/), driven
through the C ABI (tests/hipshim/run_scenarios.py, its own process: no torch, no real HIP runtime), every trace replayed with
vector clocks (tests/hipshim/trace_check.py).

Kernels are names in a trace and nothing is computed - but every launch carries ACCESS NOTES: the library declares, next to each of
its launches, the windows the kernel reads and writes (cap_acc_* in capital_amd/csrc/common.h; a null hook in the product, the
device code is bit-identical with and without them), the stand-in adds its own for copies, memsets and collectives, and the replay
looks for two operations that touch the same bytes, one of them writing, without being ordered (stream order, event edge, host
synchronisation, NULL-stream rules).  The traces of ALL ranks of a multi-rank configuration are also replayed together: collectives
are matched across the ranks (kind, count, root, position in their communicator's sequence), and the peer copies of the IPC exchanges
are checked against the peer's own kernels.
What it shows, for every rank of every simulated grid and for the sizes of BASELINE.json (N = 65536 on 8 ranks included):
  * no two operations race on a buffer - look-ahead rings, strip buffers, message rings, gathered-strip buffers, the inverse tree's
    scratch slots, the IPC pushes into a peer's buffers behind their token all-reduces (removing ONE event edge from a recorded
    schedule is noticed for 45 - 95 % of the edges, the rest are edges another path implies - the test below looks at them);
  * every declared kernel footprint stays inside one allocation (2.4e5 launches, 1.5e6 windows);
  * every rank of a communicator enqueues the same collectives in the same order with the same counts and roots;
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


def _start(tmp, streams, only="", keep=""):
    if not os.path.exists(SO) or not os.path.isdir(os.path.join(ROOT, "capital_amd", "lib", "obj")):
        from capital_amd import build
        build.build(verbose=False)
    out = os.path.join(str(tmp), "scenarios_%s.json" % streams.replace(",", "_"))
    env = dict(os.environ); env.pop("LD_PRELOAD", None)
    env["SHIM_FILTER"] = only; env["SHIM_KEEP_TRACE"] = keep
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "hipshim", "run_scenarios.py"), out, streams], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, env=env)
    return p, out


def _finish(p, out):
    try:
        text, _ = p.communicate(timeout=900)
    except subprocess.TimeoutExpired:
        p.kill()
        raise
    assert p.returncode == 0, text[-3000:]
    return json.load(open(out))


def _run(tmp, streams, only="", keep=""):
    return _finish(*_start(tmp, streams, only, keep))


@pytest.fixture(scope="module")
def per_mode(tmp_path_factory):
    """every scenario with the caller on the NULL stream (torch's default) and on a non-blocking stream of its own: two processes, side by side"""
    tmp = tmp_path_factory.mktemp("shim")
    from capital_amd import build
    build.build(verbose=False)
    started = {m: _start(tmp, m) for m in ("0", "1")}
    return {m: _finish(*started[m]) for m in ("0", "1")}


@pytest.fixture(scope="module")
def scenarios(per_mode):
    return {"results": per_mode["0"]["results"] + per_mode["1"]["results"]}


def test_every_schedule_joins_its_streams_and_stays_inside_its_buffers(scenarios):
    res = scenarios["results"]
    assert len(res) >= 1100, len(res)
    bad = [(x["name"], x["findings"][:4]) for x in res if x["findings"]]
    assert not bad, "\n".join("%s: %s" % b for b in bad[:20])
    tot, unannotated = {}, {}
    for x in res:
        for k, v in x["stats"].items():
            if not isinstance(v, dict):
                tot[k] = tot.get(k, 0) + v
        unannotated.update(x["stats"].get("unannotated", {}))
    # the scenarios really ran the schedules: tens of thousands of launches, event edges and collectives went through the stand-in
    assert tot["kernels"] > 50000 and tot["waits"] > 50000 and tot["records"] > 50000 and tot["ops"] > 10000 and tot["oob"] == 0, tot
    # ... every launch came with its access notes, and millions of pairs of unordered operations were compared window by window
    assert not unannotated, unannotated
    assert tot["accesses"] > 2000000 and tot["race_checks"] > 10000000 and tot["races"] == 0, tot
    # ... the ranks of every multi-rank configuration were replayed together, collective by collective
    joint = [x for x in res if "joint replay" in x["name"]]
    assert len(joint) >= 120 and tot["collectives"] > 60000, (len(joint), tot)
    names = " ".join(x["name"] for x in res)
    for must in ("dist n=65536 nb=512 P=8  ci=-1 [joint replay of 8 ranks", "dist2d n=65536 nb=512 2x4  [joint replay of 8 ranks",
                 "dist n=8192 nb=512 P=8 {'ipc': 1} [joint replay of 8 ranks", "dist2d n=4096 nb=128 2x4 {'ipc': 1} [joint replay of 8 ranks",
                 "dmp n=8192 nb=512 P=8 [joint replay", "summa size=27 c=3 270x270x270 chunks=0 [joint replay of 27 ranks", "cholinv n=65536", "dist n=65536 nb=512 P=8 rank=7", "dist2d n=65536 nb=512 2x4 at (1,3)", "mpchol n=65536", "dmp n=8192 nb=512 P=8",
                 "cacqr m=2097152 n=256 iter=2 P=8", "cyclic_c=2", "{'ipc': 1}", "summa size=27 c=3 rank=26", "cacqr grid size=16 c=2 rank=15",
                 "redist n=1000 nb=128 size=8 c=2 Pr=2 rank=7", "desc n=300 nb=128 2x4 at (1,3)", "desc n=6144 nb=512 1x1",
                 "dist2d n=4096 nb=128 4x4 {'ipc': 1} [joint replay of 16 ranks", "dist2d n=1152 nb=128 4x8  [joint replay of 32 ranks", "operators m=1000 n=777 k=515", "plan life cycles"):
        assert must in names, must


def test_plans_give_back_what_they_allocate(per_mode):
    """Caller on the NULL stream: after 450+ plans / bundles / descriptors were created, used and destroyed the process holds what is
    per PROCESS by design - the chain's fall-back counters, the counter words + backup of the NULL stream and of the panel stream
    cap_dpotrf keeps per device, the NULL stream's split-K scratch - and nothing per plan."""
    d = per_mode["0"]
    assert len(d["results"]) >= 450 and not any(x["findings"] for x in d["results"])
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


def test_the_race_check_catches_seeded_defects():
    """Two kernels on different streams, window notes on one allocation: unordered overlapping writes are a race; an event edge, a host
    synchronisation in between, disjoint columns, disjoint triangles or two atomic accumulations are not."""
    def trace(a1, a2, glue=""):
        return _lines("""
            STREAM 1 nonblocking flags
            STREAM 2 nonblocking flags
            K 1 producer 1 1 1 0 1
            %s
            %s
            K 2 consumer 1 1 1 0 1
            %s
        """ % (a1, glue, a2))
    full = "A %d 7 0 1024 512 8 0 8"        # 8 columns of 64 doubles, pitch 128 doubles
    f, st = trace_check.check(trace(full % 2, full % 1))
    assert len(f) == 1 and f[0].startswith("race:") and "producer" in f[0] and "consumer" in f[0] and st["races"] == 1, f
    assert trace_check.check(trace(full % 1, full % 1))[0] == []                                   # two readers
    assert trace_check.check(trace(full % 4, full % 4))[0] == []                                   # two atomic accumulations
    assert len(trace_check.check(trace(full % 4, full % 2))[0]) == 1                               # atomic against a plain write
    assert trace_check.check(trace(full % 2, full % 1, "RECORD 1 5\nWAIT 2 5"))[0] == []           # ordered by an event edge
    assert trace_check.check(trace(full % 2, full % 1, "HOSTSYNC stream 1"))[0] == []              # the host waited before it enqueued the reader
    assert len(trace_check.check(trace(full % 2, full % 1, "RECORD 2 5\nWAIT 1 5"))[0]) == 1       # an edge in the wrong direction orders nothing
    # windows: the same rows of OTHER columns, other rows of the same columns, another allocation
    assert trace_check.check(trace("A 2 7 0 1024 512 4 0 8", "A 1 7 4096 1024 512 4 0 8"))[0] == []
    assert trace_check.check(trace("A 2 7 0 1024 256 8 0 8", "A 1 7 256 1024 256 8 0 8"))[0] == []
    assert trace_check.check(trace(full % 2, "A 1 8 0 1024 512 8 0 8"))[0] == []
    assert len(trace_check.check(trace("A 2 7 0 1024 264 8 0 8", "A 1 7 256 1024 256 8 0 8"))[0]) == 1      # one row too many
    # triangles of a 64 x 64 window (pitch 64 doubles): upper against strictly lower is disjoint, upper against lower shares the diagonal
    up, lo = "A 2 7 0 512 512 64 1 8", "A 1 7 0 512 512 64 2 8"
    assert len(trace_check.check(trace(up, lo))[0]) == 1
    assert trace_check.check(trace(up, "A 1 7 8 512 504 63 2 8"))[0] == []                         # rows 1.., columns 0..62: strictly below the diagonal
    # a note that no launch followed, a window that leaves its allocation: reported by the stand-in, passed on by the check
    f, _ = trace_check.check(_lines("ORPHAN 2 access notes in front of hipEventRecord\nOOB access of k: 4 columns of 64 bytes, pitch 64, leave the allocation of 128 bytes by 128"))
    assert len(f) == 2


def test_the_joint_replay_catches_seeded_defects():
    """Two ranks, a peer copy into a mapping of the other rank's buffer: only the all-reduce in front of it orders it behind the owner's
    reader; collectives that differ between the ranks or can never meet are findings of their own."""
    owner = """
        STREAM 1 nonblocking flags
        IPCGET 0 3 0
        K 1 reader 1 1 1 0 1
        A 1 3 0 0 4096 1 0 1
        %s
    """
    pusher = """
        STREAM 1 nonblocking flags
        ALIAS 9 0 0 0
        %s
        COPY 1 0x1 0x2 4096
        A 2 9 0 0 4096 1 0 1
    """
    ar = "OP 1 allreduce world 2 %d 1 -1"
    f, st = trace_check.check_joint([_lines(owner % (ar % 0)), _lines(pusher % (ar % 1))])
    assert f == [] and st["collectives"] == 2, f
    f, _ = trace_check.check_joint([_lines(owner % ""), _lines(pusher % "")])                       # no barrier: the push races with the reader
    assert len(f) == 1 and f[0].startswith("race:") and "rank 1 COPY" in f[0] and "rank 0 K reader" in f[0], f
    # a broadcast only orders the receivers behind the ROOT: with the pusher as root the owner's reader is not ordered in front of the push
    bc = "OP 1 bcast world 2 %d 1 1"
    f, _ = trace_check.check_joint([_lines(owner % (bc % 0)), _lines(pusher % (bc % 1))])
    assert len(f) == 1 and f[0].startswith("race:"), f
    bc0 = "OP 1 bcast world 2 %d 1 0"                                                                # ... with the owner as root it is
    assert trace_check.check_joint([_lines(owner % (bc0 % 0)), _lines(pusher % (bc0 % 1))])[0] == []
    # the same position of a communicator's sequence, different collectives
    f, _ = trace_check.check_joint([_lines(owner % (ar % 0)), _lines(pusher % "OP 1 allreduce world 2 1 2 -1")])
    assert any("collective mismatch" in x for x in f), f
    # one rank never enters
    f, _ = trace_check.check_joint([_lines(owner % (ar % 0)), _lines(pusher % "")])
    assert any("can never meet" in x for x in f), f


def test_removing_an_event_edge_from_a_recorded_schedule_is_noticed(tmp_path):
    """How sharp are the access notes?  Every hipStreamWaitEvent of three recorded schedules is removed in turn (tests/hipshim/mutate.py):
    the removals must show up as a race or as dangling work unless another path implies the edge (the fork of a helper stream that
    waits for a later event anyway, an event that two buffer rings wait for, the fixed-order edges between atomic accumulations -
    on the Pr x Pc plan half of the edges are of that kind: removing BOTH waits on such an event is noticed again)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipshim"))
    import mutate
    keep = str(tmp_path / "traces")
    floor = {"cholinv n=4096 ci=-1": 0.85, "dist n=4096 nb=128 P=4 rank=1  ci=-1": 0.5, "dist2d n=4096 nb=128 2x2 at (0,1) ": 0.4}
    for only, frac in floor.items():
        d = _run(tmp_path, "0", only=only, keep=keep)
        assert d["results"] and not any(x["findings"] for x in d["results"])
    import glob
    files = sorted(glob.glob(os.path.join(keep, "*.txt")))
    assert len(files) >= 3
    seen = 0
    for f in files:
        key = [k for k in floor if "".join(ch if ch.isalnum() else "_" for ch in k) in os.path.basename(f)]
        if not key or "safe" in f:
            continue
        base, nw, silent, kinds = mutate.mutate(open(f).read().splitlines(), limit=100)
        assert not base and nw >= 60
        assert (nw - len(silent)) >= floor[key[0]] * nw, (f, nw, len(silent), kinds)
        assert kinds["race"] >= 0.9 * (nw - len(silent)), (f, kinds)          # ... and nearly always as a race between two named kernels
        seen += 1
    assert seen == 3


def test_random_configurations_through_the_replay():
    """tests/hipshim/fuzz_scenarios.py: random sizes (to N = 18000), block widths, grids and option mixes of five plan kinds through the
    structural replay, the race check and the joint replay of all ranks (7000 scenarios passed when it was written; 60 configurations
    with a fixed seed here)"""
    env = dict(os.environ); env.pop("LD_PRELOAD", None); env["SHIM_FILTER"] = ""; env["SHIM_KEEP_TRACE"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipshim", "fuzz_scenarios.py"), "9", "60"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    last = [l for l in r.stdout.strip().splitlines() if "scenarios (incl. joint replays)" in l][-1]
    assert last.endswith(" 0 with findings") and int(last.split()[0]) >= 150, r.stdout[-3000:]
