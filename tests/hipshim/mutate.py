"""TEST INFRASTRUCTURE: how sharp is the replay of trace_check.py on a given trace?  Every hipStreamWaitEvent of the trace is removed
in turn and the trace checked again: a removal nobody notices is either a redundant edge of the schedule or a blind spot of the
access notes.  python tests/hipshim/mutate.py trace.txt [...]  prints, per trace, waits / removals noticed / silent removals."""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import trace_check  # noqa: E402


def mutate(lines, limit=None):
    base, _ = trace_check.check(lines)
    waits = [i for i, l in enumerate(lines) if l.startswith("WAIT ")]
    if limit and len(waits) > limit:
        step = len(waits) / float(limit)
        waits = [waits[int(j * step)] for j in range(limit)]
    silent, kinds = [], {"race": 0, "dangling": 0, "other": 0}
    for i in waits:
        f, _ = trace_check.check(lines[:i] + lines[i + 1:], max_race_reports=1)
        f = [x for x in f if x not in base]
        if not f:
            silent.append(i)
        else:
            kinds["race" if any(x.startswith("race") for x in f) else "dangling" if any(x.startswith("dangling") for x in f) else "other"] += 1
    return base, len(waits), silent, kinds


if __name__ == "__main__":
    for path in sys.argv[1:]:
        lines = open(path).read().splitlines()
        base, nw, silent, kinds = mutate(lines, limit=int(os.environ.get("MUTATE_LIMIT", "0")) or None)
        print("%-90s waits %4d  noticed %4d (race %d, dangling %d)  silent %3d  base findings %d" % (os.path.basename(path)[:90], nw, nw - len(silent), kinds["race"],
                                                                                                  kinds["dangling"], len(silent), len(base)))
        if os.environ.get("MUTATE_SHOW"):
            for i in silent[:40]:
                ctx = [l for l in lines[max(0, i - 6):i] if not l.startswith("A ")][-2:]
                print("      line %d: %s   (after: %s)" % (i + 1, lines[i], " | ".join(c[:70] for c in ctx)))
