"""CPU model of the meeting protocol of the one-launch diagonal-block chain (csrc/leaf.hip chain64_coop_kernel).

G workgroups run nblk steps; per step a mid-step meeting (every workgroup arrives, only the workers wait: workgroup 0 goes on with its
own block) and an end-of-step meeting (everybody arrives and waits).  A meeting is "add one to a word, poll until it reaches the
number of arrivals expected so far".  The kernel keeps ONE word per kind of meeting; the first version shared a word between the two,
and workgroup 0's early arrival at the end-of-step meeting could stand in for a worker that had not stored its solved block yet -
invisible at normal timing, wrong factors when eight processes time-sliced one GPU (round 4).  The model replays both protocols under
a random scheduler and checks the one property the kernel needs: nobody reads step i's solved blocks before every workgroup has
stored its own."""
import random

import pytest


def _run(G, steps, shared_word, seed, stall=None):
    """returns the first violation (step, reader, missing writer) or None.  stall = (workgroup, ticks): that workgroup is not
    scheduled for `ticks` picks right before it stores its solved block of step 1 (a preempted / late wave)."""
    rng = random.Random(seed)
    ctr = {"mid": 0, "end": 0}
    word = (lambda kind: "end") if shared_word else (lambda kind: kind)
    stored = [set() for _ in range(steps)]             # workgroups whose phase-S block of step i is in memory
    # per workgroup: program counter over the action list of all steps
    prog = []
    for w in range(G):
        acts = []
        for i in range(steps):
            acts += [("store", i), ("arrive", "mid", i)]
            if w > 0:
                acts += [("wait", "mid", i), ("read", i)]
            acts += [("arrive", "end", i), ("wait", "end", i)]
        prog.append(acts)
    pc = [0] * G
    expect = {"mid": [0] * G, "end": [0] * G}          # arrivals a workgroup expects at its next wait on that kind
    stalled = 0
    while any(pc[w] < len(prog[w]) for w in range(G)):
        w = rng.randrange(G)
        if pc[w] >= len(prog[w]):
            continue
        a = prog[w][pc[w]]
        if stall and w == stall[0] and a == ("store", 1) and stalled < stall[1]:
            stalled += 1
            continue
        if a[0] == "store":
            stored[a[1]].add(w)
        elif a[0] == "arrive":
            ctr[word(a[1])] += 1
            if shared_word:
                expect["end"][w] += G               # one running total for both kinds
            else:
                expect[a[1]][w] += G
        elif a[0] == "wait":
            k = word(a[1])
            if ctr[k] < expect[k][w]:
                continue                               # keep polling
        elif a[0] == "read":
            missing = set(range(G)) - stored[a[1]]
            if missing:
                return (a[1], w, sorted(missing)[0])
        pc[w] += 1
    return None


@pytest.mark.parametrize("G", [2, 3, 8, 32])
def test_one_word_per_meeting_kind_is_safe_under_any_schedule(G):
    for seed in range(60):
        assert _run(G, 6, False, seed) is None
        assert _run(G, 6, False, seed, stall=(G - 1, 4000)) is None


def test_the_shared_word_of_the_first_version_is_caught_by_the_model():
    """the model is sensitive: with one word for both meetings a late worker's block is read before it is stored"""
    hits = [_run(4, 6, True, seed, stall=(3, 4000)) for seed in range(40)]
    assert any(h is not None for h in hits)
    step, reader, missing = next(h for h in hits if h is not None)
    assert reader != 0 and missing == 3
