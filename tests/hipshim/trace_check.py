"""TEST INFRASTRUCTURE: replay of a trace of the recording HIP stand-in (tests/hipshim/hipshim.cpp) with vector clocks.

Happens-before as the HIP stream model defines it: operations of one stream are ordered; hipEventRecord captures the clock of
its stream, hipStreamWaitEvent merges the captured clock into the waiting stream (a wait on an event that was never recorded is
a no-op - and reported); the legacy NULL stream synchronises with every BLOCKING stream in both directions and with no
non-blocking one; host synchronisations (stream / event / device) merge into the host's clock, and whatever the host enqueues
afterwards is ordered behind what it has waited for.

check(lines) -> (findings, stats); empty findings = the trace is sound:
  * "dangling": at the end of a marked call ("MARK end <name> user=<stream>") a stream that received work during the call has
    work that neither the caller's stream nor the host is ordered behind - the caller cannot know when it completes;
  * "unrecorded wait", "use of a destroyed stream", out-of-range copies / access notes ("OOB"), refused launches ("BADLAUNCH"),
    bad frees, access notes that no launch followed ("ORPHAN");
  * "race": two operations on different streams touch the same bytes (their "A" lines: the access notes the library declares next
    to every launch, capital_amd/csrc/common.h, and the stand-in's own for copies, memsets and collectives), at least one of them
    writing (or one atomic and the other not), and neither is ordered behind the other.

Access notes: "A mode alloc offset pitch row_bytes cols tri elem" = `cols` columns of `row_bytes` bytes, `pitch` bytes apart, starting
`offset` bytes into allocation number `alloc`; tri 1 / 2: only the elements (of `elem` bytes) with row <= col / row >= col.
mode 1 read, 2 write, 3 read-write, 4 device-scope atomic.

check_joint(traces) replays the traces of ALL ranks of one configuration together.  Collectives ("OP stream kind label size me count
root" lines of the tests' communicators) are matched by (label, position in that communicator's sequence): a mismatch of kind / count /
root between the ranks, or ranks that enqueue their collectives in orders that can never meet, are findings.  A collective orders its
completion on a rank behind the ENTRY of every rank whose data it needs (all-reduce / all-gather / all-to-all: everybody; broadcast:
the root) - that is what RCCL guarantees, and all it guarantees.  Mappings of a peer's buffer (hipIpcOpenMemHandle of a tagged handle,
"ALIAS" lines) resolve to the peer's own allocation, so the peer copies of the IPC exchanges are checked against the peer's kernels."""
import numpy as np


def _join(a, b):
    for k, v in b.items():
        if a.get(k, 0) < v:
            a[k] = v


class Access:
    __slots__ = ("mode", "alloc", "off", "pitch", "rb", "cols", "tri", "elem", "lo", "hi", "_iv", "single")

    def __init__(self, mode, alloc, off, pitch, rb, cols, tri, elem):
        self.mode, self.alloc, self.off, self.pitch, self.rb, self.cols, self.tri, self.elem = mode, alloc, off, pitch, rb, cols, tri, elem
        if cols > 1 and pitch < rb:          # overlapping columns (never produced by the library): one interval over everything
            self.rb = rb = (cols - 1) * max(pitch, 0) + rb
            self.cols = cols = 1
            self.tri = tri = 0
        self.lo = off
        self.hi = off + (cols - 1) * pitch + rb
        self.single = cols == 1 or (tri == 0 and pitch == rb)
        self._iv = None

    def intervals(self):
        if self._iv is None:
            j = np.arange(self.cols, dtype=np.int64)
            s = self.off + j * self.pitch
            e = s + self.rb
            if self.tri == 1:                # rows 0 .. min(j, rows - 1)
                e = s + np.minimum((j + 1) * self.elem, self.rb)
            elif self.tri == 2:              # rows j .. rows - 1; columns past the last row are empty
                s = s + np.minimum(j * self.elem, self.rb)
            keep = e > s
            self._iv = (s[keep], e[keep])
        return self._iv

    def overlaps(self, o):
        if self.hi <= o.lo or o.hi <= self.lo:
            return False
        if self.single and o.single:
            return True
        sa, ea = self.intervals()
        sb, eb = o.intervals()
        if sa.size == 0 or sb.size == 0:
            return False
        if sa.size < sb.size:
            sa, ea, sb, eb = sb, eb, sa, ea
        idx = np.searchsorted(sa, eb, side="left") - 1          # last interval of A that starts before B's interval ends
        ok = idx >= 0
        return bool(np.any(ok & (ea[np.maximum(idx, 0)] > sb)))

    def describe(self):
        return "%s alloc %d +%d: %d col(s) x %d B, pitch %d%s" % ({1: "R", 2: "W", 3: "RW", 4: "ATOMIC"}.get(self.mode, "?"), self.alloc, self.off, self.cols,
                                                                  self.rb, self.pitch, {0: "", 1: ", upper", 2: ", lower"}[self.tri])


def _conflict(a, b):
    if a == 1 and b == 1:
        return False
    if a == 4 and b == 4:
        return False
    return True




class _Shared:
    def __init__(self, nranks, races, max_race_reports):
        self.findings = []
        self.hist = {}               # (rank, allocation) -> stream key -> [(clock, Access, description)]
        self.reported = set()
        self.nraces = 0
        self.races = races
        self.max_reports = max_race_reports
        self.coll = {}               # (label, seq) -> {"size", "sig", "enter": {me: clock}, "root"}
        self.exports = [dict() for _ in range(nranks)]     # rank -> export number -> (allocation, offset)
        self.stats = {"kernels": 0, "copies": 0, "ops": 0, "records": 0, "waits": 0, "streams": 0, "calls": 0, "accesses": 0, "race_checks": 0,
                      "collectives": 0}
        self.unannotated = {}


def _replay(rank, lines, sh, allow_unrecorded, joint):
    """generator over one rank's trace; yields (label, seq) whenever it has to wait for its peers to enter a collective"""
    R = rank
    vc = {(R, 0): {}}            # stream key -> clock
    blocking = set()
    dead = set()
    ev = {}
    host = {}
    touched = None
    call = None
    alias = {}                   # allocation number of a mapping -> (peer, export number, offset)
    seqno = {}                   # label -> collectives entered so far
    findings, stats = sh.findings, sh.stats
    tag = ("rank %d: " % R) if joint else ""

    def tick(s):
        if s in dead:
            findings.append(tag + "use of a destroyed stream %d (in %s)" % (s[1], call))
        c = vc.setdefault(s, {})
        _join(c, host)           # enqueued now: behind everything the host has already waited for
        if s[1] == 0:
            for b in blocking:
                _join(c, vc.get(b, {}))
        elif s in blocking:
            _join(c, vc[(R, 0)])
        c[s] = c.get(s, 0) + 1
        if touched is not None:
            touched.add(s)

    def resolve(a):
        """the allocation an access really touches: a mapping of a peer's buffer is the peer's allocation"""
        if joint and a.alloc in alias:
            peer, exp, off = alias[a.alloc]
            tgt = sh.exports[peer].get(exp) if 0 <= peer < len(sh.exports) else None
            if tgt is None:
                return (R, a.alloc)
            a.off += tgt[1]; a.lo += tgt[1]; a.hi += tgt[1]; a._iv = None
            return (peer, tgt[0])
        return (R, a.alloc)

    def access(a, s, view, clk, desc):
        key = resolve(a)
        per = sh.hist.setdefault(key, {})
        if sh.races:
            for o, lst in per.items():
                if o == s:
                    continue
                seen = view.get(o, 0)
                for oclk, b, bdesc in reversed(lst):
                    if oclk <= seen:
                        break            # everything earlier on that stream is ordered in front of this operation
                    stats["race_checks"] += 1
                    if _conflict(a.mode, b.mode) and a.overlaps(b):
                        sh.nraces += 1
                        rk = (desc.split(" @")[0], bdesc.split(" @")[0])
                        if rk not in sh.reported and len(sh.reported) < sh.max_reports:
                            sh.reported.add(rk)
                            findings.append("race: %s%s [%s] is not ordered behind %s [%s] (it has seen %d of that stream's %d operations; in %s)" % (
                                tag, desc, a.describe(), bdesc, b.describe(), seen, oclk, call))
        per.setdefault(s, []).append((clk, a, desc))

    i, n = 0, len(lines)
    while i < n:
        raw = lines[i]; lineno = i; i += 1
        t = raw.split()
        if not t:
            continue
        k = t[0]
        if k == "A":
            stats["accesses"] += 1      # (an access line without an operation in front of it: ignored)
            continue
        if k == "HA":
            # the host thread reads / writes a window NOW: ordered behind what the host has waited for; everything it enqueues
            # afterwards is ordered behind this access (tick() joins the host's clock, which carries the host's own counter)
            hk = (R, -1)
            host[hk] = host.get(hk, 0) + 1
            stats["accesses"] += 1
            access(Access(*(int(x) for x in t[1:9])), hk, host, host[hk], "%shost access @line %d" % (("rank %d " % R) if joint else "", lineno + 1))
            continue
        if k in ("K", "COPY", "COPY2D", "SET", "OP"):
            s = (R, int(t[1]))
            accs = []
            while i < n and lines[i].startswith("A "):
                accs.append(Access(*(int(x) for x in lines[i].split()[1:9]))); i += 1
            stats["accesses"] += len(accs)
            stats["kernels" if k == "K" else "ops" if k == "OP" else "copies"] += 1
            name = t[2] if k in ("K", "OP") else k
            if k == "K" and len(t) >= 8 and t[7] == "0":
                sh.unannotated[name] = sh.unannotated.get(name, 0) + 1
            desc = "%s%s %s on stream %d @line %d" % (("rank %d " % R) if joint else "", k, name, s[1], lineno + 1)
            if k == "OP" and joint and len(t) >= 8:
                # a collective of a labelled communicator: OP stream kind label size me count root
                kind, label, size, me, count, root = t[2], t[3], int(t[4]), int(t[5]), int(t[6]), int(t[7])
                tick(s)
                enter = dict(vc[s])
                q = seqno.get(label, 0); seqno[label] = q + 1
                inst = sh.coll.setdefault((label, q), {"size": size, "sig": (kind, count, root), "enter": {}, "who": {}})
                if inst["sig"] != (kind, count, root) or inst["size"] != size:
                    findings.append("collective mismatch on %s #%d: rank %d enqueues %s count %d root %d, rank %d enqueued %s count %d root %d" % (
                        label, q, R, kind, count, root, next(iter(inst["who"].values()), -1), inst["sig"][0], inst["sig"][1], inst["sig"][2]))
                inst["enter"][me] = enter; inst["who"][me] = R
                stats["collectives"] += 1
                while len(inst["enter"]) < inst["size"]:
                    yield (label, q, desc)
                need = [inst["enter"][root]] if kind == "bcast" and me != root else ([] if kind == "bcast" else list(inst["enter"].values()))
                for c in need:
                    _join(vc[s], c)
                vc[s][s] = vc[s].get(s, 0) + 1          # the exit tick: the collective's own accesses end here
                for a in accs:
                    access(a, s, enter, vc[s][s], desc)
                continue
            tick(s)
            for a in accs:
                access(a, s, vc[s], vc[s][s], desc)
        elif k == "MARK":
            if t[1] == "begin":
                call = " ".join(t[2:]); touched = set(); stats["calls"] += 1
            elif t[1] == "end":
                user = (R, int(t[-1].split("=")[1]))
                ucl = vc.get(user, {})
                for s in sorted(touched or ()):
                    last = vc.get(s, {}).get(s, 0)
                    if user[1] == 0 and s in blocking:
                        continue                 # the next operation on the NULL stream waits for every blocking stream
                    if s != user and last > max(ucl.get(s, 0), host.get(s, 0)):
                        findings.append(tag + "dangling: stream %d has work (clock %d) that neither the caller's stream %d (sees %d) nor the host (%d) "
                                        "is ordered behind at the end of %s" % (s[1], last, user[1], ucl.get(s, 0), host.get(s, 0), call))
                touched = None; call = None
            # (other marks - "scenario ..." - only label the trace)
        elif k == "STREAM":
            s = (R, int(t[1])); vc[s] = {}; stats["streams"] += 1
            if t[2] == "blocking":
                blocking.add(s)
        elif k == "STREAMDESTROY":
            dead.add((R, int(t[1]))); blocking.discard((R, int(t[1])))
        elif k == "RECORD":
            s, e = (R, int(t[1])), int(t[2]); tick(s); ev[e] = dict(vc[s]); stats["records"] += 1
        elif k == "WAIT":
            s, e = (R, int(t[1])), int(t[2]); stats["waits"] += 1
            if e not in ev:
                if call not in allow_unrecorded:
                    findings.append(tag + "unrecorded wait: stream %d waits for event %d that was never recorded (in %s)" % (s[1], e, call))
                tick(s)
            else:
                tick(s); _join(vc[s], ev[e])
        elif k == "HOSTSYNC":
            if t[1] == "stream":
                s = (R, int(t[2]))
                c = dict(vc.get(s, {}))
                if s[1] == 0:
                    for b in blocking:
                        _join(c, vc.get(b, {}))
                _join(host, c)
            elif t[1] == "event":
                _join(host, ev.get(int(t[2]), {}))
            else:
                for c in vc.values():
                    _join(host, c)
        elif k in ("OOB", "BADLAUNCH", "BADFREE", "BADSTREAMDESTROY", "ORPHAN"):
            findings.append(tag + raw.strip() + " (in %s)" % call)
        elif k == "FREE":
            a = int(t[1])
            if a in alias:
                del alias[a]
            else:
                sh.hist.pop((R, a), None)
        elif k == "IPCGET":
            sh.exports[R][int(t[1])] = (int(t[2]), int(t[3]))
        elif k == "ALIAS":
            alias[int(t[1])] = (int(t[2]), int(t[3]), int(t[4]))
        elif k == "EVENTDESTROY":
            pass
        else:
            findings.append(tag + "unknown trace line: " + raw.strip())


def check_joint(traces, allow_unrecorded=(), races=True, max_race_reports=12):
    """traces: one list of lines per rank (rank = position).  -> (findings, stats)"""
    joint = len(traces) > 1
    sh = _Shared(len(traces), races, max_race_reports)
    gens = [_replay(r, lines, sh, allow_unrecorded, joint) for r, lines in enumerate(traces)]
    waiting = [None] * len(gens)
    alive = set(range(len(gens)))
    while alive:
        progress = False
        for r in sorted(alive):
            w = waiting[r]
            if w is not None:
                inst = sh.coll[(w[0], w[1])]
                if len(inst["enter"]) < inst["size"]:
                    continue
            try:
                nxt = next(gens[r])
                progress = progress or nxt != w
                waiting[r] = nxt
            except StopIteration:
                alive.discard(r); waiting[r] = None; progress = True
        if not progress:
            sh.findings.append("collectives that can never meet: " + "; ".join(
                "%s waits in %s #%d for %d of %d ranks" % (waiting[r][2].split(" @")[0], waiting[r][0], waiting[r][1],
                                                           sh.coll[(waiting[r][0], waiting[r][1])]["size"] - len(sh.coll[(waiting[r][0], waiting[r][1])]["enter"]),
                                                           sh.coll[(waiting[r][0], waiting[r][1])]["size"]) for r in sorted(alive) if waiting[r]))
            break
    if sh.nraces > len(sh.reported):
        sh.findings.append("race: %d conflicting pairs in all, the first %d distinct ones are listed" % (sh.nraces, len(sh.reported)))
    sh.stats["races"] = sh.nraces
    sh.stats["unannotated"] = sh.unannotated
    return sh.findings, sh.stats


def check(lines, allow_unrecorded=(), races=True, max_race_reports=12):
    """one rank's trace on its own (collectives are local operations with their access notes)"""
    return check_joint([lines], allow_unrecorded, races, max_race_reports)
