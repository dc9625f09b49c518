"""TEST INFRASTRUCTURE: replay of a trace of the recording HIP stand-in (tests/hipshim/hipshim.cpp) with vector clocks.

Happens-before as the HIP stream model defines it: operations of one stream are ordered; hipEventRecord captures the clock of
its stream, hipStreamWaitEvent merges the captured clock into the waiting stream (a wait on an event that was never recorded is
a no-op - and reported); the legacy NULL stream synchronises with every BLOCKING stream in both directions and with no
non-blocking one; host synchronisations (stream / event / device) merge into the host's clock.

check(lines) -> list of findings (strings); empty = the trace is structurally sound:
  * "dangling": at the end of a marked call ("MARK end <name> user=<stream>") a stream that received work during the call has
    work that neither the caller's stream nor the host is ordered behind - the caller cannot know when it completes;
  * "unrecorded wait", "use of a destroyed stream", out-of-range copies ("OOB"), refused launches ("BADLAUNCH"), bad frees."""


def _join(a, b):
    for k, v in b.items():
        if a.get(k, 0) < v:
            a[k] = v


def check(lines, allow_unrecorded=()):
    vc = {0: {}}                 # stream -> clock
    blocking = set()
    dead = set()
    ev = {}                      # event -> captured clock
    host = {}
    findings = []
    touched = None               # streams with work since the last "MARK begin"
    call = None
    stats = {"kernels": 0, "copies": 0, "ops": 0, "records": 0, "waits": 0, "streams": 0, "calls": 0}

    def tick(s):
        if s in dead:
            findings.append("use of a destroyed stream %d (in %s)" % (s, call))
        c = vc.setdefault(s, {})
        if s == 0:
            for b in blocking:
                _join(c, vc.get(b, {}))
        elif s in blocking:
            _join(c, vc[0])
        c[s] = c.get(s, 0) + 1
        if touched is not None:
            touched.add(s)

    for raw in lines:
        t = raw.split()
        if not t:
            continue
        k = t[0]
        if k == "MARK":
            if t[1] == "begin":
                call = " ".join(t[2:]); touched = set(); stats["calls"] += 1
            elif t[1] == "end":
                user = int(t[-1].split("=")[1])
                ucl = vc.get(user, {})
                for s in sorted(touched or ()):
                    last = vc.get(s, {}).get(s, 0)
                    if user == 0 and s in blocking:
                        continue                 # the next operation on the NULL stream waits for every blocking stream
                    if s != user and last > max(ucl.get(s, 0), host.get(s, 0)):
                        findings.append("dangling: stream %d has work (clock %d) that neither the caller's stream %d (sees %d) nor the host (%d) "
                                        "is ordered behind at the end of %s" % (s, last, user, ucl.get(s, 0), host.get(s, 0), call))
                touched = None; call = None
            # (other marks - "scenario ..." - only label the trace)
        elif k == "STREAM":
            s = int(t[1]); vc[s] = {}; stats["streams"] += 1
            if t[2] == "blocking":
                blocking.add(s)
        elif k == "STREAMDESTROY":
            dead.add(int(t[1])); blocking.discard(int(t[1]))
        elif k in ("K", "COPY", "COPY2D", "SET", "OP"):
            tick(int(t[1]))
            stats["kernels" if k == "K" else "ops" if k == "OP" else "copies"] += 1
        elif k == "RECORD":
            s, e = int(t[1]), int(t[2]); tick(s); ev[e] = dict(vc[s]); stats["records"] += 1
        elif k == "WAIT":
            s, e = int(t[1]), int(t[2]); stats["waits"] += 1
            if e not in ev:
                if call not in allow_unrecorded:
                    findings.append("unrecorded wait: stream %d waits for event %d that was never recorded (in %s)" % (s, e, call))
                tick(s)
            else:
                tick(s); _join(vc[s], ev[e])
        elif k == "HOSTSYNC":
            if t[1] == "stream":
                s = int(t[2])
                c = dict(vc.get(s, {}))
                if s == 0:
                    for b in blocking:
                        _join(c, vc.get(b, {}))
                _join(host, c)
            elif t[1] == "event":
                _join(host, ev.get(int(t[2]), {}))
            else:
                for c in vc.values():
                    _join(host, c)
        elif k in ("OOB", "BADLAUNCH", "BADFREE", "BADSTREAMDESTROY"):
            findings.append(raw.strip() + " (in %s)" % call)
        elif k == "EVENTDESTROY":
            pass
        else:
            findings.append("unknown trace line: " + raw.strip())
    return findings, stats
