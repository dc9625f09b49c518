"""TEST INFRASTRUCTURE: random configurations through the structural replay + race check + joint replay (run_scenarios.py / trace_check.py).
python tests/hipshim/fuzz_scenarios.py SEED COUNT  prints every scenario with findings."""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import run_scenarios as rs   # noqa: E402


def main(seed, count):
    rng = random.Random(seed)
    for i in range(count):
        kind = rng.choice(["cholinv", "dist", "dist", "dist2d", "dist2d", "dmp", "mpchol"])
        us = rng.choice([0, 1])
        if kind == "cholinv":
            n = rng.choice([rng.randint(65, 3000), 128 * rng.randint(8, 130)])
            ci, split, bc = rng.choice([-1, -1, 0, 1]), rng.choice([1, 1, 2]), rng.choice([0, -2, -3, -5])
            nb = rng.choice([64, 128, 256, 512])
            opts = [("nb", nb)]
            for k, vals in [("outer", [nb, 2 * nb, 4 * nb]), ("tail", [0, n // 8, n // 2]), ("depth2", [0, 1]), ("pair_rest", [0, 1]), ("use_sb", [0, 1]), ("lookahead", [0, 1, 1]),
                            ("fastdiag", [0, 1, 1]), ("inner_la", [0, 0, 1]), ("inv_overlap", [0, 1]), ("inv_fast", [0, 1, 1]), ("inv_start_m", [0, n // 2, 1 << 30]),
                            ("chain_coop", [0, 4, 32]), ("serial_m", [0, 0, n // 2]), ("fuse_copy", [0, 1]), ("reserve", [0, 0, 8]), ("reserve_m", [0, n // 2]), ("occ1_m", [0, 4096, 16384])]:
                if rng.random() < 0.4:
                    opts.append((k, rng.choice(vals)))
            rs.scenario("fuzz cholinv n=%d ci=%d split=%d bc=%d %s" % (n, ci, split, bc, dict(opts)), us)(lambda r, a=(n, ci, split, bc, tuple(opts)): rs.cholinv_case(r, *a))
        elif kind == "dist":
            nb = rng.choice([128, 256, 512]); P = rng.randint(2, 8); n = rng.choice([rng.randint(nb + 1, 6000), nb * rng.randint(2, 40)]); ci = rng.choice([-1, -1, 0, 1])
            opts = [(k, rng.choice(v)) for k, v in [("strip", [1, 2]), ("depth2", [0, 1]), ("safe", [0, 1]), ("ipc", [0, 1]), ("split", [1, 2])] if rng.random() < 0.5]
            g = "fuzz dist n=%d nb=%d P=%d ci=%d %s" % (n, nb, P, ci, dict(opts))
            for p in range(P):
                rs.scenario(g + " rank=%d" % p, us, g, p, P)(lambda r, a=(n, nb, P, p, tuple(opts), ci): rs.dist_case(r, *a))
        elif kind == "dist2d":
            nb = rng.choice([128, 256, 512]); Pr = rng.choice([1, 2, 2, 4]); Pc = Pr * rng.choice([k for k in (1, 2, 4, 8) if Pr * k <= 8])
            n = rng.choice([rng.randint(nb + 1, 6000), nb * rng.randint(2, 40)])
            opts = [(k, rng.choice(v)) for k, v in [("strip", [1, 2]), ("depth2", [0, 1]), ("safe", [0, 1]), ("ipc", [0, 1]), ("complete_inv", [0, 1]), ("split", [1, 2])] if rng.random() < 0.5]
            if Pr * Pc == 1:
                opts = [o for o in opts if o[0] != "ipc"]
            g = "fuzz dist2d n=%d nb=%d %dx%d %s" % (n, nb, Pr, Pc, dict(opts))
            for pr in range(Pr):
                for pc in range(Pc):
                    rs.scenario(g + " at (%d,%d)" % (pr, pc), us, g, pr * Pc + pc, Pr * Pc)(lambda r, a=(n, nb, Pr, Pc, pr, pc, tuple(opts)): rs.dist2d_case(r, *a))
        elif kind == "dmp":
            nb = rng.choice([128, 256, 512]); P = rng.randint(1, 8); n = 128 * rng.randint(max(2, nb // 128), 40)
            g = "fuzz dmp n=%d nb=%d P=%d" % (n, nb, P)
            for p in range(P):
                rs.scenario(g + " rank=%d" % p, us, g, p, P)(lambda r, a=(n, nb, P, p): rs.dmp_case(r, *a))
        else:
            n = 128 * rng.randint(8, 140)
            opts = [(k, rng.choice(v)) for k, v in [("strip", [1, 2]), ("split", [0, 1]), ("pair_rest", [0, 1]), ("solve3", [0, 1]), ("reserve", [0, 8]), ("chain_coop", [0, 32])] if rng.random() < 0.5]
            rs.scenario("fuzz mpchol n=%d %s" % (n, dict(opts)), us)(lambda r, a=(n, 8, tuple(opts)): rs.mpchol_case(r, *a))
    bad = [x for x in rs.RESULTS if x["findings"]]
    print("%d scenarios (incl. joint replays), %d with findings" % (len(rs.RESULTS), len(bad)))
    for x in bad[:30]:
        print(" *", x["name"])
        for f in x["findings"][:4]:
            print("     ", f[:400])


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]))
