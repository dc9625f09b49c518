"""Single-GPU REPLAY of one rank of the P-rank 1 x P Cholesky schedule (csrc/dist.hip) - a projection of the multi-GPU run measured on the
one GPU there is.  The real plan of rank r runs at full speed; the peers' contributions are copied out of a
finished factor R of the same matrix by the replay communicator (csrc/replay/replay_comm.hip -> capital_amd/lib/libcap_replay.so, built on the
product's cap_comm_create_callbacks) behind a link model (lat_us + bytes / link_GBps per collective; every peer's piece on its own xGMI
link), and a foreign owner's diagonal-block chain is a spin of `chain_us` behind this rank's own panel-stream position (dist option
remote_chain_us; "auto" = this rank's own measured chain time per block).

    python tools/replay.py [--n 65536] [--of 8] [--ranks 0,1,...] [--steps 3] [--link-gbps 100] [--lat-us 10] [--chain-us auto] [--occ1-m M] [--strip S]

What it measures: rank r's kernels next to each other as in the real run (bulk-bound strips, chain-bound strips, per-stream busy time, what is
exposed).  What it cannot: RCCL's own kernels and their CU share, arrival skew between real ranks, xGMI contention.  bench.py --replay-rank uses run().
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _replay_lib():
    from capital_amd import build as b
    path = b.REPLAY_LIB
    if not os.path.exists(path):
        path = b.build_replay(verbose=False)
    from capital_amd import _lib
    _lib.lib()                                    # libcapital_amd.so first (the replay library links against it)
    R = C.CDLL(path)
    R.cap_replay_create.restype = C.c_int
    R.cap_replay_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                                    C.c_void_p, C.c_double, C.c_double]
    R.cap_replay_stats.restype = C.c_int
    R.cap_replay_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    R.cap_replay_set_strip.restype = C.c_int
    R.cap_replay_set_strip.argtypes = [C.c_void_p, C.c_int]
    R.cap_replay_set_channels.restype = C.c_int
    R.cap_replay_set_channels.argtypes = [C.c_void_p, C.c_int]
    R.cap_replay_destroy.restype = None
    R.cap_replay_destroy.argtypes = [C.c_void_p]
    R.cap_replay2d_create.restype = C.c_int
    R.cap_replay2d_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                      C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_double, C.c_double]
    R.cap_replay2d_stats.restype = C.c_int
    R.cap_replay2d_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    R.cap_replay2d_set_strip.restype = C.c_int
    R.cap_replay2d_set_strip.argtypes = [C.c_void_p, C.c_int]
    R.cap_replay2d_destroy.restype = None
    R.cap_replay2d_destroy.argtypes = [C.c_void_p]
    return R


class ReplayComm:
    """what dist_cholesky.Context expects of a communicator: rank, size, handle"""

    def __init__(self, RL, rank, size, Rref, ldr, n, nb, strip, dinv, link_GBps, lat_us):
        from capital_amd import _lib
        self.RL, self.rank, self.size = RL, rank, size
        h, ctx = C.c_void_p(), C.c_void_p()
        _lib.check(RL.cap_replay_create(C.byref(h), C.byref(ctx), rank, size, Rref.data_ptr(), ldr, n, nb, strip, dinv.data_ptr(), link_GBps, lat_us),
                   "cap_replay_create")
        self.handle, self.ctx = h, ctx
        self._keep = (Rref, dinv)

    def stats(self):
        out = (C.c_double * 3)()
        self.RL.cap_replay_stats(self.ctx, out)
        return {"bytes_from_peers": out[0], "link_model_ms": out[1] / 1e3, "collectives": int(out[2])}

    def close(self):
        from capital_amd import _lib
        if self.handle:
            _lib.lib().cap_comm_destroy(self.handle)
            self.RL.cap_replay_destroy(self.ctx)
            self.handle = None


def reference_factor(n, nb):
    """(R as a [col, row] device tensor with ld = n, table of the diagonal blocks' inverses, single-GPU seconds per factor)"""
    import torch
    from capital_amd import cholinv, _lib
    from capital_amd.matrix import matrix
    L = _lib.lib()
    A = matrix(n, n, 1, 1)
    A.distribute_symmetric(0, 0, 1, 1, 0, True)
    pack = cholinv.info(-1, 1, 0, 'U')
    cholinv.factor(A, pack, None)
    assert pack.last_info() == 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cholinv.factor(A, pack, None)
    torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    Rm = cholinv.construct_R(pack)
    assert Rm.ld() == n, "the replay reads R with ld = n"
    R = Rm.data()
    pack._release()
    del A, pack
    torch.cuda.empty_cache()
    nblk = n // nb
    dinv = torch.zeros(nblk, nb, nb, dtype=torch.float64, device=R.device)
    work = torch.zeros(max(1, int(L.cap_dtrtri_work_size(nb))), dtype=torch.float64, device=R.device)
    s = torch.cuda.current_stream().cuda_stream
    for k in range(nblk):
        dinv[k].copy_(R[k * nb:(k + 1) * nb, k * nb:(k + 1) * nb])       # [col, row] view of the block = the column-major block
        _lib.check(L.cap_dtrtri(1, nb, dinv[k].data_ptr(), nb, work.data_ptr(), s), "cap_dtrtri")   # 1 = AlapackUpper
    dinv = torch.triu(dinv.transpose(1, 2)).transpose(1, 2).contiguous()   # strictly-lower part (row > col) zero, as the chain writes it
    torch.cuda.synchronize()
    return Rm, R, dinv, t1


def run(n=65536, P=8, ranks=None, nb=512, steps=3, warmup=1, link_GBps=100.0, lat_us=10.0, chain_us="auto", occ1_m=None, strip=None, verbose=False, channels=0):
    import torch
    from capital_amd import dist_cholesky, _lib
    L = _lib.lib()
    RL = _replay_lib()
    assert n % nb == 0
    Rm, Rref, dinv, t_single = reference_factor(n, nb)
    ranks = list(range(P)) if ranks is None else list(ranks)
    out = {"n": n, "P": P, "nb": nb, "link_GBps_per_link": link_GBps, "lat_us": lat_us, "collective_kernel_workgroups": channels, "single_gpu_ms": t_single * 1e3,
           "single_gpu_tf": n ** 3 / 3 / t_single / 1e12, "ranks": []}
    names = ["chains", "row_solves", "head_updates", "msg_broadcasts", "strip_exchanges", "bulk_updates"]
    for r in ranks:
        comm = ReplayComm(RL, r, P, Rref, n, n, nb, 2, dinv, link_GBps, lat_us)
        ctx = dist_cholesky.Context(n, nb, comm)
        ctx.set_option("safe", 1)
        if strip: ctx.set_option("strip", strip)
        if occ1_m is not None: ctx.set_option("occ1_m", occ1_m)
        RL.cap_replay_set_strip(comm.ctx, ctx.get_option("strip"))
        RL.cap_replay_set_channels(comm.ctx, int(channels))
        ctx.fill_symmetric(True)

        launches = []

        def timed(k):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(k): ctx.factor()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / k

        def profile():
            ctx.set_option("profile", 1)
            ctx.factor(); torch.cuda.synchronize()
            busy = (C.c_double * 6)()
            _lib.check(L.cap_dist_profile_streams(ctx.plan, busy), "cap_dist_profile_streams")
            nl, ms, fl = C.c_int64(0), C.c_double(0), C.c_double(0)
            _lib.check(L.cap_dist_profile(ctx.plan, C.byref(nl), C.byref(ms), C.byref(fl)), "cap_dist_profile")
            cnt = C.c_int64(0); msv = (C.c_double * 512)(); flv = (C.c_double * 512)()
            _lib.check(L.cap_dist_profile_launches(ctx.plan, msv, flv, 512, C.byref(cnt)), "cap_dist_profile_launches")
            launches[:] = [(msv[i], flv[i]) for i in range(min(512, cnt.value))]
            ctx.set_option("profile", 0)
            return list(busy), nl.value, ms.value, fl.value

        # pass 1: peers' chains cost nothing -> this rank's own chain time per block under its real contention
        ctx.set_option("remote_chain_us", 0)
        for _ in range(warmup): ctx.factor()
        assert ctx.last_info() == 0
        t_free = timed(1)
        busy0, _, _, _ = profile()
        own_blocks = len([k for k in range(n // nb) if k % P == r])
        chain_ms_per_block = busy0[0] / max(1, own_blocks)
        cu = int(round(chain_ms_per_block * 1e3)) if chain_us == "auto" else int(chain_us)
        ctx.set_option("remote_chain_us", cu)
        ctx.factor()
        comm.stats()
        t = timed(steps)
        st = comm.stats()
        busy, nl, bulk_ms, bulk_fl = profile()
        assert ctx.last_info() == 0
        # my columns of R against the single-GPU factor (another blocking of the same sums: to rounding)
        Rl = ctx.local_R_device()
        err, ref = 0.0, 0.0
        for lb in range(ctx.local_cols // nb):
            J = lb * P + r
            a = Rl[lb * nb:(lb + 1) * nb, :(J + 1) * nb]; b = Rref[J * nb:(J + 1) * nb, :(J + 1) * nb]
            err = max(err, float((a - b).abs().max())); ref = max(ref, float(b.abs().max()))
        del Rl
        rec = {"rank": r, "ms": t * 1e3, "ms_peer_chains_free": t_free * 1e3, "remote_chain_us": cu, "own_chain_ms_per_block": chain_ms_per_block,
               "busy_ms": dict(zip(names, [round(x, 2) for x in busy])), "bulk_launches": nl, "bulk_tf": (bulk_fl / bulk_ms / 1e9) if bulk_ms > 0 else None,
               "bulk_frac_of_peak": (bulk_fl / bulk_ms / 1e9 / 78.6) if bulk_ms > 0 else None,
               "not_bulk_ms": t * 1e3 - bulk_ms, "link_model_ms_per_step": st["link_model_ms"] / steps, "GB_from_peers_per_step": st["bytes_from_peers"] / steps / 1e9,
               "R_max_abs_diff_vs_single_gpu": err, "R_max_abs": ref}
        if verbose: print(json.dumps(rec), flush=True)
        if verbose and os.environ.get("REPLAY_LAUNCHES"):
            for i, (ms, fl) in enumerate(launches):
                print("  bulk launch %3d: %8.3f ms %8.2f GFLOP %6.1f TF" % (i, ms, fl / 1e9, fl / ms / 1e9 if ms > 0 else 0.0))
        out["ranks"].append(rec)
        _lib.lib().cap_dist_plan_destroy(ctx.plan); ctx.plan = None
        del ctx
        comm.close()
        torch.cuda.empty_cache()
    worst = max(x["ms"] for x in out["ranks"])
    out["projected_ms_max_over_ranks"] = worst
    out["projected_tf_whole_job"] = n ** 3 / 3 / (worst * 1e-3) / 1e12
    out["projected_frac_of_P_gpu_peak"] = out["projected_tf_whole_job"] / (78.6 * P)
    out["projected_speedup_vs_1gpu"] = t_single * 1e3 / worst
    del Rm
    return out


class _Comm:
    def __init__(self, handle, rank, size):
        self.handle, self.rank, self.size = handle, rank, size


def run2d(n=65536, Pr=2, Pc=4, ranks=None, nb=512, steps=3, warmup=1, link_GBps=100.0, lat_us=10.0, chain_us=500, solve_us=300, occ1_m=None, verbose=False):
    """The same projection for the Pr x Pc block-cyclic plan (csrc/dist2d.hip, safe mode): three replay communicators (world, my process row,
    my process column) whose broadcasts are generated from the plan's own loop; a foreign owner's chain (chain_us) and a foreign process row's
    block-row solve (solve_us) are spins behind my panel stream's position (plan options remote_chain_us / remote_solve_us; the defaults are what
    the 1 x 8 replay measured for a rank's own chain and row solve)."""
    import torch
    from capital_amd import dist_cholesky, _lib
    L = _lib.lib()
    RL = _replay_lib()
    assert n % nb == 0 and Pc % Pr == 0
    Rm, Rref, dinv, t_single = reference_factor(n, nb)
    P = Pr * Pc
    ranks = list(range(P)) if ranks is None else list(ranks)
    out = {"n": n, "grid": "%d x %d" % (Pr, Pc), "nb": nb, "link_GBps_per_link": link_GBps, "lat_us": lat_us, "remote_chain_us": chain_us, "remote_solve_us": solve_us,
           "single_gpu_ms": t_single * 1e3, "single_gpu_tf": n ** 3 / 3 / t_single / 1e12, "ranks": []}
    for r in ranks:
        hw, hr, hc, ctxp = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(RL.cap_replay2d_create(C.byref(hw), C.byref(hr), C.byref(hc), C.byref(ctxp), r, Pr, Pc, Rref.data_ptr(), n, n, nb, 2, dinv.data_ptr(), link_GBps, lat_us),
                   "cap_replay2d_create")
        world, row, col = _Comm(hw, r, P), _Comm(hr, r % Pc, Pc), _Comm(hc, r // Pc, Pr)
        ctx = dist_cholesky.Context2D(n, nb, world, Pr, row, col)
        ctx.set_option("safe", 1)
        ctx.set_option("remote_chain_us", int(chain_us)); ctx.set_option("remote_solve_us", int(solve_us))
        if occ1_m is not None: ctx.set_option("occ1_m", occ1_m)
        ctx.fill_symmetric(True)

        def timed(k):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(k): ctx.factor()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / k
        for _ in range(warmup): ctx.factor()
        assert ctx.last_info() == 0
        st0 = (C.c_double * 3)(); RL.cap_replay2d_stats(ctxp, st0)
        t = timed(steps)
        st = (C.c_double * 3)(); RL.cap_replay2d_stats(ctxp, st)
        assert ctx.last_info() == 0
        # my piece of R against the single-GPU factor
        Rl = ctx.local_R_device()[: ctx.local_cols, : ctx.local_rows]                      # [local col, local row]
        rows = torch.from_numpy(dist_cholesky.global_index_2d(n, nb, Pr, ctx.pr)).to(Rl.device)
        cols = torch.from_numpy(dist_cholesky.global_index_2d(n, nb, Pc, ctx.pc)).to(Rl.device)
        err, ref = 0.0, 0.0
        for c0 in range(0, cols.numel(), 4096):                                            # in slabs of columns: Rref[cols] would copy 8 GiB at once
            cc = cols[c0:c0 + 4096]
            want = Rref[cc][:, rows]
            err = max(err, float((Rl[c0:c0 + 4096] - want).abs().max())); ref = max(ref, float(want.abs().max()))
            del want
        del Rl
        rec = {"rank": r, "pr": ctx.pr, "pc": ctx.pc, "ms": t * 1e3, "link_model_ms_per_step": st[1] / 1e3 / steps, "GB_from_peers_per_step": st[0] / steps / 1e9,
               "launch_counts": ctx.launch_counts(), "R_max_abs_diff_vs_single_gpu": err, "R_max_abs": ref}
        if verbose: print(json.dumps(rec), flush=True)
        out["ranks"].append(rec)
        ctx.close()
        for h in (hw, hr, hc): L.cap_comm_destroy(h)
        RL.cap_replay2d_destroy(ctxp)
        del ctx
        torch.cuda.empty_cache()
    worst = max(x["ms"] for x in out["ranks"])
    out["projected_ms_max_over_ranks"] = worst
    out["projected_tf_whole_job"] = n ** 3 / 3 / (worst * 1e-3) / 1e12
    out["projected_frac_of_P_gpu_peak"] = out["projected_tf_whole_job"] / (78.6 * P)
    out["projected_speedup_vs_1gpu"] = t_single * 1e3 / worst
    del Rm
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536); ap.add_argument("--of", type=int, default=8); ap.add_argument("--ranks", default="")
    ap.add_argument("--steps", type=int, default=3); ap.add_argument("--link-gbps", type=float, default=100.0); ap.add_argument("--lat-us", type=float, default=10.0)
    ap.add_argument("--chain-us", default="auto"); ap.add_argument("--occ1-m", type=int, default=None); ap.add_argument("--strip", type=int, default=None)
    ap.add_argument("--channels", type=int, default=0, help="workgroups (512 threads, 32 KiB LDS) a collective's stand-in occupies for its modelled time: the CU share of an RCCL kernel (0: none)")
    ap.add_argument("--grid-rows", type=int, default=1, help="Pr of the Pr x Pc block-cyclic plan (1: the 1 x P plan)")
    a = ap.parse_args()
    if a.grid_rows > 1:
        res = run2d(a.n, a.grid_rows, a.of // a.grid_rows, [int(x) for x in a.ranks.split(",")] if a.ranks else None, 512, a.steps, 1, a.link_gbps, a.lat_us,
                    500 if a.chain_us == "auto" else int(a.chain_us), 300, a.occ1_m, verbose=True)
        print(json.dumps({k: v for k, v in res.items() if k != "ranks"}))
        sys.exit(0)
    res = run(a.n, a.of, [int(x) for x in a.ranks.split(",")] if a.ranks else None, 512, a.steps, 1, a.link_gbps, a.lat_us, a.chain_us, a.occ1_m, a.strip, verbose=True, channels=a.channels)
    print(json.dumps({k: v for k, v in res.items() if k != "ranks"}))
