"""Worker for the multi-rank tests: launched under torch.distributed.run with the gloo backend.

  --mode index : CPU only - block-cyclic maps + gloo assembly of a distributed matrix (no GPU, no kernels)
  --mode gpu   : every rank drives cuda:0 through the C ABI with the host-staged communicator and the
                 result is checked against the oracle on rank 0."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="index")
    ap.add_argument("--size", dest="n", type=int, default=1024)
    ap.add_argument("--nb", type=int, default=128)
    args = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, size = dist.get_rank(), dist.get_world_size()
    from oracle import capital_oracle as orc   # checker
    n, nb = args.n, args.nb

    if args.mode == "index":
        # import only the pure helpers (no library, no GPU)
        import importlib.util
        src = open(os.path.join(ROOT, "capital_amd", "dist_cholesky.py")).read()
        helpers = {}
        exec(compile(src.split("# ------------------------------------------------------------------ communicators")[0]
                     .replace("from . import _lib", "").replace("from ._util import cur_stream", ""), "helpers", "exec"), helpers)
        a = orc.symmetric_global(n, True)
        cols = helpers["global_cols_of_rank"](n, nb, size, rank)
        mine = torch.from_numpy(np.ascontiguousarray(a[:, cols]))
        # ragged all-gather through equal-sized padded pieces (what the GPU schedule does with RCCL)
        lc_max = max(helpers["global_cols_of_rank"](n, nb, size, r).size for r in range(size))
        pad = torch.zeros(n, lc_max, dtype=torch.float64); pad[:, : cols.size] = mine
        outs = [torch.empty_like(pad) for _ in range(size)]
        dist.all_gather(outs, pad)
        full = helpers["assemble_global"]([o.numpy() for o in outs], n, nb, size)
        assert np.array_equal(full, a)
        # every global block column has exactly one owner and a dense local slot
        nblk = (n + nb - 1) // nb
        seen = set()
        for J in range(nblk):
            seen.add((helpers["owner"](J, size), helpers["local_block"](J, size)))
        assert len(seen) == nblk
        t = torch.tensor([float(cols.size)]); dist.all_reduce(t)
        assert int(t.item()) == n
        if rank == 0:
            print("INDEX-OK world=%d n=%d nb=%d" % (size, n, nb), flush=True)
    else:
        torch.cuda.set_device(0)
        from capital_amd import dist_cholesky as dc
        comm = dc.HostStagedComm()
        ctx = dc.Context(n, nb, comm)
        ctx.fill_symmetric(True)
        a = orc.symmetric_global(n, True)
        cols = dc.global_cols_of_rank(n, nb, size, rank)
        torch.cuda.synchronize()
        assert np.array_equal(ctx.A[: ctx.local_cols].cpu().numpy().T, a[:, cols]), "block-cyclic generator mismatch"
        for rep in range(2):                       # plan reuse
            ctx.factor()
        info = ctx.last_info()
        rl = ctx.local_R()
        lc_max = max(dc.global_cols_of_rank(n, nb, size, r).size for r in range(size))
        pad = torch.zeros(n, lc_max, dtype=torch.float64); pad[:, : cols.size] = torch.from_numpy(rl)
        outs = [torch.empty_like(pad) for _ in range(size)]
        dist.all_gather(outs, pad)
        if rank == 0:
            R = np.triu(dc.assemble_global([o.numpy() for o in outs], n, nb, size))
            ref = np.linalg.cholesky(a).T
            err = np.linalg.norm(R - ref) / np.linalg.norm(ref)
            res = orc.cholesky_residual(a, R)
            assert info == 0, info
            assert err < 1e-13, err
            assert res < 1e-14, res
            print("DIST-OK world=%d n=%d nb=%d err=%.2e residual=%.2e collectives=%s" % (size, n, nb, err, res, comm.calls), flush=True)
        ctx.close(); comm.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
