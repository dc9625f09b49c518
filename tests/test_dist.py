"""Multi-rank coverage.
 not gpu : world_size-2/3 gloo runs of the block-cyclic index logic + ragged gather assembly (CPU only)
 gpu     : the real schedule (csrc/dist.hip, HIP kernels) with 1, 2 and 4 ranks sharing cuda:0 through the
           host-staged gloo communicator; result vs oracle.  The RCCL path differs only in the three
           collectives (cap_comm_*), exercised at P = 1 here and at P = 2/4/8 by the driver's scaling run."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(nproc, mode, n, nb, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), "--mode", mode, "--size", str(n),
           "--nb", str(nb)]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)


@pytest.mark.parametrize("nproc,n,nb", [(2, 1024, 128), (3, 1000, 100), (2, 640, 256)])
def test_block_cyclic_index_logic_gloo(nproc, n, nb):
    r = _launch(nproc, "index", n, nb, 29611 + nproc)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "INDEX-OK" in r.stdout


def test_index_helpers_match_library_when_built():
    """The pure-Python maps agree with the C helpers the schedule uses (cap_bc_*); skipped if the .so is absent."""
    so = os.path.join(ROOT, "capital_amd", "lib", "libcapital_amd.so")
    if not os.path.exists(so):
        pytest.skip("library not built")
    from capital_amd import _lib, dist_cholesky as dc
    L = _lib.lib()
    for (n, nb, P) in [(4096, 512, 8), (1024, 128, 3), (65536, 512, 8), (2048, 256, 1)]:
        nblk = (n + nb - 1) // nb
        tot = 0
        for p in range(P):
            assert L.cap_bc_num_local_cols(n, nb, P, p) == dc.num_local_blocks(nblk, P, p) * nb
            tot += dc.global_cols_of_rank(n, nb, P, p).size
        assert tot == n
        for J in range(0, nblk, 3):
            assert L.cap_bc_owner(J, P) == dc.owner(J, P) and L.cap_bc_local_block(J, P) == dc.local_block(J, P)


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,n,nb", [(1, 1024, 128), (2, 1024, 128), (4, 2048, 128), (3, 1536, 256), (2, 2048, 512),
                                        (8, 8192, 512)])   # the driver's 8-GPU shape: P = 8, nb = 512 (2 block columns per rank)
def test_multirank_schedule_on_one_gpu(nproc, n, nb):
    r = _launch(nproc, "gpu", n, nb, 29621 + nproc)
    assert r.returncode == 0, (r.stdout[-3000:] + r.stderr[-3000:])
    assert "DIST-OK" in r.stdout, r.stdout[-2000:]


@pytest.mark.gpu
def test_single_rank_dist_path_matches_single_gpu_plan():
    """P = 1 through the RCCL-less self communicator: same R as the single-GPU plan, at a larger size."""
    import torch
    from capital_amd import cholinv, dist_cholesky as dc
    from capital_amd.matrix import matrix
    n, nb = 4096, 512
    ctx = dc.setup(n, nb)
    ctx.factor()
    assert ctx.last_info() == 0
    R1 = np.triu(ctx.local_R())
    A = matrix(n, n, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
    pack = cholinv.info(-1, 1, -3, 'U'); cholinv.factor(A, pack, None)
    R2 = cholinv.construct_R(pack).to_numpy()
    assert np.linalg.norm(R1 - R2) / np.linalg.norm(R2) < 1e-14
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,m,n", [(2, 4096, 64), (4, 10000, 48), (3, 6000, 128)])
def test_multirank_cacqr_on_one_gpu(nproc, m, n):
    """CholeskyQR2 1D: row-cyclic pieces on several ranks, Gram all-reduce through the (host-staged) communicator."""
    r = _launch(nproc, "cacqr", m, n, 29641 + nproc)
    assert r.returncode == 0, (r.stdout[-3000:] + r.stderr[-3000:])
    assert "CACQR-OK" in r.stdout, r.stdout[-2000:]


@pytest.mark.gpu
def test_bench_multi_gpu_code_path_emulated():
    """bench.py --gpus 2 end to end (rank bootstrap, timing reduce, JSON) with both ranks on cuda:0 (gloo + host-staged
    collectives instead of RCCL); the numbers are meaningless, the contract fields are checked."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29655", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--size", "4096",
           "--no-cpu-baseline"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", CAPITAL_BENCH_EMULATE="1", OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["unit"] == "TFLOP/s" and d["value"] > 0
    assert d["config"]["info"] == 0 and d["dtype"] == "f64" and d["higher_is_better"] is True
