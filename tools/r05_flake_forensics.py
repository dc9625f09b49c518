"""CPU forensics of the one failure of tests/dist_worker.py --mode mixed --size 2048 --nb 256 --hard 1 (round 5, 4 ranks):
   ||R_distributed - R_one_rank|| / ||R_one_rank|| was 0.29143640971.  Which state of the one-rank buffer gives that number?

   The mixed factorization is emulated in NumPy (fp64 diagonal blocks and row solves stored as fp32, bf16 panels, fp32 updates) and
   stopped at every stage of the first steps; dropped / doubled Schur updates are listed too.  Result: dropped updates give
   0.01 - 0.16, doubled ones 0.01 - 0.12, the un-factored matrix 0.2947, "diagonal block 0 factored" 0.2932, "block row 0 solved"
   0.2911 - the observed 0.29144 sits between the last two (6 of 7 column blocks of row 0 solved: 0.29136): the buffer was read
   while the FIRST step of the one-rank factorization was running.  Nothing under oracle/ or the product is used here."""
import numpy as np

n, nb = 2048, 256
g = np.random.default_rng(17).standard_normal((n, n))
a = g @ g.T / n + 0.5 * np.eye(n); a = 0.5 * (a + a.T)
nblk = n // nb


def bf16(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def mixed(stop=None, stage="full", drop_head=(), drop_bulk=(), twice_head=(), twice_bulk=()):
    R = np.triu(a).astype(np.float32)
    for k in range(nblk):
        s = slice(k * nb, (k + 1) * nb)
        if stop is not None and k > stop:
            break
        D = R[s, s].astype(np.float64); D = np.triu(D) + np.triu(D, 1).T
        try:
            Rkk = np.linalg.cholesky(D).T
        except np.linalg.LinAlgError:
            return None
        R[s, s] = Rkk.astype(np.float32)
        if stop == k and stage == "diag":
            break
        if k + 1 < nblk:
            S = np.linalg.solve(Rkk.T, R[s, (k + 1) * nb:].astype(np.float64))
            R[s, (k + 1) * nb:] = S.astype(np.float32)
            if stop == k and stage == "trsm":
                break
            P = bf16(S.astype(np.float32))
            upd = (P.T @ P).astype(np.float32)
            T = R[(k + 1) * nb:, (k + 1) * nb:]
            mh = 0 if k in drop_head else (2 if k in twice_head else 1)
            mb = 0 if k in drop_bulk else (2 if k in twice_bulk else 1)
            T[:nb, :] -= mh * upd[:nb, :]
            if stop == k and stage == "head":
                break
            T[nb:, nb:] -= mb * upd[nb:, nb:]
    return np.triu(R).astype(np.float64)


Rd = mixed()
ref = np.linalg.cholesky(a).T
print("emulated factor vs fp64: %.2e" % (np.linalg.norm(Rd - ref) / np.linalg.norm(ref)))


def d(R1):
    return float("nan") if R1 is None else np.linalg.norm(Rd - R1) / np.linalg.norm(R1)


print("observed            0.29143640971")
print("un-factored         %.5f" % d(np.triu(a).astype(np.float32).astype(np.float64)))
for k in range(3):
    for st in ("diag", "trsm", "head", "full"):
        print("stopped in step %d after %-5s %.5f" % (k, st, d(mixed(k, st))))
S0, D0 = mixed(0, "trsm"), mixed(0, "diag")
for j in range(8):
    R1 = D0.copy(); R1[:nb, nb:nb * (1 + j)] = S0[:nb, nb:nb * (1 + j)]
    print("block row 0 solved for %d of 7 column blocks  %.5f" % (j, d(R1)))
allk = tuple(range(nblk))
print("every update dropped     %.5f" % d(mixed(drop_head=allk, drop_bulk=allk)))
print("every bulk update dropped %.5f" % d(mixed(drop_bulk=allk)))
print("every head update dropped %.5f" % d(mixed(drop_head=allk)))
for k in range(nblk - 1):
    print("step %d: head dropped %.5f  bulk dropped %.5f  head twice %.5f  bulk twice %.5f" % (
        k, d(mixed(drop_head=(k,))), d(mixed(drop_bulk=(k,))), d(mixed(twice_head=(k,))), d(mixed(twice_bulk=(k,)))))
