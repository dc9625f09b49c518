"""SUMMA-GEMM timing driver (reference bench/matmult/summa_gemm.cpp:7-55): argv M N K c num_chunks iters.
One process per GPU under torch.distributed.run (RCCL), or a single process (d = c = 1: the local MFMA GEMM + the
plan's stream/event skeleton).  Prints the max-over-ranks time per call and the aggregate TFLOP/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
M, N, K, c, chunks, iters = (int(x) for x in (sys.argv[1:7] + ["8192", "8192", "8192", "1", "4", "5"][len(sys.argv) - 1:]))
world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
dist = None
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
from capital_amd import blas, summa, topo
from capital_amd.matrix import matrix
T = topo.square(c, 0, chunks)
d = T.d
A = matrix(K, M, d, d); B = matrix(N, K, d, d); Cm = matrix(N, M, d, d)
A.distribute_random(T.x, T.y, d, d, rank // T.c); B.distribute_random(T.x, T.y, d, d, 1000 + rank // T.c)
pack = blas.ArgPack_gemm(blas.Order.AblasColumnMajor, blas.Transpose.AblasNoTrans, blas.Transpose.AblasNoTrans, 1.0, 0.0)
summa.invoke(A, B, Cm, T, pack); torch.cuda.synchronize()
if dist: dist.barrier()
t0 = time.perf_counter()
for _ in range(iters): summa.invoke(A, B, Cm, T, pack)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / iters
if dist:
    tt = torch.tensor([t], device="cuda"); dist.all_reduce(tt, op=dist.ReduceOp.MAX); t = float(tt.item())
if rank == 0:
    print("summa_gemm M=%d N=%d K=%d grid %dx%dx%d chunks=%d: %.3f ms  %.1f TFLOP/s aggregate" % (M, N, K, d, d, T.c, chunks, t * 1e3, 2.0 * M * N * K / t / 1e12))
summa.release(T); T.close()
if dist: dist.destroy_process_group()
