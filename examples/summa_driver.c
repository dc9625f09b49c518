/* Plain-C host driver over the C ABI - the shape of the reference's bench/matmult/summa_gemm.cpp:7-55 (positional ints, the
 * three element-cyclic operands filled by distribute_random, a loop of summa::invoke calls behind a barrier) with the MPI / MKL
 * path replaced by libcapital_amd.so: cap_topo_create = topo::square, cap_summa_plan_create + cap_summa_dgemm = summa::invoke's
 * GEMM overload (summa.hpp:6-44), cap_summa_dtrmm / cap_summa_dsyrk + cap_util_transpose = its TRMM / SYRK overloads
 * (summa.hpp:46-161, util.hpp:232-247).
 *
 *   build: gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/summa_driver.c -Lcapital_amd/lib -lcapital_amd \
 *              -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/capital_amd/lib -Wl,-rpath,/opt/rocm/lib -o examples/summa_driver.bin
 *          add -DCAPITAL_WITH_MPI and an MPI compiler / -lmpi for one process per GPU (the 128-byte RCCL id travels by MPI_Bcast)
 *   run:   examples/summa_driver.bin M N K c layout num_chunks num_iter [validate]
 *          (c = depth of the d x d x c grid, as topo::square takes it; one rank: c = 1)
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#ifdef CAPITAL_WITH_MPI
#include <mpi.h>
#endif

#include "capital_amd.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP: %s (%s:%d)\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define CAPCHECK(x) do { int s_ = (x); if (s_ != CAP_OK) { fprintf(stderr, "capital_amd: %s (%s:%d)\n", cap_status_string(s_), __FILE__, __LINE__); return 3; } } while (0)

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char** argv) {
  int rank = 0, size = 1;
#ifdef CAPITAL_WITH_MPI
  MPI_Init(&argc, &argv);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank); MPI_Comm_size(MPI_COMM_WORLD, &size);
#endif
  if (argc < 8) { fprintf(stderr, "usage: %s M N K c layout num_chunks num_iter [validate]\n", argv[0]); return 1; }
  const int64_t M = atoll(argv[1]), N = atoll(argv[2]), K = atoll(argv[3]);
  const int c = atoi(argv[4]), layout = atoi(argv[5]), num_chunks = atoi(argv[6]), num_iter = atoi(argv[7]);
  const int validate = argc > 8 ? atoi(argv[8]) : 0;

  cap_comm* world = NULL;                                         /* MPI_COMM_WORLD */
#ifdef CAPITAL_WITH_MPI
  {
    int ndev = 0; HIPCHECK(hipGetDeviceCount(&ndev)); HIPCHECK(hipSetDevice(rank % ndev));
    unsigned char id[128];
    if (rank == 0) CAPCHECK(cap_comm_unique_id(id));
    MPI_Bcast(id, 128, MPI_BYTE, 0, MPI_COMM_WORLD);
    CAPCHECK(cap_comm_create(&world, id, rank, size, NULL));        /* ncclCommInitRank */
  }
#else
  CAPCHECK(cap_comm_create_self(&world));
#endif
  cap_topo* grid = NULL;                                          /* topo::square(MPI_COMM_WORLD, c, layout, num_chunks) */
  CAPCHECK(cap_topo_create(&grid, /*square*/0, world, c, layout, num_chunks));
  const int d = cap_topo_get(grid, 3), x = cap_topo_get(grid, 4), y = cap_topo_get(grid, 5);
  const int cc = cap_topo_get(grid, 2);

  cap_summa_plan* sp = NULL;
  CAPCHECK(cap_summa_plan_create(&sp, grid, M, N, K, num_chunks));
  int64_t ml, nl, kl;
  cap_summa_local_dims(sp, &ml, &nl, &kl);                         /* ceil(M/d), ceil(N/d), ceil(K/d): matrix.hpp:8-11 */
  double *A = NULL, *B = NULL, *C = NULL;                          /* matA(K, M, d, d), matB(N, K, d, d), matC(N, M, d, d) */
  HIPCHECK(hipMalloc((void**)&A, sizeof(double) * ml * kl));
  HIPCHECK(hipMalloc((void**)&B, sizeof(double) * kl * nl));
  HIPCHECK(hipMalloc((void**)&C, sizeof(double) * ml * nl));
  CAPCHECK(cap_fill_random(A, ml, M, K, x, y, d, d, rank / cc, NULL));          /* matA.distribute_random(x, y, d, d, rank / c) */
  CAPCHECK(cap_fill_random(B, kl, K, N, x, y, d, d, -(rank / cc), NULL));
  CAPCHECK(cap_fill_random(C, ml, M, N, x, y, d, d, -(rank / cc), NULL));
  if (rank == 0) printf("grid %d x %d x %d, local pieces %lld x %lld x %lld\n", d, d, cc, (long long)ml, (long long)nl, (long long)kl);

  for (int i = 0; i < num_iter; i++) {                             /* summa_gemm.cpp:40-51 */
    CAPCHECK(cap_comm_barrier(world, NULL));                       /* MPI_Barrier(MPI_COMM_WORLD) */
    HIPCHECK(hipDeviceSynchronize());
    const double t0 = now();
    CAPCHECK(cap_summa_dgemm(sp, 1.0, A, ml, B, kl, 0.0, C, ml, NULL));          /* summa::invoke(matA, matB, matC, topo, {NoTrans, NoTrans, 1, 0}) */
    HIPCHECK(hipDeviceSynchronize());
    const double t = now() - t0;
    if (rank == 0) printf("%lld %lld %lld %.6f s  %.2f TFLOP/s (2MNK, whole grid)\n", (long long)M, (long long)N, (long long)K, t, 2.0 * M * N * K / t / 1e12);
  }

  if (validate && size == 1) {
    /* one rank: the piece is the whole matrix - check C against the local operator (blas::engine::_gemm = cap_dgemm), then the
     * other two overloads against the same operators on a square case built from the same operands */
    double *D = NULL, *out = NULL, h[2];
    HIPCHECK(hipMalloc((void**)&D, sizeof(double) * ml * nl));
    HIPCHECK(hipMalloc((void**)&out, 2 * sizeof(double)));
    HIPCHECK(hipMemcpy(D, C, sizeof(double) * ml * nl, hipMemcpyDeviceToDevice));
    CAPCHECK(cap_dgemm(CAP_NOTRANS, CAP_NOTRANS, M, N, K, 1.0, A, ml, B, kl, -1.0, D, ml, NULL));     /* D = A B - C */
    CAPCHECK(cap_sumsq(D, ml, M, N, 0, 0, out, NULL));
    CAPCHECK(cap_sumsq(C, ml, M, N, 0, 0, out + 1, NULL));
    HIPCHECK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
    printf("gemm  ||A B - C||_F / ||C||_F = %.3e\n", sqrt(h[0]) / sqrt(h[1]));
    if (sqrt(h[0]) > 1e-13 * sqrt(h[1])) return 4;

    /* SYRK overload (summa.hpp:85-161; call site cholinv.hpp:128-133): S <- -A^T A + S, A = the K x N operand B */
    cap_summa_plan* sk = NULL;
    double *S = NULL, *S2 = NULL;
    CAPCHECK(cap_summa_plan_create(&sk, grid, N, N, K, num_chunks));
    HIPCHECK(hipMalloc((void**)&S, sizeof(double) * nl * nl));
    HIPCHECK(hipMalloc((void**)&S2, sizeof(double) * nl * nl));
    CAPCHECK(cap_fill_random(S, nl, N, N, x, y, d, d, 5, NULL));
    HIPCHECK(hipMemcpy(S2, S, sizeof(double) * nl * nl, hipMemcpyDeviceToDevice));
    CAPCHECK(cap_summa_dsyrk(sk, CAP_UPPER, CAP_TRANS, -1.0, B, kl, 1.0, S, nl, /*c_packed*/0, NULL));
    CAPCHECK(cap_dsyrk(CAP_UPPER, CAP_TRANS, N, K, -1.0, B, kl, 1.0, S2, nl, NULL));
    /* upper triangles must agree: D2 = S - S2 on the upper part */
    {
      double* I = NULL; int64_t j;
      HIPCHECK(hipMalloc((void**)&I, sizeof(double) * nl * nl));
      HIPCHECK(hipMemset(I, 0, sizeof(double) * nl * nl));
      { double one = 1.0; for (j = 0; j < N; j++) HIPCHECK(hipMemcpy(I + j + j * nl, &one, sizeof one, hipMemcpyHostToDevice)); }
      CAPCHECK(cap_dgemm(CAP_NOTRANS, CAP_NOTRANS, N, N, N, 1.0, S, nl, I, nl, -1.0, S2, nl, NULL));      /* S2 <- S I - S2 */
      CAPCHECK(cap_sumsq(S2, nl, N, N, 0, /*upper_only*/1, out, NULL));
      CAPCHECK(cap_sumsq(S, nl, N, N, 0, 1, out + 1, NULL));
      HIPCHECK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
      printf("syrk  ||summa - local||_F / ||.||_F (upper) = %.3e\n", sqrt(h[0]) / sqrt(h[1]));
      if (sqrt(h[0]) > 1e-13 * sqrt(h[1])) return 4;
      /* TRMM overload (summa.hpp:46-83; call site cholinv.hpp:114-120): W <- triu(S)^T W against cap_dtrmm */
      {
        cap_summa_plan* st = NULL;
        double *W = NULL, *W2 = NULL, *work = NULL;
        CAPCHECK(cap_summa_plan_create(&st, grid, N, N, N, num_chunks));
        HIPCHECK(hipMalloc((void**)&W, sizeof(double) * nl * nl));
        HIPCHECK(hipMalloc((void**)&W2, sizeof(double) * nl * nl));
        HIPCHECK(hipMalloc((void**)&work, sizeof(double) * cap_dtrmm_work_size(CAP_LEFT, N, N)));
        CAPCHECK(cap_fill_random(W, nl, N, N, x, y, d, d, 9, NULL));
        HIPCHECK(hipMemcpy(W2, W, sizeof(double) * nl * nl, hipMemcpyDeviceToDevice));
        /* upstream's call site transposes T's piece first (util::transpose, cholinv.hpp:115); on one rank the partner is oneself */
        CAPCHECK(cap_util_transpose(grid, S, S2, nl * nl, NULL));
        CAPCHECK(cap_summa_dtrmm(st, CAP_LEFT, CAP_UPPER, CAP_TRANS, CAP_NONUNIT, 0.5, S, nl, /*t_packed*/0, W, nl, NULL));
        CAPCHECK(cap_dtrmm(CAP_LEFT, CAP_UPPER, CAP_TRANS, CAP_NONUNIT, N, N, 0.5, S, nl, W2, nl, work, NULL));
        CAPCHECK(cap_dgemm(CAP_NOTRANS, CAP_NOTRANS, N, N, N, 1.0, W, nl, I, nl, -1.0, W2, nl, NULL));
        CAPCHECK(cap_sumsq(W2, nl, N, N, 0, 0, out, NULL));
        CAPCHECK(cap_sumsq(W, nl, N, N, 0, 0, out + 1, NULL));
        HIPCHECK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
        printf("trmm  ||summa - local||_F / ||.||_F = %.3e\n", sqrt(h[0]) / sqrt(h[1]));
        if (sqrt(h[0]) > 1e-13 * sqrt(h[1])) return 4;
        hipFree(W); hipFree(W2); hipFree(work);
        CAPCHECK(cap_summa_plan_destroy(st));
      }
      hipFree(I);
    }
    hipFree(S); hipFree(S2); hipFree(D); hipFree(out);
    CAPCHECK(cap_summa_plan_destroy(sk));
  }
  CAPCHECK(cap_summa_plan_destroy(sp));
  CAPCHECK(cap_topo_destroy(grid));
  CAPCHECK(cap_comm_destroy(world));
  hipFree(A); hipFree(B); hipFree(C);
#ifdef CAPITAL_WITH_MPI
  MPI_Finalize();
#endif
  return 0;
}
