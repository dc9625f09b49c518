/* Plain-C host driver over the C ABI - the shape of the reference's bench/qr/cacqr.cpp:8-77 (the same twelve positional ints: variant,
 * rows, columns, rep_factor range, cholinv knobs, layout, chunks, iterations; distribute_random input with key = rank / c; warm-up
 * calls, one timed factor with the max over the ranks, and the validation block upstream keeps commented out) with the MPI / MKL path
 * replaced by libcapital_amd.so: cap_topo_create(kind 1) = topo::rect, cap_cacqr_plan_create(_grid) = cacqr::info, cap_cacqr_factor =
 * cacqr::factor (cacqr.hpp:217-248).
 *
 *   build: gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/cacqr_driver.c -Lcapital_amd/lib -lcapital_amd \
 *              -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/capital_amd/lib -Wl,-rpath,/opt/rocm/lib -o examples/cacqr_driver.bin
 *          add -DCAPITAL_WITH_MPI and an MPI compiler / -lmpi for one process per GPU (the 128-byte RCCL id travels by MPI_Bcast)
 *   run:   examples/cacqr_driver.bin variant rows columns rep_start rep_end complete_inv split bc_start bc_end layout num_chunks num_iter [validate]
 *          (complete_inv, split, bcMult are accepted for upstream's command line: the n x n Cholesky factor and its inverse are
 *           computed redundantly per GPU by the in-LDS / one-launch chain, there is no recursion to tune)
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
  if (argc < 13) {
    fprintf(stderr, "usage: %s variant rows columns rep_start rep_end complete_inv split bc_start bc_end layout num_chunks num_iter [validate]\n", argv[0]);
    return 1;
  }
  const int variant = atoi(argv[1]);                               /* 1 - cacqr, 2 - cacqr2 */
  const int64_t num_rows = atoll(argv[2]), num_columns = atoll(argv[3]);
  const int rep_start = atoi(argv[4]), rep_end = atoi(argv[5]);
  const int bc_start = atoi(argv[8]), bc_end = atoi(argv[9]);
  const int layout = atoi(argv[10]), num_chunks = atoi(argv[11]), num_iter = atoi(argv[12]);
  const int validate = argc > 13 ? atoi(argv[13]) : 0;

  cap_comm* world = NULL;
#ifdef CAPITAL_WITH_MPI
  {
    int ndev = 0; HIPCHECK(hipGetDeviceCount(&ndev)); HIPCHECK(hipSetDevice(rank % ndev));
    unsigned char id[128];
    if (rank == 0) CAPCHECK(cap_comm_unique_id(id));
    MPI_Bcast(id, 128, MPI_BYTE, 0, MPI_COMM_WORLD);
    CAPCHECK(cap_comm_create(&world, id, rank, size, NULL));
  }
#else
  CAPCHECK(cap_comm_create_self(&world));
#endif

  for (int rep = rep_start; rep <= rep_end; rep++) {               /* cacqr.cpp:31 */
    cap_topo* grid = NULL;                                         /* topo::rect(MPI_COMM_WORLD, rep, layout, num_chunks): c x d x c */
    CAPCHECK(cap_topo_create(&grid, /*rect*/1, world, rep, layout, num_chunks));
    const int c = cap_topo_get(grid, 2), d = cap_topo_get(grid, 3), x = cap_topo_get(grid, 4), y = cap_topo_get(grid, 5);
    const int64_t ml = (num_rows + d - 1) / d, nl = (num_columns + c - 1) / c;   /* matrix A(num_columns, num_rows, c, d): matrix.hpp:8-11 */
    double* A = NULL;
    HIPCHECK(hipMalloc((void**)&A, sizeof(double) * ml * nl));
    for (int bc = bc_start; bc <= bc_end; bc++) {
      cap_cacqr_plan* pack = NULL;                                 /* qr_type::info pack(variant, ci_pack) */
      if (c == 1) CAPCHECK(cap_cacqr_plan_create(&pack, ml, num_columns, variant, world));             /* 1D: invoke_1d, cacqr.hpp:172-193 */
      else CAPCHECK(cap_cacqr_plan_create_grid(&pack, num_rows, num_columns, variant, grid));          /* 3D / tunable: invoke_3d, :195-215 */
      double t = 0.0;
      for (int k = 0; k <= num_iter; k++) {                        /* num_iter warm-ups, then the timed call (cacqr.cpp:43-53) */
        CAPCHECK(cap_fill_random(A, ml, num_rows, num_columns, x, y, c, d, rank / c, NULL));           /* A.distribute_random(x, y, c, d, rank / c) */
        CAPCHECK(cap_comm_barrier(world, NULL));
        HIPCHECK(hipDeviceSynchronize());
        const double t0 = now();
        CAPCHECK(cap_cacqr_factor(pack, A, ml, NULL));
        HIPCHECK(hipDeviceSynchronize());
        t = now() - t0;
      }
#ifdef CAPITAL_WITH_MPI
      MPI_Allreduce(MPI_IN_PLACE, &t, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
#endif
      int64_t info = 0;
      CAPCHECK(cap_cacqr_info(pack, NULL, &info));
      if (rank == 0) printf("%lld %lld %d %d %.6f s  %.2f TFLOP/s (%d m n^2)  info=%lld\n", (long long)num_rows, (long long)num_columns, rep, bc, t,
                            2.0 * variant * num_rows * (double)num_columns * num_columns / t / 1e12, 2 * variant, (long long)info);
      if (validate && size == 1) {
        /* qr::validate::residual / orthogonality (test/qr/validate.hpp:24-31,46-51) on one rank: ||Q R - A||_F / ||A||_F, ||Q^T Q - I||_F / sqrt(n) */
        int64_t ldq = 0, ldr = 0;
        double* Q = cap_cacqr_Q_ptr(pack, &ldq); double* R = cap_cacqr_R_ptr(pack, &ldr);
        double *D = NULL, *G = NULL, *out = NULL, h[3];
        const int64_t n = num_columns;
        HIPCHECK(hipMalloc((void**)&D, sizeof(double) * ml * n));
        HIPCHECK(hipMalloc((void**)&G, sizeof(double) * n * n));
        HIPCHECK(hipMalloc((void**)&out, 3 * sizeof(double)));
        HIPCHECK(hipMemcpy(D, A, sizeof(double) * ml * n, hipMemcpyDeviceToDevice));
        CAPCHECK(cap_dgemm(CAP_NOTRANS, CAP_NOTRANS, num_rows, n, n, 1.0, Q, ldq, R, ldr, -1.0, D, ml, NULL));   /* D = Q R - A */
        CAPCHECK(cap_sumsq(D, ml, num_rows, n, 0, 0, out, NULL));
        CAPCHECK(cap_sumsq(A, ml, num_rows, n, 0, 0, out + 1, NULL));
        CAPCHECK(cap_dgemm(CAP_TRANS, CAP_NOTRANS, n, n, num_rows, 1.0, Q, ldq, Q, ldq, 0.0, G, n, NULL));       /* G = Q^T Q */
        CAPCHECK(cap_sumsq(G, n, n, n, /*sub_identity*/1, 0, out + 2, NULL));
        HIPCHECK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
        const double res = sqrt(h[0]) / sqrt(h[1]), orth = sqrt(h[2]) / sqrt((double)n);
        printf("residual %.3e orthogonality %.3e\n", res, orth);
        hipFree(D); hipFree(G); hipFree(out);
        /* one sweep (variant 1) leaves ||Q^T Q - I|| ~ eps kappa(A)^2; the second sweep of CholeskyQR2 brings it to eps */
        if (!(res < 1e-13) || !(orth < (variant == 2 ? 1e-14 : 1e-11)) || info != 0) return 4;
      }
      CAPCHECK(cap_cacqr_plan_destroy(pack));
    }
    hipFree(A);
    CAPCHECK(cap_topo_destroy(grid));
  }
  CAPCHECK(cap_comm_destroy(world));
#ifdef CAPITAL_WITH_MPI
  MPI_Finalize();
#endif
  return 0;
}
