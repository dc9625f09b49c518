/* Timing of the mixed-precision factorization (bf16 MFMA trailing updates) without Python:  tools/mp_bench.bin N reps
 *   build: gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/mp_bench.c -Lcapital_amd/lib -lcapital_amd -L/opt/rocm/lib -lamdhip64 -lm ... */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "capital_amd.h"
#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP: %s (%s:%d)\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define CAPCHECK(x) do { int s_ = (x); if (s_ != CAP_OK) { fprintf(stderr, "capital_amd: %s (%s:%d)\n", cap_status_string(s_), __FILE__, __LINE__); return 3; } } while (0)
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 65536;
  const int reps = argc > 2 ? atoi(argv[2]) : 3;
  double* A = NULL;
  HIPCHECK(hipMalloc((void**)&A, sizeof(double) * n * n));
  CAPCHECK(cap_fill_symmetric(A, n, n, 0, 0, 1, 1, NULL));
  cap_mpchol_plan* p = NULL;
  CAPCHECK(cap_mpchol_plan_create(&p, n, 8));
  CAPCHECK(cap_mpchol_factor(p, A, n, NULL));
  HIPCHECK(hipDeviceSynchronize());
  double best = 1e30;
  for (int i = 0; i < reps; i++) {
    const double t0 = now();
    CAPCHECK(cap_mpchol_factor(p, A, n, NULL));
    HIPCHECK(hipDeviceSynchronize());
    const double t = now() - t0;
    if (t < best) best = t;
  }
  int64_t info = 0, nl = 0; double ms = 0, fl = 0, by = 0;
  cap_mpchol_info(p, NULL, &info);
  CAPCHECK(cap_mpchol_set_option(p, "profile", 1));
  CAPCHECK(cap_mpchol_factor(p, A, n, NULL));
  HIPCHECK(hipDeviceSynchronize());
  CAPCHECK(cap_mpchol_profile(p, &nl, &ms, &fl, &by));
  printf("N=%lld mixed factor: best %.2f ms = %.1f TF fp64-equivalent | info %lld | bf16 update: %lld launches, %.2f ms total, %.1f TF, %.0f GB/s\n",
         (long long)n, best * 1e3, (double)n * n * n / 3.0 / best / 1e12, (long long)info, (long long)nl, ms, fl / ms * 1e-9, by / ms * 1e-6);
  if (argc > 3) {   /* solve with 8 right-hand sides; the residual B - A X recomputed with the skinny kernel (8 columns) and with the tile kernel (128) */
    const int64_t w = 128;
    double *B, *X, *R8, *R128, *X128, *nrm, h[3];
    HIPCHECK(hipMalloc((void**)&B, sizeof(double) * n * w)); HIPCHECK(hipMalloc((void**)&X, sizeof(double) * n * 8));
    HIPCHECK(hipMalloc((void**)&R8, sizeof(double) * n * 8)); HIPCHECK(hipMalloc((void**)&R128, sizeof(double) * n * w));
    HIPCHECK(hipMalloc((void**)&X128, sizeof(double) * n * w)); HIPCHECK(hipMalloc((void**)&nrm, sizeof(double) * 3));
    HIPCHECK(hipMemset(B, 0, sizeof(double) * n * w)); HIPCHECK(hipMemset(X128, 0, sizeof(double) * n * w));
    CAPCHECK(cap_fill_random(B, n, n, 8, 0, 0, 1, 1, 7, NULL));
    int it = 0; double rr = 0;
    HIPCHECK(hipDeviceSynchronize());
    const double t0 = now();
    CAPCHECK(cap_mpchol_solve(p, A, n, B, n, X, n, 8, 30, 1e-15, &it, &rr, NULL));
    HIPCHECK(hipDeviceSynchronize());
    const double ts = now() - t0;
    HIPCHECK(hipMemcpy(R8, B, sizeof(double) * n * 8, hipMemcpyDeviceToDevice));
    HIPCHECK(hipMemcpy(R128, B, sizeof(double) * n * w, hipMemcpyDeviceToDevice));
    HIPCHECK(hipMemcpy(X128, X, sizeof(double) * n * 8, hipMemcpyDeviceToDevice));
    CAPCHECK(cap_dgemm(CAP_TRANS, CAP_NOTRANS, n, 8, n, -1.0, A, n, X, n, 1.0, R8, n, NULL));
    CAPCHECK(cap_dgemm(CAP_TRANS, CAP_NOTRANS, n, w, n, -1.0, A, n, X128, n, 1.0, R128, n, NULL));
    CAPCHECK(cap_sumsq(B, n, n, 8, 0, 0, nrm, NULL)); CAPCHECK(cap_sumsq(R8, n, n, 8, 0, 0, nrm + 1, NULL)); CAPCHECK(cap_sumsq(R128, n, n, 8, 0, 0, nrm + 2, NULL));
    HIPCHECK(hipMemcpy(h, nrm, sizeof h, hipMemcpyDeviceToHost));
    printf("solve: %.2f ms, %d sweeps, relres %.3e | ||B - A X||/||B||: 8-column kernel %.3e, 128-column tile kernel %.3e\n", ts * 1e3, it, rr,
           sqrt(h[1] / h[0]), sqrt(h[2] / h[0]));
  }
  cap_mpchol_plan_destroy(p);
  return 0;
}
