/* Plain-C host driver over the C ABI - the shape of the reference's bench/cholesky/cholinv.cpp:8-71
 * (positional ints, warm-up call, timed loop, optional residual like the commented-out validation
 * block :61-66), with the MPI/MKL path replaced by libcapital_amd.so on one MI355X.
 *
 *   build: gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/cholinv_driver.c \
 *              -Lcapital_amd/lib -lcapital_amd -L/opt/rocm/lib -lamdhip64 -lm \
 *              -Wl,-rpath,$PWD/capital_amd/lib -Wl,-rpath,/opt/rocm/lib -o examples/cholinv_driver.bin
 *   run:   examples/cholinv_driver.bin N complete_inv split bcMult num_iter [validate]
 *          (complete_inv = -1: blocked Cholesky without the explicit inverse)
 */
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
  if (argc < 6) { fprintf(stderr, "usage: %s N complete_inv split bcMult num_iter [validate]\n", argv[0]); return 1; }
  const int64_t n = atoll(argv[1]);
  const int complete_inv = atoi(argv[2]);
  const int64_t split = atoll(argv[3]), bc = atoll(argv[4]);
  const int num_iter = atoi(argv[5]);
  const int validate = argc > 6 ? atoi(argv[6]) : 0;

  char name[128]; int cus = 0; int64_t hbm = 0;
  CAPCHECK(cap_device_info(name, sizeof name, &cus, &hbm));
  printf("device: %s, %d CUs, %.0f GB\n", name, cus, hbm / 1e9);

  double* A = NULL;                                              /* matrix<double,int64_t,rect> A(N,N,1,1) */
  HIPCHECK(hipMalloc((void**)&A, sizeof(double) * n * n));
  CAPCHECK(cap_fill_symmetric(A, n, n, 0, 0, 1, 1, NULL));       /* A.distribute_symmetric(..., true) */

  cap_cholinv_plan* pack = NULL;                                 /* cholinv<...>::info pack(complete_inv, split, bcMult, 'U') */
  CAPCHECK(cap_cholinv_plan_create(&pack, n, complete_inv, split, bc, 'U', NULL));
  CAPCHECK(cap_cholinv_factor(pack, A, n, NULL));                /* warm-up (cholinv.cpp:44) */
  int64_t info = 0;
  CAPCHECK(cap_cholinv_info(pack, NULL, &info));

  for (int i = 0; i < num_iter; i++) {                           /* cholinv.cpp:46-60 */
    HIPCHECK(hipDeviceSynchronize());
    const double t0 = now();
    CAPCHECK(cap_cholinv_factor(pack, A, n, NULL));
    HIPCHECK(hipDeviceSynchronize());
    const double t = now() - t0;
    printf("%lld %.6f s  %.2f TFLOP/s (N^3/3)\n", (long long)n, t, (double)n * n * n / 3.0 / t / 1e12);
  }

  if (validate) {                                                /* cholesky::validate<...>::residual (test/cholesky/validate.hpp:33-46) */
    double *R = NULL, *work = NULL, *out = NULL, h[2];
    HIPCHECK(hipMalloc((void**)&R, sizeof(double) * n * n));
    HIPCHECK(hipMalloc((void**)&work, sizeof(double) * n * n));
    HIPCHECK(hipMalloc((void**)&out, 2 * sizeof(double)));
    CAPCHECK(cap_cholinv_get_R(pack, R, n, NULL));               /* construct_R */
    CAPCHECK(cap_cholesky_residual_terms(A, n, R, n, n, work, out, NULL));
    HIPCHECK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
    printf("residual ||A - R^T R||_F / ||A||_F (upper) = %.3e\n", sqrt(h[0]) / sqrt(h[1]));
    hipFree(R); hipFree(work); hipFree(out);
  }
  CAPCHECK(cap_cholinv_plan_destroy(pack));
  hipFree(A);
  return 0;
}
