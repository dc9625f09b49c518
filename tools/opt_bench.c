/* A/B runs of the factorization's schedule options without Python (no torch import on a fresh box): one matrix, several
 * option sets, each on its own plan.
 *   build: gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/opt_bench.c -Lcapital_amd/lib -lcapital_amd \
 *              -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/capital_amd/lib -Wl,-rpath,/opt/rocm/lib -o tools/opt_bench.bin
 *   run:   tools/opt_bench.bin N complete_inv reps [-- key=value ...]...      (an empty set = the defaults)
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "capital_amd.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP: %s (%s:%d)\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define CAPCHECK(x) do { int s_ = (x); if (s_ != CAP_OK) { fprintf(stderr, "capital_amd: %s (%s:%d)\n", cap_status_string(s_), __FILE__, __LINE__); return 3; } } while (0)

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s N complete_inv reps [-- key=value ...]...\n", argv[0]); return 1; }
  const int64_t n = atoll(argv[1]);
  const int ci = atoi(argv[2]);
  const int reps = atoi(argv[3]);
  double *A = NULL, *R = NULL, *work = NULL, *out = NULL;
  HIPCHECK(hipMalloc((void**)&A, sizeof(double) * n * n));
  CAPCHECK(cap_fill_symmetric(A, n, n, 0, 0, 1, 1, NULL));
  HIPCHECK(hipMalloc((void**)&R, sizeof(double) * n * n));
  HIPCHECK(hipMalloc((void**)&work, sizeof(double) * n * n));
  HIPCHECK(hipMalloc((void**)&out, 2 * sizeof(double)));
  int a = 4;
  do {
    if (a < argc && !strcmp(argv[a], "--")) a++;
    cap_cholinv_plan* pack = NULL;
    CAPCHECK(cap_cholinv_plan_create(&pack, n, ci, 1, 0, 'U', NULL));
    char desc[512] = "";
    for (; a < argc && strcmp(argv[a], "--"); a++) {
      char kv[128]; strncpy(kv, argv[a], sizeof kv - 1); kv[sizeof kv - 1] = 0;
      char* eq = strchr(kv, '=');
      if (!eq) { fprintf(stderr, "bad option %s\n", kv); return 1; }
      *eq = 0;
      CAPCHECK(cap_cholinv_set_option(pack, kv, atoll(eq + 1)));
      strncat(desc, argv[a], sizeof desc - strlen(desc) - 2); strcat(desc, " ");
    }
    CAPCHECK(cap_cholinv_factor(pack, A, n, NULL));
    HIPCHECK(hipDeviceSynchronize());
    double best = 1e30, sum = 0;
    for (int i = 0; i < reps; i++) {
      const double t0 = now();
      CAPCHECK(cap_cholinv_factor(pack, A, n, NULL));
      HIPCHECK(hipDeviceSynchronize());
      const double t = now() - t0;
      sum += t; if (t < best) best = t;
    }
    int64_t info = 0;
    CAPCHECK(cap_cholinv_info(pack, NULL, &info));
    double h[2];
    CAPCHECK(cap_cholinv_get_R(pack, R, n, NULL));
    CAPCHECK(cap_cholesky_residual_terms(A, n, R, n, n, work, out, NULL));
    HIPCHECK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
    const double fl = (double)n * n * n / 3.0;
    printf("N=%lld ci=%d [%s]: mean %.2f ms %.2f TF | best %.2f ms %.2f TF | info %lld residual %.2e\n", (long long)n, ci, desc,
           sum / reps * 1e3, fl / (sum / reps) / 1e12, best * 1e3, fl / best / 1e12, (long long)info, sqrt(h[0]) / sqrt(h[1]));
    fflush(stdout);
    CAPCHECK(cap_cholinv_plan_destroy(pack));
  } while (a < argc);
  return 0;
}
