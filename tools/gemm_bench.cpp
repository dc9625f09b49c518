// Standalone timing of cap_dgemm / cap_dsyrk (TN) for profiling with rocprofv3.
// usage: gemm_bench.bin m n k [syrk=0|1] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include "../include/capital_amd.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
__global__ void fill(double* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long x = (i + 1) * 6364136223846793005ull + seed * 1442695040888963407ull;
    x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 32;
    p[i] = ((double)(x >> 11) / 9007199254740992.0) * 2.0 - 1.0;
  }
}
int main(int argc, char** argv) {
  long m = atol(argv[1]), n = atol(argv[2]), k = atol(argv[3]);
  int syrk = argc > 4 ? atoi(argv[4]) : 0, reps = argc > 5 ? atoi(argv[5]) : 5;
  double *A, *B, *C;
  const long ld = k + (getenv("LDPAD") ? atol(getenv("LDPAD")) : 0);     // leading dimension of the K-contiguous operands
  CK(hipMalloc(&A, sizeof(double) * ld * m)); CK(hipMalloc(&B, sizeof(double) * ld * n)); CK(hipMalloc(&C, sizeof(double) * m * n));
  fill<<<2048, 256>>>(A, (size_t)ld * m, 1); fill<<<2048, 256>>>(B, (size_t)ld * n, 2); fill<<<2048, 256>>>(C, (size_t)m * n, 3);
  CK(hipDeviceSynchronize());
  // MASK_OFF=n: run on a stream whose CU mask has its first n bits (CU i of XCD i % 8) cleared
  hipStream_t st_ = nullptr;
  if (getenv("MASK_OFF")) {
    uint32_t mk[8]; for (int i = 0; i < 8; i++) mk[i] = 0xffffffffu;
    for (int b = 0; b < atoi(getenv("MASK_OFF")); b++) mk[b / 32] &= ~(1u << (b % 32));
    CK(hipExtStreamCreateWithCUMask(&st_, 8, mk));
  }
  const double beta = getenv("BETA") ? atof(getenv("BETA")) : 1.0;
  auto run = [&]() {
    int st = syrk ? cap_dsyrk(CAP_UPPER, CAP_TRANS, n, k, -1.0, A, ld, beta, C, m, st_)
                  : cap_dgemm(CAP_TRANS, CAP_NOTRANS, m, n, k, -1.0, A, ld, B, ld, beta, C, m, st_);
    if (st) { printf("status %d\n", st); exit(1); }
  };
  run(); CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st_));
  for (int i = 0; i < reps; i++) run();
  CK(hipEventRecord(e1, st_)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  double fl = syrk ? (double)n * (n + 1) * k : 2.0 * m * n * k;
  printf("%s m=%ld n=%ld k=%ld ld=%ld: %.3f ms %.2f TFLOP/s\n", syrk ? "syrk" : "gemm", m, n, k, ld, ms, fl / ms * 1e-9);
  return 0;
}
