// Which resource slows the diagonal-block chain down when it shares the chip with the bulk update?  (experiment, not product)
// Runs cap_dpotrf(n = 512) - the chain's kernels: leaf_cholinv, panel64_solve_update, small GEMMs - on a high-priority
// stream while a synthetic background load occupies every CU on another stream:
//   0 none | 1 MFMA only (registers)  | 2 LDS reads + MFMA | 3 HBM streaming (no MFMA) | 4 the real DSYRK trailing update
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "../include/capital_amd.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256, 2) load_kernel(double* buf, size_t elems, unsigned long long ticks, double a0) {
  extern __shared__ double lds[];     // 64 KiB: two of these workgroups fill a CU like the bulk kernel does
  d4 acc[16];
  for (int i = 0; i < 16; i++) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 * (threadIdx.x + 1), b = a0 * 3;
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = a0 * i;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  size_t pos = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
  double s = 0;
  while (wall_clock64() - t0 < ticks) {
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (MODE == 2) {
          const double2 v = *reinterpret_cast<const double2*>(&lds[((threadIdx.x * 2 + r * 512) & 8190)]);
          a = v.x + a0; b = v.y + a0;
        }
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      }
    }
    if (MODE == 3) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const double2 v = *reinterpret_cast<const double2*>(&buf[pos]);
        s += v.x + v.y;
        pos += (size_t)gridDim.x * 512;
        if (pos + 2 >= elems) pos = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
      }
    }
  }
  for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][3];
  if (s == 12345.678) buf[0] = s;
}

int main(int argc, char** argv) {
  const int n = 512, reps = 20;
  const long m = 32768, k = 1024;
  double *D, *Dsave, *W, *A, *Cm; int* info;
  CK(hipMalloc(&D, sizeof(double) * n * n)); CK(hipMalloc(&Dsave, sizeof(double) * n * n));
  CK(hipMalloc(&W, sizeof(double) * cap_dpotrf_work_size(n))); CK(hipMalloc(&info, 4));
  CK(hipMalloc(&A, sizeof(double) * k * m)); CK(hipMalloc(&Cm, sizeof(double) * m * m));
  std::vector<double> h((size_t)n * n, 0.01);
  for (int i = 0; i < n; i++) h[(size_t)i * n + i] = n;
  CK(hipMemcpy(Dsave, h.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
  CK(hipMemset(A, 0, sizeof(double) * k * m)); CK(hipMemset(Cm, 0, sizeof(double) * m * m));
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  // argv[1]: chain stream priority (h high, n normal, l low), argv[2]: background stream priority
  const char pc = argc > 1 ? argv[1][0] : 'h', pb = argc > 2 ? argv[2][0] : 'l';
  auto prio = [&](char c) { return c == 'h' ? hi : (c == 'l' ? lo : (lo + hi) / 2); };
  printf("priority range lo=%d hi=%d; chain stream %c (%d), background stream %c (%d)\n", lo, hi, pc, prio(pc), pb, prio(pb));
  hipStream_t sb, sp; CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, prio(pb))); CK(hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, prio(pc)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[5] = {"none", "MFMA only", "LDS reads + MFMA", "HBM streaming", "real DSYRK m=32768 K=1024"};
  for (int mode = 0; mode < 5; mode++) {
    CK(hipDeviceSynchronize());
    // workgroups live ~200 us (like a K = 1024 tile) and 20 launches x 20 rounds of 512 keep the chip full for ~80 ms,
    // so slots are vacated and refilled all the time like under the real bulk kernel
    const unsigned long long ticks = 20000ull;
    for (int r = 0; r < 20; r++) {
      if (mode == 1) hipLaunchKernelGGL(load_kernel<1>, dim3(512 * 20), dim3(256), 65536, sb, A, (size_t)k * m, ticks, 1e-9);
      if (mode == 2) hipLaunchKernelGGL(load_kernel<2>, dim3(512 * 20), dim3(256), 65536, sb, A, (size_t)k * m, ticks, 1e-9);
      if (mode == 3) hipLaunchKernelGGL(load_kernel<3>, dim3(512 * 20), dim3(256), 65536, sb, Cm, (size_t)m * m, ticks, 1e-9);
    }
    if (mode == 4) for (int r = 0; r < 5; r++) cap_dsyrk(CAP_UPPER, CAP_TRANS, m, k, -1.0, A, k, 1.0, Cm, m, sb);
    // let the load spread over the chip, then time the chain
    hipLaunchKernelGGL(load_kernel<0>, dim3(1), dim3(64), 65536, sp, A, 16, 200000ull, 0.0);   // 2 ms idle spacer on the chain stream
    float tot = 0; std::vector<float> all;
    for (int r = 0; r < reps; r++) {
      CK(hipMemcpyAsync(D, Dsave, sizeof(double) * n * n, hipMemcpyDeviceToDevice, sp));
      CK(hipEventRecord(e0, sp));
      int st = cap_dpotrf(CAP_UPPER, n, D, n, info, W, sp);
      if (st) { printf("dpotrf status %d\n", st); return 1; }
      CK(hipEventRecord(e1, sp)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; all.push_back(ms);
    }
    hipEvent_t eb; CK(hipEventCreate(&eb)); CK(hipEventRecord(eb, sb));
    const bool still_busy = hipEventQuery(eb) == hipErrorNotReady;
    CK(hipDeviceSynchronize());
    std::sort(all.begin(), all.end());
    printf("background %-28s: cap_dpotrf(512) chain avg %.3f ms  min %.3f  median %.3f  max %.3f   (background still running at the end: %s)\n",
           names[mode], tot / reps, all.front(), all[all.size() / 2], all.back(), still_busy ? "yes" : "NO");
  }
  return 0;
}
