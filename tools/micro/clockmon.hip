// Shader-clock monitor (a tool, not product): ONE wave that samples (s_memrealtime: constant 100 MHz, s_memtime: the shader clock) every ~ 100 us
// while other kernels run, so that a workload's matrix-pipe rate can be priced against the clock the chip actually held under it (DVFS) instead
// of the nominal 2.4 GHz.  Scalar instructions only: a wave streaming fp64 MFMAs on the same SIMD does not starve it (profiles/r06_experiments.md).
//   built and driven by tools/clock_probe.py
#include <hip/hip_runtime.h>
#include <cstdint>
__global__ void __launch_bounds__(64) clockmon_kernel(unsigned long long* out, int nsamp, int naps, volatile int* stop, int* count) {
  if (threadIdx.x != 0) return;
  int i = 0;
  for (; i < nsamp; i++) {
    out[2 * i] = __builtin_amdgcn_s_memrealtime();
    out[2 * i + 1] = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < naps; j++) __builtin_amdgcn_s_sleep(127);
    if (*stop) { i++; break; }
  }
  *count = i;
}
__global__ void mark_kernel(unsigned long long* marks, int idx) { if (threadIdx.x == 0) marks[idx] = __builtin_amdgcn_s_memrealtime(); }
static hipStream_t g_s; static unsigned long long* g_out; static unsigned long long* g_marks; static int* g_stop; static int* g_count; static int g_nsamp;
extern "C" int cm_start(int nsamp, int naps) {
  g_nsamp = nsamp;
  if (hipStreamCreateWithFlags(&g_s, hipStreamNonBlocking) != hipSuccess) return 1;
  if (hipMalloc((void**)&g_out, sizeof(unsigned long long) * 2 * nsamp) != hipSuccess) return 2;
  if (hipMalloc((void**)&g_marks, sizeof(unsigned long long) * 256) != hipSuccess) return 3;
  if (hipMalloc((void**)&g_count, sizeof(int)) != hipSuccess) return 4;
  if (hipHostMalloc((void**)&g_stop, sizeof(int), hipHostMallocMapped) != hipSuccess) return 5;
  *g_stop = 0;
  (void)hipMemset(g_marks, 0, sizeof(unsigned long long) * 256);
  hipLaunchKernelGGL(clockmon_kernel, dim3(1), dim3(64), 0, g_s, g_out, nsamp, naps, (volatile int*)g_stop, g_count);
  return (int)hipGetLastError();
}
extern "C" int cm_mark(void* stream, int idx) { hipLaunchKernelGGL(mark_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g_marks, idx); return (int)hipGetLastError(); }
extern "C" int cm_stop(unsigned long long* samples, unsigned long long* marks) {
  *g_stop = 1;
  if (hipStreamSynchronize(g_s) != hipSuccess) return -1;
  int n = 0;
  (void)hipMemcpy(&n, g_count, sizeof(int), hipMemcpyDeviceToHost);
  (void)hipMemcpy(samples, g_out, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost);
  (void)hipMemcpy(marks, g_marks, sizeof(unsigned long long) * 256, hipMemcpyDeviceToHost);
  return n;
}
