// Probe 2 (not product): what limits fp64 throughput on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

// MODE 0: mfma 16x16x4 ; 1: mfma 4x4x4_4b ; 2: v_fma_f64 VALU
template <int MODE, int NACC>
__global__ void __launch_bounds__(1024) k(double* out, int iters, unsigned long long* cyc, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 * (threadIdx.x + 1), b = b0 * (threadIdx.x + 3) + b0;
  unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      if (MODE == 1) { double t = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i][0], 0, 0, 0); acc[i][0] = t; }
      if (MODE == 2) { acc[i][0] = __builtin_fma(a, b, acc[i][0]); acc[i][1] = __builtin_fma(a, b, acc[i][1]);
                       acc[i][2] = __builtin_fma(a, b, acc[i][2]); acc[i][3] = __builtin_fma(a, b, acc[i][3]); }
    }
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = __builtin_readcyclecounter() - t0; cyc[1] = wall_clock64() - w0; }
}

template <int MODE, int NACC>
void run(int blocks, int threads, double a0, double b0, const char* tag) {
  double* out; CK(hipMalloc(&out, sizeof(double) * blocks * threads));
  unsigned long long* cyc; CK(hipMalloc(&cyc, 16)); unsigned long long hc[2];
  int iters = (MODE == 2) ? 400000 : (MODE == 1 ? 200000 : 50000);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<MODE, NACC><<<blocks, threads>>>(out, iters, cyc, a0, b0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k<MODE, NACC><<<blocks, threads>>>(out, iters, cyc, a0, b0);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost));
  double per = (MODE == 0) ? 2048.0 : (MODE == 1 ? 512.0 : 4 * 128.0);
  double flops = (double)blocks * (threads / 64) * iters * NACC * per;
  printf("%-28s mode=%d blocks=%d thr=%d nacc=%d: %8.3f ms %7.2f TFLOP/s  cyc/op(wave0)=%.1f clk=%.0f MHz\n", tag, MODE, blocks,
         threads, NACC, ms, flops / ms * 1e-9, (double)hc[0] / ((double)iters * NACC), (double)hc[0] / (hc[1] / 100.0));
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  // waves/SIMD sweep with ONE block per CU (threads = 256*w)
  run<0, 8>(256, 256, 1e-3, 2e-3, "mfma16 1w/SIMD");
  run<0, 8>(256, 512, 1e-3, 2e-3, "mfma16 2w/SIMD");
  run<0, 8>(256, 768, 1e-3, 2e-3, "mfma16 3w/SIMD");
  run<0, 8>(256, 1024, 1e-3, 2e-3, "mfma16 4w/SIMD");
  run<0, 8>(512, 1024, 1e-3, 2e-3, "mfma16 8w/SIMD");
  run<0, 8>(256, 512, 0.0, 0.0, "mfma16 2w/SIMD zeros");
  run<0, 8>(256, 256, 0.0, 0.0, "mfma16 1w/SIMD zeros");
  run<0, 8>(64, 256, 1e-3, 2e-3, "mfma16 1w/SIMD 64 CUs only");
  run<0, 8>(64, 512, 1e-3, 2e-3, "mfma16 2w/SIMD 64 CUs only");
  run<0, 8>(1, 64, 1e-3, 2e-3, "mfma16 single wave");
  run<0, 8>(1, 128, 1e-3, 2e-3, "mfma16 two waves (2 SIMDs)");
  run<1, 8>(256, 256, 1e-3, 2e-3, "mfma4x4x4 1w/SIMD");
  run<1, 8>(256, 512, 1e-3, 2e-3, "mfma4x4x4 2w/SIMD");
  run<1, 8>(1, 64, 1e-3, 2e-3, "mfma4x4x4 single wave");
  run<2, 8>(256, 256, 1e-3, 2e-3, "valu fma64 1w/SIMD");
  run<2, 8>(256, 512, 1e-3, 2e-3, "valu fma64 2w/SIMD");
  run<2, 8>(256, 1024, 1e-3, 2e-3, "valu fma64 4w/SIMD");
  run<2, 8>(1, 64, 1e-3, 2e-3, "valu fma64 single wave");
  return 0;
}
