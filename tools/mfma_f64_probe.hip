// Probe (not product): (1) lane->element maps of v_mfma_f64_16x16x4_f64 found empirically,
// (2) back-to-back issue rate -> measured fp64 MFMA peak of the chip.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

__global__ void layout_k(const double* A, const double* B, double* D) {
  int l = threadIdx.x;
  // hypothesis: lane l supplies A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[l * 4 + r] = c[r];
}

template <int NACC>
__global__ void __launch_bounds__(256) peak_k(double* out, int iters, unsigned long long* cyc) {
  d4 acc[NACC];
  unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long w0 = wall_clock64();
  for (int i = 0; i < NACC; i++) acc[i] = (d4){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3 + 1.0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = __builtin_readcyclecounter() - t0; cyc[1] = wall_clock64() - w0; }
}

template <int NACC>
void run_peak(int blocks, int threads, const char* tag) {
  double* out; CK(hipMalloc(&out, sizeof(double) * blocks * threads));
  int iters = 100000;
  unsigned long long* cyc; CK(hipMalloc(&cyc, 16)); unsigned long long hc[2];
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  peak_k<NACC><<<blocks, threads>>>(out, iters, cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  peak_k<NACC><<<blocks, threads>>>(out, iters, cyc);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double flops = (double)blocks * (threads / 64) * iters * NACC * 2048.0;
  double waves = (double)blocks * (threads / 64);
  // cycles per MFMA per SIMD assuming 2.4 GHz and waves spread evenly over 1024 SIMDs
  double per_simd_mfma = waves * iters * NACC / 1024.0;
  CK(hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost));
  printf("   s_memtime cycles/MFMA(wave0)=%.1f  wallclock ticks=%llu (100MHz => %.3f ms) => shader clock %.0f MHz\n",
         (double)hc[0] / ((double)iters * NACC), hc[1], hc[1] / 1e5, (double)hc[0] / (hc[1] / 100.0));
  printf("%s: blocks=%d threads=%d nacc=%d  %.3f ms  %.2f TFLOP/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", tag, blocks, threads,
         NACC, ms, flops / ms * 1e-9, ms * 1e-3 * 2.4e9 / per_simd_mfma);
  CK(hipFree(out));
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d clock=%d kHz mem=%.1f GB arch=%s\n", p.name, p.multiProcessorCount, p.clockRate,
         p.totalGlobalMem / 1e9, p.gcnArchName);
  // ---- layout
  std::vector<double> A(64), B(64), D(256), ref(256);
  for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i * 4 + k] = 1.0 + i * 0.37 + k * 1.93;
  for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = 0.5 + k * 2.11 - j * 0.53;
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 16 + j]; ref[i * 16 + j] = s; }
  double *dA, *dB, *dD; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
  CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
  layout_k<<<1, 64>>>(dA, dB, dD); CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
  int ok_guide = 1, found_all = 1;
  for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
    double v = D[l * 4 + r]; int fi = -1, fj = -1;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) if (fabs(ref[i * 16 + j] - v) < 1e-9) { fi = i; fj = j; }
    if (fi < 0) found_all = 0;
    int gi = (l >> 4) + 4 * r, gj = l & 15;  // guide: col=lane&15,row=(lane>>4)+4*reg
    if (fi != gi || fj != gj) ok_guide = 0;
    if (l < 2 || l == 16 || l == 17 || l == 63) printf("lane %2d reg %d -> (row %2d, col %2d)\n", l, r, fi, fj);
  }
  printf("layout: all_found=%d matches_guide(col=l&15,row=(l>>4)+4*reg)=%d\n", found_all, ok_guide);
  // ---- peak
  run_peak<4>(256 * 1, 256, "1 wave/SIMD");
  run_peak<8>(256 * 1, 256, "1 wave/SIMD");
  run_peak<16>(256 * 1, 256, "1 wave/SIMD");
  run_peak<4>(256 * 2, 256, "2 waves/SIMD");
  run_peak<8>(256 * 2, 256, "2 waves/SIMD");
  run_peak<16>(256 * 4, 256, "4 waves/SIMD");
  return 0;
}
