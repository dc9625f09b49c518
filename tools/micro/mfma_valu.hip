// Microbenchmark (a tool, not product): does vector work of ANOTHER wave on the same SIMD proceed while a wave streams MFMAs on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_valu tools/micro/mfma_valu.hip && gpurun -- tools/micro/mfma_valu
// 8 waves per workgroup (2 per SIMD), one workgroup per CU.  Waves 0-3 stream v_mfma_f64_16x16x4_f64 or v_mfma_f32_32x32x16_bf16 (independent
// accumulators); waves 4-7 run `mode`: 0 exit at once, 1 int32 VALU adds, 2 f64 VALU fma, 3 LDS reads (ds_read_b128), 4 s_sleep loop, 5 v_mov only.
// Result (profiles/r06_experiments.md section 3): an fp64 MFMA stream starves every vector instruction of the other wave; a bf16 stream does not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int PM, int PY, int BF>
__global__ void __launch_bounds__(512, 2) k(int mode, int iters, int yiters, unsigned long long* out, double* sink, int swap) {
  __shared__ double lds[8192];
  const int wid0 = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wid = swap ? (wid0 ^ 4) : wid0;
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i;
  __syncthreads();
  if (wid < 4) {
    __builtin_amdgcn_s_setprio(PM);
    unsigned long long t0, t1;
    if (BF == 0) {
      d4 acc[16];
      for (int j = 0; j < 16; j++) acc[j] = (d4){0, 0, 0, 0};
      double a = lane * 0.5, b = lane * 0.25;
      t0 = __builtin_amdgcn_s_memtime();
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int j = 0; j < 16; j++) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
      }
      t1 = __builtin_amdgcn_s_memtime();
      double s = 0;
      for (int j = 0; j < 16; j++) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
      if (s == 123.456) sink[0] = s;
    } else {
      f32x16 acc[8];
      for (int j = 0; j < 8; j++) for (int e = 0; e < 16; e++) acc[j][e] = 0;
      bf16x8 a, b;
      for (int e = 0; e < 8; e++) { a[e] = (__bf16)(lane * 0.5f); b[e] = (__bf16)(lane * 0.25f); }
      t0 = __builtin_amdgcn_s_memtime();
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
          for (int j = 0; j < 8; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
      }
      t1 = __builtin_amdgcn_s_memtime();
      float s = 0;
      for (int j = 0; j < 8; j++) for (int e = 0; e < 16; e++) s += acc[j][e];
      if (s == 123.456f) sink[0] = s;
    }
    if (lane == 0 && blockIdx.x == 5) out[wid] = t1 - t0;
  } else {
    __builtin_amdgcn_s_setprio(PY);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 1) {
      unsigned x = lane, y = lane * 3;
      for (int it = 0; it < yiters; it++) {
#pragma unroll
        for (int r = 0; r < 32; r++) { x = x + y; y = y ^ x; }
      }
      if (x == 0x12345) sink[1] = x;
    } else if (mode == 2) {
      double x = lane, y = 1.0000001;
      for (int it = 0; it < yiters; it++) {
#pragma unroll
        for (int r = 0; r < 64; r++) x = __builtin_fma(x, y, 1e-9);
      }
      if (x == 0.12345) sink[1] = x;
    } else if (mode == 3) {
      double s = 0;
      for (int it = 0; it < yiters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) { const d4 v = *reinterpret_cast<const d4*>(&lds[((lane * 4 + r * 256 + it * 4) & 8188)]); s += v[0]; }
      }
      if (s == 0.12345) sink[1] = s;
    } else if (mode == 4) {
      for (int it = 0; it < yiters; it++) __builtin_amdgcn_s_sleep(8);
    } else if (mode == 5) {
      float x = lane;
      for (int it = 0; it < yiters; it++) {
#pragma unroll
        for (int r = 0; r < 64; r++) asm volatile("v_mov_b32 %0, %0" : "+v"(x));
      }
      if (x == 0.12345f) sink[1] = x;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 5) out[wid] = t1 - t0;
  }
}
int main(int argc, char** argv) {
  unsigned long long* out; double* sink;
  hipMalloc(&out, 64); hipMalloc(&sink, 64);
  const int iters = 200;      // 200 x 64 MFMAs = 12800 MFMAs = 819200 pipe cycles
  for (int bf = 0; bf < 2; bf++)
  for (int pr = 0; pr < 2; pr++)
  for (int mode : {0, 1, 3})
    for (int yi : {0, 50, 1000}) {
      if (mode == 0 && yi) continue;
      if (mode && !yi) continue;
      hipMemset(out, 0, 64);
      if (bf == 0 && pr == 0) hipLaunchKernelGGL((k<0, 0, 0>), dim3(256), dim3(512), 0, 0, mode, iters, yi, out, sink, 0);
      if (bf == 0 && pr == 1) hipLaunchKernelGGL((k<0, 3, 0>), dim3(256), dim3(512), 0, 0, mode, iters, yi, out, sink, 0);
      if (bf == 1 && pr == 0) hipLaunchKernelGGL((k<0, 0, 1>), dim3(256), dim3(512), 0, 0, mode, iters, yi, out, sink, 0);
      if (bf == 1 && pr == 1) hipLaunchKernelGGL((k<0, 3, 1>), dim3(256), dim3(512), 0, 0, mode, iters, yi, out, sink, 0);
      hipDeviceSynchronize();
      unsigned long long h[8];
      hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
      printf("%s prio(mfma,other)=%s mode %d yiters %5d : MFMA wave %8llu cycles (%.1f per MFMA) | other wave %8llu cycles\n", bf ? "bf16 32x32x16" : "f64 16x16x4", pr == 0 ? "0,0" : "0,3", mode, yi, h[0], (double)h[0] / (iters * 64.0), h[4]);
    }
  return 0;
}
