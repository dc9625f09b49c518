// Precision-conversion element kernels shared by the single-GPU and the distributed mixed-precision factorizations
// (mixed.hip, dist_mixed.hip): HBM-bound, coalesced along the column-major fast axis.
#pragma once
#include <algorithm>

#include "common.h"

namespace {
static __global__ void f64_to_f32_upper_kernel(const double* A, int64_t lda, float* R, int64_t ldr, int64_t n) {
  const int64_t col = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (col >= n) return;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < n; row += (int64_t)gridDim.x * blockDim.x)
    R[row + col * ldr] = row <= col ? (float)A[row + col * lda] : 0.0f;
}
static __global__ void f32_to_f64_kernel(const float* S, int64_t lds_, double* D, int64_t ldd, int64_t rows, int64_t cols, int upper_only) {
  const int64_t col = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (col >= cols) return;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < rows; row += (int64_t)gridDim.x * blockDim.x)
    D[row + col * ldd] = (!upper_only || row <= col) ? (double)S[row + col * lds_] : 0.0;
}
// the solved block row: fp64 -> fp32 (into the factor) and bf16 (into the K-contiguous panel of the trailing update)
static __global__ void f64_to_f32_bf16_kernel(const double* S, int64_t lds_, float* R, int64_t ldr, __bf16* P, int64_t ldp, int64_t rows, int64_t cols,
                                       int upper_only) {
  const int64_t col = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (col >= cols) return;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < rows; row += (int64_t)gridDim.x * blockDim.x) {
    const double v = (!upper_only || row <= col) ? S[row + col * lds_] : 0.0;
    R[row + col * ldr] = (float)v;
    if (P) P[row + col * ldp] = (__bf16)(float)v;
  }
}
static __global__ void axpy_cols_kernel(double* X, int64_t ldx, const double* D, int64_t ldd, int64_t rows, int64_t cols) {
  const int64_t col = blockIdx.y;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < rows; row += (int64_t)gridDim.x * blockDim.x)
    X[row + col * ldx] += D[row + col * ldd];
}

static inline dim3 grid2(int64_t rows, int64_t cols) {
  return dim3((unsigned)std::min<int64_t>(cap_ceil_div(rows, 256), 4096), (unsigned)std::min<int64_t>(cols, 65535), (unsigned)cap_ceil_div(cols, 65535));
}
}  // namespace
