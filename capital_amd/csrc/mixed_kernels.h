// Precision-conversion element kernels shared by the single-GPU and the distributed mixed-precision factorizations
// (mixed.hip, dist_mixed.hip): HBM-bound, coalesced along the column-major fast axis.
#pragma once
#include <algorithm>

#include "common.h"

namespace {
// fp32 import of A's upper triangle.  Only rows <= col are touched: the strictly-lower part of the factor is zeroed once per plan and
// never written afterwards (every update and solve is masked to the upper triangle), so the import moves half the matrix
static __global__ void f64_to_f32_upper_kernel(const double* A, int64_t lda, float* R, int64_t ldr, int64_t n) {
  const int64_t col = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (col >= n) return;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row <= col; row += (int64_t)gridDim.x * blockDim.x)
    R[row + col * ldr] = (float)A[row + col * lda];
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
// ---- block-row solve on the bf16 matrix pipe with fp32-class accuracy ("bf16 x 3"): x ~ hi + lo with hi = bf16(x), lo = bf16(x - hi)
// (relative 2^-16), and  A^T B ~ A_hi^T B_hi + A_hi^T B_lo + A_lo^T B_hi  is ONE bf16 product over a three times longer K:
//   A3 = [A_hi; A_hi; A_lo],  B3 = [B_hi; B_lo; B_hi]   (K-contiguous, leading dimension 3 K)
// the solved row S = Dinv^T Row then carries ~2e-5 relative error where the factor it goes into is a bf16-update factor (1e-3).
static __device__ __forceinline__ void bf16_split(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
}
// B3 from the fp32 block row S (rows x cols, lds_); zero_src: the row is cleared behind the read (it becomes the atomic accumulator)
static __global__ void split3_row_kernel(float* S, int64_t lds_, __bf16* B3, int64_t rows, int64_t cols, int zero_src) {
  const int64_t col = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (col >= cols) return;
  __bf16* out = B3 + col * 3 * rows;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < rows; row += (int64_t)gridDim.x * blockDim.x) {
    __bf16 hi, lo;
    bf16_split(S[row + col * lds_], hi, lo);
    out[row] = hi; out[rows + row] = lo; out[2 * rows + row] = hi;
    if (zero_src) S[row + col * lds_] = 0.0f;
  }
}
// A3 from the fp64 upper-triangular inverse D (n x n, ldd): column i of A3 = [hi; hi; lo] of column i of D (zero below the diagonal)
static __global__ void split3_tri_kernel(const double* D, int64_t ldd, __bf16* A3, int64_t n) {
  const int64_t col = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (col >= n) return;
  __bf16* out = A3 + col * 3 * n;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < n; row += (int64_t)gridDim.x * blockDim.x) {
    __bf16 hi, lo;
    bf16_split(row <= col ? (float)D[row + col * ldd] : 0.0f, hi, lo);
    out[row] = hi; out[n + row] = hi; out[2 * n + row] = lo;
  }
}
static __global__ void f32_to_bf16_kernel(const float* S, int64_t lds_, __bf16* P, int64_t ldp, int64_t rows, int64_t cols) {
  const int64_t col = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (col >= cols) return;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < rows; row += (int64_t)gridDim.x * blockDim.x)
    P[row + col * ldp] = (__bf16)S[row + col * lds_];
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
