// Precision-conversion element kernels shared by the single-GPU and the distributed mixed-precision factorizations
// (mixed.hip, dist_mixed.hip): HBM-bound, coalesced along the column-major fast axis.
#pragma once
#include <algorithm>

#include "common.h"

namespace {
// The element-wise kernels below move four consecutive rows per lane (16-byte accesses) when sizes, leading dimensions and base
// addresses allow it - one element per lane and 256 elements per workgroup left them at 1 - 2.5 TB/s (round 4: the import of
// N = 65536 took 10.2 ms of the 170 ms factorization) - and fall back to the scalar loop otherwise.  grid2 sizes the launch for the
// vector form (1024 rows per workgroup pass); the scalar loop just takes more passes.
static __device__ __forceinline__ bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static __device__ __forceinline__ bool al8(const void* p) { return ((uintptr_t)p & 7) == 0; }
struct __attribute__((aligned(8))) bf16x4 { __bf16 v[4]; };

// fp32 import of A's upper triangle.  Only rows <= col are touched: the strictly-lower part of the factor is zeroed once per plan and
// never written afterwards (every update and solve is masked to the upper triangle), so the import moves half the matrix
static __global__ void f64_to_f32_upper_kernel(const double* A, int64_t lda, float* R, int64_t ldr, int64_t n) {
  const int64_t col = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (col >= n) return;
  if ((lda % 2 == 0) && (ldr % 4 == 0) && al16(A) && al16(R)) {
    const double* a = A + col * lda; float* r = R + col * ldr;
    for (int64_t r4 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; r4 <= col; r4 += (int64_t)gridDim.x * blockDim.x * 4) {
      if (r4 + 3 <= col) {
        const double2 lo = *reinterpret_cast<const double2*>(a + r4), hi = *reinterpret_cast<const double2*>(a + r4 + 2);
        *reinterpret_cast<float4*>(r + r4) = make_float4((float)lo.x, (float)lo.y, (float)hi.x, (float)hi.y);
      } else {
        for (int64_t row = r4; row <= col; row++) r[row] = (float)a[row];
      }
    }
    return;
  }
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row <= col; row += (int64_t)gridDim.x * blockDim.x)
    R[row + col * ldr] = (float)A[row + col * lda];
}
static __global__ void f32_to_f64_kernel(const float* S, int64_t lds_, double* D, int64_t ldd, int64_t rows, int64_t cols, int upper_only) {
  const int64_t col = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (col >= cols) return;
  if ((rows % 4 == 0) && (lds_ % 4 == 0) && (ldd % 2 == 0) && al16(S) && al16(D)) {
    const float* sc = S + col * lds_; double* dc = D + col * ldd;
    for (int64_t r4 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; r4 < rows; r4 += (int64_t)gridDim.x * blockDim.x * 4) {
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (!upper_only || r4 <= col) v = *reinterpret_cast<const float4*>(sc + r4);
      if (upper_only) { if (r4 > col) v.x = 0.0f; if (r4 + 1 > col) v.y = 0.0f; if (r4 + 2 > col) v.z = 0.0f; if (r4 + 3 > col) v.w = 0.0f; }
      *reinterpret_cast<double2*>(dc + r4) = make_double2((double)v.x, (double)v.y);
      *reinterpret_cast<double2*>(dc + r4 + 2) = make_double2((double)v.z, (double)v.w);
    }
    return;
  }
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
  if ((rows % 4 == 0) && (lds_ % 4 == 0) && al16(S) && al8(B3)) {
    float* sc = S + col * lds_;
    for (int64_t r4 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; r4 < rows; r4 += (int64_t)gridDim.x * blockDim.x * 4) {
      const float4 v = *reinterpret_cast<const float4*>(sc + r4);
      bf16x4 hi, lo;
      bf16_split(v.x, hi.v[0], lo.v[0]); bf16_split(v.y, hi.v[1], lo.v[1]); bf16_split(v.z, hi.v[2], lo.v[2]); bf16_split(v.w, hi.v[3], lo.v[3]);
      *reinterpret_cast<bf16x4*>(out + r4) = hi; *reinterpret_cast<bf16x4*>(out + rows + r4) = lo; *reinterpret_cast<bf16x4*>(out + 2 * rows + r4) = hi;
      if (zero_src) *reinterpret_cast<float4*>(sc + r4) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    return;
  }
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
  if ((rows % 4 == 0) && (lds_ % 4 == 0) && (ldp % 4 == 0) && al16(S) && al8(P)) {
    const float* sc = S + col * lds_; __bf16* pc = P + col * ldp;
    for (int64_t r4 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; r4 < rows; r4 += (int64_t)gridDim.x * blockDim.x * 4) {
      const float4 v = *reinterpret_cast<const float4*>(sc + r4);
      bf16x4 o; o.v[0] = (__bf16)v.x; o.v[1] = (__bf16)v.y; o.v[2] = (__bf16)v.z; o.v[3] = (__bf16)v.w;
      *reinterpret_cast<bf16x4*>(pc + r4) = o;
    }
    return;
  }
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < rows; row += (int64_t)gridDim.x * blockDim.x)
    P[row + col * ldp] = (__bf16)S[row + col * lds_];
}
static __global__ void axpy_cols_kernel(double* X, int64_t ldx, const double* D, int64_t ldd, int64_t rows, int64_t cols) {
  const int64_t col = blockIdx.y;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < rows; row += (int64_t)gridDim.x * blockDim.x)
    X[row + col * ldx] += D[row + col * ldd];
}

static inline dim3 grid2(int64_t rows, int64_t cols) {
  return dim3((unsigned)std::min<int64_t>(cap_ceil_div(rows, 1024), 4096), (unsigned)std::min<int64_t>(cols, 65535), (unsigned)cap_ceil_div(cols, 65535));
}

// launchers: the access notes of the schedule checker (common.h) next to the launch they describe
static inline void launch_f64_to_f32_upper(hipStream_t s, const double* A, int64_t lda, float* R, int64_t ldr, int64_t n) {
  cap_acc_r(A, lda, n, n, 1); cap_acc_w(R, ldr, n, n, 1, 4);
  hipLaunchKernelGGL(f64_to_f32_upper_kernel, grid2(n, n), dim3(256), 0, s, A, lda, R, ldr, n);
}
static inline void launch_f32_to_f64(hipStream_t s, const float* S, int64_t lds_, double* D, int64_t ldd, int64_t rows, int64_t cols, int upper_only) {
  cap_acc_r(S, lds_, rows, cols, upper_only ? 1 : 0, 4); cap_acc_w(D, ldd, rows, cols);
  hipLaunchKernelGGL(f32_to_f64_kernel, grid2(rows, cols), dim3(256), 0, s, S, lds_, D, ldd, rows, cols, upper_only);
}
static inline void launch_f64_to_f32_bf16(hipStream_t s, const double* S, int64_t lds_, float* R, int64_t ldr, __bf16* P, int64_t ldp, int64_t rows,
                                          int64_t cols, int upper_only) {
  cap_acc_r(S, lds_, rows, cols, upper_only ? 1 : 0); cap_acc_w(R, ldr, rows, cols, 0, 4);
  if (P) cap_acc_w(P, ldp, rows, cols, 0, 2);
  hipLaunchKernelGGL(f64_to_f32_bf16_kernel, grid2(rows, cols), dim3(256), 0, s, S, lds_, R, ldr, P, ldp, rows, cols, upper_only);
}
static inline void launch_split3_row(hipStream_t s, float* S, int64_t lds_, __bf16* B3, int64_t rows, int64_t cols, int zero_src) {
  cap_acc(zero_src ? CAP_ACC_RW : CAP_ACC_R, S, lds_, rows, cols, 0, 4); cap_acc_w(B3, 3 * rows, 3 * rows, cols, 0, 2);
  hipLaunchKernelGGL(split3_row_kernel, grid2(rows, cols), dim3(256), 0, s, S, lds_, B3, rows, cols, zero_src);
}
static inline void launch_split3_tri(hipStream_t s, const double* D, int64_t ldd, __bf16* A3, int64_t n) {
  cap_acc_r(D, ldd, n, n, 1); cap_acc_w(A3, 3 * n, 3 * n, n, 0, 2);
  hipLaunchKernelGGL(split3_tri_kernel, grid2(n, n), dim3(256), 0, s, D, ldd, A3, n);
}
static inline void launch_f32_to_bf16(hipStream_t s, const float* S, int64_t lds_, __bf16* P, int64_t ldp, int64_t rows, int64_t cols) {
  cap_acc_r(S, lds_, rows, cols, 0, 4); cap_acc_w(P, ldp, rows, cols, 0, 2);
  hipLaunchKernelGGL(f32_to_bf16_kernel, grid2(rows, cols), dim3(256), 0, s, S, lds_, P, ldp, rows, cols);
}
}  // namespace
