// Wavefront-cooperative base case: Cholesky factor AND triangular inverse of one small
// diagonal block, entirely inside one workgroup's LDS.
//
// Replaces the reference's base case (cholinv.hpp:168-183 -> policy.h:199-201:
// `_potrf`; memcpy; `_trtri` on the gathered bcN x bcN block, i.e. LAPACKE_dpotrf +
// LAPACKE_dtrtri, lapack/interface.hpp:30-58).  Like the reference it produces both R
// (A = R^T R, upper) and R^-1.
//
// Algorithm: the reference's own recursion (cholinv.hpp:85-165) applied inside LDS:
//   cholinv(n): n == 16 -> unblocked potrf + back-substitution inverse
//               else h = n/2: cholinv(A11); R12 = Ri11^T A12; A22 -= R12^T R12;
//                             cholinv(A22); Ri12 = -Ri11 R12 Ri22
// All inner products run on v_mfma_f64_16x16x4_f64 with operands read straight from LDS;
// the 4 waves of the workgroup split the 16x16 output blocks.  n <= 64 is padded to
// 16/32/64 with an identity tail so any size works (the reference handles ragged sizes
// with its `span` trick, policy.h:196).
#include "common.h"

namespace {

constexpr int LMAX = 64;       // largest leaf
constexpr int LLD = LMAX + 2;  // LDS leading dimension (doubles) - breaks the 64-dword bank period
constexpr int LTHREADS = 256;

// element (i,j) of a column-major LDS matrix
#define SM(P, i, j) (P)[(j) * LLD + (i)]

// C[m x n] = alpha * op(A) * op(B) (+ C if ACC), all LDS-resident column-major blocks.
// op(A)[i][k] = TA ? A[k][i] : A[i][k];  m, n multiples of 16, k multiple of 4.
// UPPER: only blocks with bi <= bj are computed (C is the window's diagonal-aligned square).
template <bool TA, bool ACC, bool UPPER>
__device__ __forceinline__ void lds_mm(double* C, const double* A, const double* B, int m, int n, int k, double alpha) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lr = lane & 15, kg = lane >> 4;
  const int mb = m >> 4, nb = n >> 4;
  for (int blk = wid; blk < mb * nb; blk += LTHREADS / 64) {
    const int bi = blk % mb, bj = blk / mb;
    if (UPPER && bi > bj) continue;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int kk = 0; kk < k; kk += 4) {
      // MFMA A operand: lane -> op(A)[bi*16 + lr][kk + kg];  B operand: B[kk + kg][bj*16 + lr]
      double a = TA ? SM(A, kk + kg, bi * 16 + lr) : SM(A, bi * 16 + lr, kk + kg);
      double b = SM(B, kk + kg, bj * 16 + lr);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    // D layout: row = kg + 4r, col = lr
#pragma unroll
    for (int r = 0; r < 4; r++) {
      double* p = &SM(C, bi * 16 + kg + 4 * r, bj * 16 + lr);
      double v = alpha * acc[r];
      if (ACC) v += *p;
      *p = v;
    }
  }
}

// 16 x 16 base: unblocked right-looking potrf on S[off.., off..] (upper), then inverse into T.
// One reciprocal per pivot (1/r), multiplications everywhere else; the reciprocals are kept in `dinv`
// for the back substitution, whose dependent chain then has no division.
__device__ __forceinline__ void base16(double* S, double* T, double* dinv, int off, int* bad) {
  const int t = threadIdx.x;
  const int i = t & 15, j = t >> 4;  // one (i,j) pair per thread
  for (int k = 0; k < 16; k++) {
    __syncthreads();
    double d = SM(S, off + k, off + k);
    if (!(d > 0.0) && t == 0 && *bad == 0) *bad = off + k + 1;
    double r = __builtin_sqrt(d);
    double rinv = 1.0 / r;
    double rowi = SM(S, off + k, off + i) * rinv, rowj = SM(S, off + k, off + j) * rinv;
    __syncthreads();
    if (i == k && j >= k) SM(S, off + k, off + j) = (j == k) ? r : rowj;
    if (i > k && j >= i) SM(S, off + i, off + j) -= rowi * rowj;
    if (t == 0) dinv[off + k] = rinv;
  }
  __syncthreads();
  // inverse: thread c solves R x = e_c by back substitution (column c of R^-1)
  if (t < 16) {
    const int c = t;
    double x[16];
#pragma unroll
    for (int q = 0; q < 16; q++) x[q] = 0.0;
#pragma unroll
    for (int ii = 15; ii >= 0; ii--) {
      double s = (ii == c) ? 1.0 : 0.0;
#pragma unroll
      for (int l = ii + 1; l < 16; l++) s -= SM(S, off + ii, off + l) * x[l];
      double v = s * dinv[off + ii];
      x[ii] = (ii <= c) ? v : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 16; q++) SM(T, off + q, off + c) = x[q];
  }
  __syncthreads();
}

template <int N>
__device__ void cholinv_lds(double* S, double* T, double* dinv, int off, int* bad) {
  if constexpr (N == 16) {
    base16(S, T, dinv, off, bad);
  } else {
    constexpr int H = N / 2;
    cholinv_lds<H>(S, T, dinv, off, bad);
    double* S11 = &SM(S, off, off);        (void)S11;
    double* S12 = &SM(S, off, off + H);
    double* S21 = &SM(S, off + H, off);     // scratch: the unused lower block
    double* S22 = &SM(S, off + H, off + H);
    double* T11 = &SM(T, off, off);
    double* T12 = &SM(T, off, off + H);
    double* T22 = &SM(T, off + H, off + H);
    // R12 = Ri11^T * A12   (TRMM Left/Upper/Trans of cholinv.hpp:118-121)
    lds_mm<true, false, false>(S21, T11, S12, H, H, H, 1.0);
    __syncthreads();
    for (int e = threadIdx.x; e < H * H; e += LTHREADS) SM(S12, e % H, e / H) = SM(S21, e % H, e / H);
    __syncthreads();
    // A22 -= R12^T R12     (SYRK Upper/Trans alpha=-1 beta=1, cholinv.hpp:128-137)
    lds_mm<true, true, true>(S22, S12, S12, H, H, H, -1.0);
    __syncthreads();
    cholinv_lds<H>(S, T, dinv, off + H, bad);
    // Ri12 = -Ri11 * R12 * Ri22   (two TRMMs, cholinv.hpp:150-154)
    lds_mm<false, false, false>(S21, S12, T22, H, H, H, 1.0);
    __syncthreads();
    lds_mm<false, false, false>(T12, T11, S21, H, H, H, -1.0);
    __syncthreads();
  }
}

// A: n x n block (column-major, lda), upper triangle consumed and overwritten by R (lower
// untouched).  Rinv (may be NULL): n x n, ldr; upper triangle written, strictly lower part
// zero-filled when zero_lower != 0.  info: device int, set to info_base + (1-based pivot
// index) on the first non-positive pivot if currently 0.
__global__ void __launch_bounds__(LTHREADS) leaf_cholinv_kernel(double* A, int64_t lda, double* Rinv, int64_t ldr,
                                                                int n, int zero_lower, int* info, int info_base) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* S = lds;
  double* T = lds + LMAX * LLD;
  int& bad = *reinterpret_cast<int*>(lds + 2 * LMAX * LLD);   // keep ALL LDS in the dynamic region (16-B aligned base)
  double* dinv = lds + 2 * LMAX * LLD + 2;                    // 1/R[k][k], 64 doubles
  const int t = threadIdx.x;
  const int np = n <= 16 ? 16 : (n <= 32 ? 32 : 64);
  __builtin_amdgcn_s_setprio(3);      // latency-critical single workgroup: win issue arbitration against co-resident bulk waves
  if (t == 0) bad = 0;
  for (int e = t; e < np * np; e += LTHREADS) {
    int i = e % np, j = e / np;
    double v = 0.0;
    if (i < n && j < n) { if (i <= j) v = A[i + (int64_t)j * lda]; }
    else if (i == j) v = 1.0;
    SM(S, i, j) = v;
    SM(T, i, j) = 0.0;
  }
  __syncthreads();
  if (np == 16) cholinv_lds<16>(S, T, dinv, 0, &bad);
  else if (np == 32) cholinv_lds<32>(S, T, dinv, 0, &bad);
  else cholinv_lds<64>(S, T, dinv, 0, &bad);
  __syncthreads();
  for (int e = t; e < n * n; e += LTHREADS) {
    int i = e % n, j = e / n;
    if (i <= j) {
      A[i + (int64_t)j * lda] = SM(S, i, j);
      if (Rinv) Rinv[i + (int64_t)j * ldr] = SM(T, i, j);
    } else if (Rinv && zero_lower) {
      Rinv[i + (int64_t)j * ldr] = 0.0;
    }
  }
  if (t == 0 && bad != 0 && bad <= n && info) atomicCAS(info, 0, info_base + bad);
}

// In-LDS inverse only (upper triangular n <= 64): Rinv = R^-1 without factoring.
__global__ void __launch_bounds__(LTHREADS) leaf_trtri_kernel(const double* R, int64_t ldr, double* Rinv, int64_t ldi, int n) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* S = lds;
  double* T = lds + LMAX * LLD;
  const int t = threadIdx.x;
  const int np = n <= 16 ? 16 : (n <= 32 ? 32 : 64);
  for (int e = t; e < np * np; e += LTHREADS) {
    int i = e % np, j = e / np;
    double v = 0.0;
    if (i < n && j < n) { if (i <= j) v = R[i + (int64_t)j * ldr]; }
    else if (i == j) v = 1.0;
    SM(S, i, j) = v;
    SM(T, i, j) = 0.0;
  }
  __syncthreads();
  // invert the 16x16 diagonal blocks (thread c of group g solves column c of block g)
  const int nblk = np / 16;
  if (t < 16 * nblk) {
    const int off = (t >> 4) * 16, c = t & 15;
    double x[16], rd[16];
#pragma unroll
    for (int q = 0; q < 16; q++) { x[q] = 0.0; rd[q] = 1.0 / SM(S, off + q, off + q); }   // independent divisions, off the chain
#pragma unroll
    for (int ii = 15; ii >= 0; ii--) {
      double s = (ii == c) ? 1.0 : 0.0;
#pragma unroll
      for (int l = ii + 1; l < 16; l++) s -= SM(S, off + ii, off + l) * x[l];
      double v = s * rd[ii];
      x[ii] = (ii <= c) ? v : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 16; q++) SM(T, off + q, off + c) = x[q];
  }
  __syncthreads();
  // combine: for block size h = 16, 32: Ri12 = -Ri11 R12 Ri22 for every aligned pair
  for (int h = 16; h < np; h *= 2) {
    for (int off = 0; off < np; off += 2 * h) {
      lds_mm<false, false, false>(&SM(S, off + h, off), &SM(S, off, off + h), &SM(T, off + h, off + h), h, h, h, 1.0);
    }
    __syncthreads();
    for (int off = 0; off < np; off += 2 * h) {
      lds_mm<false, false, false>(&SM(T, off, off + h), &SM(T, off, off), &SM(S, off + h, off), h, h, h, -1.0);
    }
    __syncthreads();
  }
  for (int e = t; e < n * n; e += LTHREADS) {
    int i = e % n, j = e / n;
    if (i <= j) Rinv[i + (int64_t)j * ldi] = SM(T, i, j);
  }
}

}  // namespace

constexpr size_t LEAF_LDS_BYTES = (2 * LMAX * LLD + 2 + LMAX) * sizeof(double);

int cap_leaf_cholinv(double* A, int64_t lda, double* Rinv, int64_t ldr, int n, int zero_lower, int* info,
                     int info_base, hipStream_t stream) {
  if (n <= 0) return CAP_OK;
  if (n > LMAX) return CAP_ERR_ARG;
  hipLaunchKernelGGL(leaf_cholinv_kernel, dim3(1), dim3(LTHREADS), LEAF_LDS_BYTES, stream, A, lda, Rinv, ldr, n,
                     zero_lower, info, info_base);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

int cap_leaf_trtri(const double* R, int64_t ldr, double* Rinv, int64_t ldi, int n, hipStream_t stream) {
  if (n <= 0) return CAP_OK;
  if (n > LMAX) return CAP_ERR_ARG;
  hipLaunchKernelGGL(leaf_trtri_kernel, dim3(1), dim3(LTHREADS), LEAF_LDS_BYTES, stream, R, ldr, Rinv, ldi, n);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}
