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
#include "kargs.h"

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
__device__ __forceinline__ void lds_mm(double* C, const double* A, const double* B, int m, int n, int k, double alpha, int tid = threadIdx.x) {
  const int lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 15, kg = lane >> 4;
  const int mb = m >> 4, nb = n >> 4;
  for (int blk = wid; blk < mb * nb; blk += LTHREADS / 64) {
    const int bi = blk % mb, bj = blk / mb;
    if (UPPER && bi > bj) continue;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    int kk = 0;
    for (; kk + 16 <= k; kk += 16) {     // four k steps per trip: eight LDS reads in flight, then the MFMAs (same summation order)
      double a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        a[u] = TA ? SM(A, kk + 4 * u + kg, bi * 16 + lr) : SM(A, bi * 16 + lr, kk + 4 * u + kg);
        b[u] = SM(B, kk + 4 * u + kg, bj * 16 + lr);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
    for (; kk < k; kk += 4) {
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

// 1/sqrt(d) from v_rsq_f64 + two Newton steps (full fp64 accuracy, ~20 dependent ops instead of the ~80 of
// sqrt followed by a division); d <= 0 yields NaN/Inf which the caller reports through `bad`.
__device__ __forceinline__ double fast_rsqrt(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  y = y * (1.5 - h * y * y);
  y = y * (1.5 - h * y * y);
  return y;
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// An opaque zero: added to an LDS pointer it keeps the compiler from turning every uniform LDS address of an unrolled loop into
// its own constant (hoisted into SGPRs, spilled to VGPR lanes and fetched back with v_readlane + v_mov in front of every ds_read -
// what the row solve below looked like in round 3); with it the accesses are "one VGPR + immediate offset".
__device__ __forceinline__ int opaque_zero() { int z = 0; asm volatile("" : "+v"(z)); return z; }

// 16 x 16 Cholesky of S[off.., off..] by ONE wave, block held in registers: lane j (< 16; the other lanes mirror it and store
// nothing) owns column j.  Per pivot k the chain that the next pivot waits for is short: the pivot itself and R[k][k+1] come out of
// their owners' registers (v_readlane, uniform lane numbers after unrolling) and only row k + 1 is updated at once; the finished row
// k goes to LDS (rt, row-major: also what the row-panel solve reads) and the remaining rows take their update from there one pivot
// later (same products in the same order as the plain right-looking loop).  ~ 180 cycles per pivot instead of 360 with an LDS row
// exchange in every pivot's chain.  Leaves 1/R[k][k] in dinv.
__device__ __forceinline__ void potrf16_wave(double* S, double* dinv, double* rt, int off, int* bad, int tid = threadIdx.x) {
  const int lane = tid & 63;
  const int j = lane & 15;
  const int z = opaque_zero();
  double* Sj = S + z + (off + j) * LLD + off;              // column j of the block
  double* rtz = rt + z;
  double a[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = Sj[i];
  double prev = 0.0;                                       // R[k-1][j]
#pragma unroll
  for (int k = 0; k < 16; k++) {
    // (row k - 1 is fetched first: its LDS latency hides behind the reciprocal square root; the scheduling fences keep the compiler
    // from sinking the reads to their uses, where every multiply-add would wait for its own read)
    double rv[16];
    if (k >= 1) {
#pragma unroll
      for (int i = k + 1; i < 16; i++) rv[i] = rtz[(k - 1) * 16 + i];
    }
    __builtin_amdgcn_sched_barrier(0);
    const double d = readlane_f64(a[k], k);
    if (!(d > 0.0) && lane == 0 && *bad == 0) *bad = off + k + 1;
    const double rinv = fast_rsqrt(d);
    const double rjs = a[k] * rinv;
    const double fin = (j == k) ? d * rinv : (j > k ? rjs : a[k]);
    if (lane < 16) rtz[k * 16 + j] = fin;
    if (k >= 1) {                                          // rows k + 1 .. 15 take the update of pivot k - 1 (in the shadow of the chain above)
#pragma unroll
      for (int i = k + 1; i < 16; i++)
        if (j >= i) a[i] -= rv[i] * prev;
    }
    if (k < 15) {                                          // row k + 1 takes the update of pivot k now
      const double r1 = readlane_f64(rjs, k + 1);
      if (j >= k + 1) a[k + 1] -= r1 * rjs;
    }
    a[k] = fin; prev = rjs;
    if (lane == 0) dinv[off + k] = rinv;
  }
  if (lane < 16) {
#pragma unroll
    for (int i = 0; i < 16; i++) Sj[i] = a[i];
  }
}

// Right-looking blocked Cholesky of the np x np (np = 16, 32 or 64) block in S, 16-wide panels:
//   potrf16 (wave 0, registers) | row panel by forward substitution (one lane per column, R11 broadcast from LDS)
//   | rank-16 update of the trailing blocks on MFMA.
__device__ __forceinline__ void potrf_lds(double* S, double* dinv, double* rt, int np, int* bad, int tid = threadIdx.x,
                                          long long* stamps = nullptr) {
  const int t = tid;
  for (int off = 0; off < np; off += 16) {
    if (t < 64) potrf16_wave(S, dinv, rt, off, bad, tid);
    __syncthreads();
    if (stamps) stamps[(off >> 4) * 3 + 0] = (long long)__builtin_amdgcn_s_memrealtime();
    const int rest = np - off - 16;
    if (rest > 0) {
      if (t < rest) {   // X = R11^-T * A12, column t of the panel: right-looking, so that the 16-long dependent chain is one multiply and
        // one multiply-add per step; R11's rows come contiguous out of rt (same products in the same order as the dot-product form)
        const int c = off + 16 + t;
        const int z = opaque_zero();
        double* Sc = S + z + c * LLD + off;                 // column c, rows off ..
        const double* rtz = rt + z;
        const double* dv = dinv + z + off;
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = Sc[i];
        double dl[16], cur[16], nxt[16];                    // reciprocals; row l of R11 (columns > l); row l + 1 on its way
#pragma unroll
        for (int i = 0; i < 16; i++) dl[i] = dv[i];
#pragma unroll
        for (int i = 1; i < 16; i++) cur[i] = rtz[i];
#pragma unroll
        for (int l = 0; l < 16; l++) {
          if (l + 1 < 16) {
#pragma unroll
            for (int i = l + 2; i < 16; i++) nxt[i] = rtz[(l + 1) * 16 + i];
          }
          __builtin_amdgcn_sched_barrier(0);                 // (the reads stay up here: a row of multiply-adds hides their latency)
          x[l] = x[l] * dl[l];
#pragma unroll
          for (int i = l + 1; i < 16; i++) x[i] -= cur[i] * x[l];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = l + 2; i < 16; i++) cur[i] = nxt[i];
        }
#pragma unroll
        for (int i = 0; i < 16; i++) Sc[i] = x[i];
      }
      __syncthreads();
      if (stamps) stamps[(off >> 4) * 3 + 1] = (long long)__builtin_amdgcn_s_memrealtime();
      lds_mm<true, true, true>(&SM(S, off + 16, off + 16), &SM(S, off, off + 16), &SM(S, off, off + 16), rest, rest, 16, -1.0, tid);
      __syncthreads();
      if (stamps) stamps[(off >> 4) * 3 + 2] = (long long)__builtin_amdgcn_s_memrealtime();
    }
  }
}

// T = R^-1 for the np x np upper-triangular block in S (T zero on entry): the 16 x 16 diagonal blocks by back
// substitution (all blocks in parallel, reciprocals from dinv), then Ri12 = -Ri11 R12 Ri22 level by level on MFMA;
// S's strictly-lower blocks serve as scratch.
__device__ __forceinline__ void trtri_lds(double* S, double* T, const double* dinv, int np, int tid = threadIdx.x) {
  const int t = tid;
  const int nblk = np / 16;
  if (t < 16 * nblk) {
    const int off = (t >> 4) * 16, c = t & 15;
    // column c of the block's inverse by back substitution, right-looking: once x[l] is known every remaining row takes its term
    // (independent multiply-adds, R's column l contiguous in LDS) - the dependent chain is one multiply + one multiply-add per row
    const int z = opaque_zero();
    const double* Sb = S + z + off * LLD + off;             // the diagonal block: element (ii, l) at Sb[l * LLD + ii]
    const double* dv = dinv + z + off;
    double x[16];
#pragma unroll
    for (int q = 0; q < 16; q++) x[q] = (q == c) ? 1.0 : 0.0;
    double dl[16], cur[16], nxt[16];                        // reciprocals; column l of R11 (rows < l); column l - 1 on its way
#pragma unroll
    for (int q = 0; q < 16; q++) dl[q] = dv[q];
#pragma unroll
    for (int ii = 0; ii < 15; ii++) cur[ii] = Sb[15 * LLD + ii];
#pragma unroll
    for (int l = 15; l >= 0; l--) {
      if (l >= 1) {
#pragma unroll
        for (int ii = 0; ii < l - 1; ii++) nxt[ii] = Sb[(l - 1) * LLD + ii];
      }
      __builtin_amdgcn_sched_barrier(0);
      const double v = x[l] * dl[l];
      x[l] = (l <= c) ? v : 0.0;
#pragma unroll
      for (int ii = 0; ii < l; ii++) x[ii] -= cur[ii] * x[l];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ii = 0; ii < l - 1; ii++) cur[ii] = nxt[ii];
    }
#pragma unroll
    for (int q = 0; q < 16; q++) SM(T, off + q, off + c) = x[q];
  }
  __syncthreads();
  for (int h = 16; h < np; h *= 2) {
    for (int off = 0; off < np; off += 2 * h)
      lds_mm<false, false, false>(&SM(S, off + h, off), &SM(S, off, off + h), &SM(T, off + h, off + h), h, h, h, 1.0, tid);
    __syncthreads();
    for (int off = 0; off < np; off += 2 * h)
      lds_mm<false, false, false>(&SM(T, off, off + h), &SM(T, off, off), &SM(S, off + h, off), h, h, h, -1.0, tid);
    __syncthreads();
  }
}

// A: n x n block (column-major, lda), upper triangle consumed and overwritten by R (lower
// untouched).  Rinv (may be NULL): n x n, ldr; upper triangle written, strictly lower part
// zero-filled when zero_lower != 0.  info: device int, set to info_base + (1-based pivot
// index) on the first non-positive pivot if currently 0.
// Optional side job (cjob_cols > 0): copy the 64 x cjob_cols block row solved by the previous fused step from its
// scratch (ld 64) to its final place in R - issued first so the traffic overlaps the factorization below.
__global__ void __launch_bounds__(LTHREADS) leaf_cholinv_kernel(double* A, int64_t lda, double* Rinv, int64_t ldr,
                                                                int n, int zero_lower, int* info, int info_base,
                                                                const double* cjob_src, double* cjob_dst, int64_t cjob_ld,
                                                                int cjob_cols) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  for (int e = threadIdx.x; e < 64 * cjob_cols; e += LTHREADS) cjob_dst[(e & 63) + (int64_t)(e >> 6) * cjob_ld] = cjob_src[e];
  double* S = lds;
  double* T = lds + LMAX * LLD;
  int& bad = *reinterpret_cast<int*>(lds + 2 * LMAX * LLD);   // keep ALL LDS in the dynamic region (16-B aligned base)
  double* dinv = lds + 2 * LMAX * LLD + 2;                    // 1/R[k][k], 64 doubles
  double* rowbuf = dinv + LMAX;                               // the finished 16 x 16 diagonal block, row-major (256 doubles)
  const int t = threadIdx.x;
  const int np = n <= 16 ? 16 : (n <= 32 ? 32 : 64);
  __builtin_amdgcn_s_setprio(3);      // latency-critical single workgroup: win issue arbitration against co-resident bulk waves
  if (t == 0) bad = 0;
  for (int e0 = 0; e0 < np * np; e0 += 4 * LTHREADS) {   // 4 independent loads in flight per thread
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int e = e0 + u * LTHREADS + t;
      const int i = e % np, j = e / np;
      const bool in = (e < np * np) && i < n && j < n && i <= j;
      v[u] = in ? A[i + (int64_t)j * lda] : ((i == j && i >= n) ? 1.0 : 0.0);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int e = e0 + u * LTHREADS + t;
      if (e < np * np) { SM(S, e % np, e / np) = v[u]; SM(T, e % np, e / np) = 0.0; }
    }
  }
  __syncthreads();
  potrf_lds(S, dinv, rowbuf, np, &bad);
  // R goes out first (the scratch use of S's lower blocks below never touches the upper triangle)
  for (int e = t; e < n * n; e += LTHREADS) {
    int i = e % n, j = e / n;
    if (i <= j) A[i + (int64_t)j * lda] = SM(S, i, j);
  }
  if (Rinv) trtri_lds(S, T, dinv, np);
  __syncthreads();
  for (int e = t; e < n * n; e += LTHREADS) {
    int i = e % n, j = e / n;
    if (i <= j) {
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
  double* dinv = lds + 2 * LMAX * LLD + 2;
  if (t < np) dinv[t] = 1.0 / SM(S, t, t);
  __syncthreads();
  trtri_lds(S, T, dinv, np);
  for (int e = t; e < n * n; e += LTHREADS) {
    int i = e % n, j = e / n;
    if (i <= j) Rinv[i + (int64_t)j * ldi] = SM(T, i, j);
  }
}

// ---------------------------------------------------------------------------------------------
// Fused step of the 64-blocked right-looking factorization of one nb x nb diagonal block (nb = 64 * nblk):
// after leaf i produced R_ii and Dinv_i = R_ii^-1, ONE launch solves the block row i and applies its rank-64
// update to every trailing 64 x 64 block (a, b), i < a <= b:
//     X_a = Dinv_i^T A_ia,  X_b = Dinv_i^T A_ib   (recomputed by every workgroup that needs them - 2 x 64^3
//     flops of redundancy buy the removal of a grid-wide dependency, i.e. of one kernel launch per step)
//     A_ab -= X_a^T X_b                          (upper part only on diagonal blocks)
// and the workgroup of the diagonal block (a, a) also stores X_a as the final R_ia.
// Replaces 3-4 dependent launches of the recursive formulation per step (cholinv.hpp:113-137 at tile scale).
// ---------------------------------------------------------------------------------------------
// LDS budget: Dinv_i packed upper (16.6 KB) + two 64 x 64 panels (33.8 KB each) = 84 KB, so the workgroup fits
// into ONE slot vacated by a bulk-update workgroup (96 KB); a 4-buffer version (135 KB) needed a whole idle CU
// and lost 4 % end to end under a concurrent bulk update.  Products stay in registers between phases.
//
// fold (f.Dnext != nullptr): the workgroup of the NEXT diagonal block (a = b = i + 1) goes straight on to factor and invert
// the block it has just updated (the leaf of step i + 1: same LDS, the two panels become S and T), so a step is ONE launch
// instead of two - under a concurrent bulk update every launch of the chain waits for a workgroup slot to come free.
// The solved block row of the previous step (f.cj_*) is moved into R by the last workgroup of the launch, the solved
// pieces of this step go to the other half of the scratch (f.direct: a single-workgroup launch writes its piece in place).
// struct Panel64Fold: csrc/kargs.h (shared with the CPU kernel models of tests/hipshim)

__global__ void __launch_bounds__(LTHREADS) panel64_solve_update_kernel(double* R, int64_t ldr, const double* Dinv, int64_t ldi,
                                                                       int i, int nblk, double* Xs, const Panel64Fold f) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* B0 = lds;                       // A_ia -> X_a
  double* B1 = lds + LMAX * LLD;          // A_ib -> X_b
  double* Dp = lds + 2 * LMAX * LLD;      // Dinv_i, packed upper: (k, p), k <= p, at p(p+1)/2 + k
  __builtin_amdgcn_s_setprio(3);
  if (blockIdx.x == gridDim.x - 1)
    for (int e = threadIdx.x; e < 64 * f.cj_cols; e += LTHREADS) f.cj_dst[(e & 63) + (int64_t)(e >> 6) * f.cj_ld] = f.cj_src[e];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int lr = lane & 15, kg = lane >> 4;
  int q = blockIdx.x, bjt = 0;            // triangular decode -> (a, b), i < a <= b < nblk
  while (q >= bjt + 1) { q -= bjt + 1; bjt++; }
  const int a = i + 1 + q, b = i + 1 + bjt;
  const bool diag = (a == b);
  const double* Aia = R + (int64_t)i * 64 + (int64_t)a * 64 * ldr;
  const double* Aib = R + (int64_t)i * 64 + (int64_t)b * 64 * ldr;
  for (int e0 = 0; e0 < 64 * 64; e0 += 4 * LTHREADS) {
    double v0[4], v1[4], v2[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int e = e0 + u * LTHREADS + t, ii = e & 63, jj = e >> 6;
      v0[u] = (ii <= jj) ? Dinv[ii + (int64_t)jj * ldi] : 0.0;
      v1[u] = Aia[ii + (int64_t)jj * ldr];
      v2[u] = diag ? 0.0 : Aib[ii + (int64_t)jj * ldr];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int e = e0 + u * LTHREADS + t, ii = e & 63, jj = e >> 6;
      if (ii <= jj) Dp[jj * (jj + 1) / 2 + ii] = v0[u];
      SM(B0, ii, jj) = v1[u];
      if (!diag) SM(B1, ii, jj) = v2[u];
    }
  }
  __syncthreads();
  // X = Dinv^T * B for the wave's four 16 x 16 output blocks (block id = wid + 4 s: bi = id & 3, bj = id >> 2);
  // Dinv upper triangular -> rows k > p of column p vanish: only k < 16 (bi + 1) contributes to row block bi
  auto solve = [&](const double* B, d4 (&acc)[4]) {
#pragma unroll
    for (int sblk = 0; sblk < 4; sblk++) {
      const int id = wid + 4 * sblk, bi = id & 3, bj = id >> 2;
      d4 c = {0.0, 0.0, 0.0, 0.0};
      const int p = bi * 16 + lr;
      for (int k0 = 0; k0 < 16 * (bi + 1); k0 += 4) {
        const int k = k0 + kg;
        const double av = (k <= p) ? Dp[p * (p + 1) / 2 + k] : 0.0;     // Dinv[k][p]
        const double bv = SM(B, k, bj * 16 + lr);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0);    // rows p (= kg + 4r), cols lr
      }
      acc[sblk] = c;
    }
  };
  auto put = [&](double* B, const d4 (&acc)[4]) {
#pragma unroll
    for (int sblk = 0; sblk < 4; sblk++) {
      const int id = wid + 4 * sblk, bi = id & 3, bj = id >> 2;
#pragma unroll
      for (int r = 0; r < 4; r++) SM(B, bi * 16 + kg + 4 * r, bj * 16 + lr) = acc[sblk][r];
    }
  };
  d4 xa[4], xb[4];
  solve(B0, xa);
  if (!diag) solve(B1, xb);
  __syncthreads();                 // every read of the unsolved panels is done
  put(B0, xa);
  if (!diag) put(B1, xb);
  __syncthreads();
  if (diag) {
    // the solved block R_ia goes to scratch (other workgroups of this launch still read the unsolved A_ia from R);
    // the next leaf launch moves the whole block row into place
    if (f.direct) {
      double* dst = R + (int64_t)i * 64 + (int64_t)a * 64 * ldr;
      for (int e = t; e < 64 * 64; e += LTHREADS) dst[(e & 63) + (int64_t)(e >> 6) * ldr] = SM(B0, e & 63, e >> 6);
    } else {
      double* dst = Xs + (int64_t)(a - i - 1) * 64 * 64;
      for (int e = t; e < 64 * 64; e += LTHREADS) dst[e] = SM(B0, e & 63, e >> 6);
    }
  }
  // C_ab -= X_a^T X_b, computed transposed (MFMA rows <- columns of C) so that a lane owns 16 consecutive ROWS of C
  const double* XB = diag ? B0 : B1;
  double* C = R + (int64_t)a * 64 + (int64_t)b * 64 * ldr;
  const bool fold = f.Dnext != nullptr && blockIdx.x == 0;       // blockIdx.x == 0 <=> a == b == i + 1
  if (fold) {                                                     // B1 is idle in a diagonal workgroup: it becomes S
    for (int e = t; e < 64 * LLD; e += LTHREADS) B1[e] = 0.0;
    __syncthreads();
  }
#pragma unroll
  for (int sblk = 0; sblk < 4; sblk++) {
    const int id = wid + 4 * sblk, bi = id & 3, bj = id >> 2;     // bi: row block of C (from X_a), bj: column block (from X_b)
    if (diag && bi > bj) continue;
    d4 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k0 = 0; k0 < 64; k0 += 4) {
      const double colv = SM(XB, k0 + kg, bj * 16 + lr);    // MFMA "A": index lr -> column of C
      const double rowv = SM(B0, k0 + kg, bi * 16 + lr);    // MFMA "B": index lr -> row of C
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(colv, rowv, c, 0, 0, 0);   // result: (col = kg + 4r, row = lr)
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = bi * 16 + lr, col = bj * 16 + kg + 4 * r;
      if (!diag || row <= col) {
        if (fold) SM(B1, row, col) = C[row + (int64_t)col * ldr] - c[r];
        else C[row + (int64_t)col * ldr] -= c[r];
      }
    }
  }
  if (!fold) return;
  // ---- leaf of step i + 1 on the block just updated: S = B1, T = B0 (cf. leaf_cholinv_kernel)
  __syncthreads();                                               // every read of B0 (X_a) is done, S is complete
  double* S = B1; double* T = B0;
  int& bad = *reinterpret_cast<int*>(Dp);                        // Dinv_i's packed copy is dead after the solve
  double* dinv = Dp + 2; double* rowbuf = dinv + LMAX;
  for (int e = t; e < 64 * LLD; e += LTHREADS) T[e] = 0.0;
  if (t == 0) bad = 0;
  __syncthreads();
  potrf_lds(S, dinv, rowbuf, 64, &bad);
  for (int e = t; e < 64 * 64; e += LTHREADS) {
    const int ii = e & 63, jj = e >> 6;
    if (ii <= jj) C[ii + (int64_t)jj * ldr] = SM(S, ii, jj);
  }
  trtri_lds(S, T, dinv, 64);
  __syncthreads();
  for (int e = t; e < 64 * 64; e += LTHREADS) {
    const int ii = e & 63, jj = e >> 6;
    f.Dnext[ii + (int64_t)jj * f.ldn] = (ii <= jj) ? SM(T, ii, jj) : 0.0;
  }
  if (t == 0 && bad != 0 && f.info) atomicCAS(f.info, 0, f.info_base + bad);
}

// ---------------------------------------------------------------------------------------------
// One launch per level of the triangular-inverse assembly (blocked_cholinv, cholinv.hip): for every aligned pair z of
// h x h diagonal blocks (h = 64 RBW)      Ri12_z = -Ri11_z (R12_z Ri22_z).
// Workgroup (s, z) owns the 16 columns s of the pair's off-diagonal block: W = R12 Ri22[:, s] (h x 16) stays in LDS between
// the two products, so a level is one dependent launch instead of two.  Operands come straight from L2 (they were written
// by the previous launches of the chain); the triangular structure cuts both K ranges (Ri22 upper: k < 16 (s + 1) rows,
// Ri11 upper: k >= 16 rb for row block rb).  Wave w owns the row blocks w, w + 4, ...; the k loop is outermost so the
// B operand of a k step is loaded once for all of them and 8 k steps of loads are in flight.
template <int RBW>
__global__ void __launch_bounds__(LTHREADS) trinv_merge_kernel(const double* R, int64_t ldr, double* Ri, int64_t ldi) {
  constexpr int H = 64 * RBW, LDW = H + 1;
  extern __shared__ __attribute__((aligned(16))) double Wl[];     // W, column-major [16][LDW]
  __builtin_amdgcn_s_setprio(3);
  const int sidx = blockIdx.x;
  const int64_t o = (int64_t)blockIdx.y * 2 * H;
  const double* R12 = R + o + (o + H) * ldr;
  const double* Ri22 = Ri + (o + H) + (o + H + 16 * sidx) * ldi;   // the strip's 16 columns
  const double* Ri11 = Ri + o + o * ldi;
  double* Out = Ri + o + (o + H + 16 * sidx) * ldi;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lr = lane & 15, kg = lane >> 4;
  d4 acc[RBW];
#pragma unroll
  for (int q = 0; q < RBW; q++) acc[q] = (d4){0.0, 0.0, 0.0, 0.0};
  const int kmax = 16 * (sidx + 1);
  for (int kb = 0; kb < kmax; kb += 16) {          // 16 k values per trip: all loads first, then the MFMAs
    double b[4], a[4][RBW];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      b[u] = Ri22[(kb + 4 * u + kg) + (int64_t)lr * ldi];
#pragma unroll
      for (int q = 0; q < RBW; q++) a[u][q] = R12[(16 * (wid + 4 * q) + lr) + (int64_t)(kb + 4 * u + kg) * ldr];
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int q = 0; q < RBW; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][q], b[u], acc[q], 0, 0, 0);   // rows kg + 4r, column lr
  }
#pragma unroll
  for (int q = 0; q < RBW; q++) {
#pragma unroll
    for (int r = 0; r < 4; r++) Wl[16 * (wid + 4 * q) + kg + 4 * r + lr * LDW] = acc[q][r];
    acc[q] = (d4){0.0, 0.0, 0.0, 0.0};
  }
  __syncthreads();
  // row block rb only meets k >= 16 rb; all of a wave's row blocks start at the smallest of them (the extra k steps of the
  // lower ones multiply stored zeros of Ri11's lower triangle)
  for (int kb = 16 * wid; kb < H; kb += 16) {
    double b[4], a[4][RBW];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      b[u] = Wl[kb + 4 * u + kg + lr * LDW];
#pragma unroll
      for (int q = 0; q < RBW; q++) a[u][q] = Ri11[(16 * (wid + 4 * q) + lr) + (int64_t)(kb + 4 * u + kg) * ldi];
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int q = 0; q < RBW; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][q], b[u], acc[q], 0, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < RBW; q++)
#pragma unroll
    for (int r = 0; r < 4; r++) Out[16 * (wid + 4 * q) + kg + 4 * r + (int64_t)lr * ldi] = -acc[q][r];
}

// ---------------------------------------------------------------------------------------------
// Whole factor phase of one diagonal block in ONE launch (round 4): the 64-blocked right-looking sweep above costs one launch per
// step, and next to a bulk update every launch of the chain waits ~ 70 us for workgroup slots before it does 30 us of work
// (profiles/r04_experiments.log: 960 fused steps x 102 us per mixed-precision factorization of N = 65536).  Here G workgroups stay
// resident for all nblk steps and meet at a counter in global memory twice per step:
//   phase S   the block row of step i is solved in place, one workgroup per 64 x 64 block (no redundant solves as in the fused step:
//             a second meeting per step costs ~ 1 us, a solve ~ 8 us)
//   phase U   the trailing blocks (a, b) are updated round-robin; workgroup 0 owns block (i+1, i+1) and goes on - still in LDS -
//             to the leaf of step i + 1 (factor + invert) while the workers finish their share
// Nothing but G <= free workgroup slots is assumed about placement: a workgroup that arrives late only delays the others at the
// counter (the kernels it waits behind are finite), there is no cooperative-launch API involved.
// Coherence across the 8 XCDs (one L2 each): everything the workgroups exchange travels with agent-scope accesses (sc1: written
// through to / read from the memory side), so the barrier is just s_waitcnt vmcnt(0) + one counter - no L2 write-back or
// invalidate, which would also hit the operand lines of the bulk update that shares the L2s.  `fence` != 0 adds the release /
// acquire fences around the counter (A/B switch, CAP_CHAIN_FENCE).
// struct Chain64: csrc/kargs.h (shared with the CPU kernel models of tests/hipshim)

__device__ __forceinline__ double gld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gst(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The meeting point: every workgroup adds one to the counter per meeting (chain_arrive) and, if it needs what the others wrote, polls
// until all have (chain_wait).  Returns false when a workgroup never showed up (tens of seconds of polling: the launch was given more
// workgroups than the stream it runs on can hold at once) - the caller stops meeting and reports through `info`.
__device__ __forceinline__ void chain_arrive(int* ctr, int fence) {
  __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): this lane's write-through stores have reached the memory side
  __syncthreads();
  if (threadIdx.x == 0) {
    if (fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
constexpr int CHAIN_POLLS = 1 << 21;         // ~ 1.5 us per poll: a workgroup gives up after ~ 3 s (the kernels it can wait behind take ms)
__device__ __forceinline__ bool chain_wait(int* ctr, int target, int fence, int* lds_flag, int* state, int gave_up) {
  if (threadIdx.x == 0) {
    int polls = 0;
    bool ok = true;
    // the recovery launch (gave_up == 3) is the last resort - a slow, time-sliced device must not end in info = -64: it waits 8 x longer
    const int limit = gave_up == 3 ? (CHAIN_POLLS << 3) : CHAIN_POLLS;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++polls >= limit || ((polls & 63) == 0 && __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gave_up)) { ok = false; break; }
    }
    if (fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *lds_flag = ok;
  }
  __syncthreads();
  return *lds_flag != 0;
}
__device__ __forceinline__ bool chain_barrier(int* ctr, int target, int fence, int* lds_flag, int* state, int gave_up) {
  chain_arrive(ctr, fence);
  return chain_wait(ctr, target, fence, lds_flag, state, gave_up);
}
// a workgroup that gave up: mark the slot (its peers leave at their next poll) and report through info - unless a failing pivot is
// already recorded there - until the recovery launch clears it
__device__ __forceinline__ void chain_give_up(int* state, int* info, int gave_up) {
  if (threadIdx.x == 0) {
    __hip_atomic_store(state, gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (info) atomicCAS(info, 0, -64);
  }
}

// One task of the triangular-inverse assembly inside the resident chain: strip `sidx` (16 columns) of Ri12 = -Ri11 (R12 Ri22) for one
// pair of h x h diagonal blocks (h = 64 RBW) - the arithmetic of trinv_merge_kernel in the same order, operands through agent-scope
// loads (they were written by other workgroups of this launch), 4 KU k values per trip so that 4 KU (RBW + 1) loads are in flight.
template <int RBW, int KU>
__device__ __forceinline__ void chain_merge_task(const double* R12, int64_t ldr, const double* Ri11, const double* Ri22, double* Out,
                                                 int64_t ldi, int sidx, double* Wl, int t) {
  constexpr int H = 64 * RBW, LDW = H + 1;
  const int lane = t & 63, wid = t >> 6, lr = lane & 15, kg = lane >> 4;
  d4 acc[RBW];
#pragma unroll
  for (int q = 0; q < RBW; q++) acc[q] = (d4){0.0, 0.0, 0.0, 0.0};
  const int kmax = 16 * (sidx + 1);
  for (int kb = 0; kb < kmax; kb += 4 * KU) {
    double b[KU], a[KU][RBW];
#pragma unroll
    for (int u = 0; u < KU; u++) {
      if (kb + 4 * u < kmax) {
        b[u] = gld(Ri22 + (kb + 4 * u + kg) + (int64_t)lr * ldi);
#pragma unroll
        for (int q = 0; q < RBW; q++) a[u][q] = gld(R12 + (16 * (wid + 4 * q) + lr) + (int64_t)(kb + 4 * u + kg) * ldr);
      }
    }
#pragma unroll
    for (int u = 0; u < KU; u++)
      if (kb + 4 * u < kmax) {
#pragma unroll
        for (int q = 0; q < RBW; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][q], b[u], acc[q], 0, 0, 0);   // rows kg + 4r, column lr
      }
  }
#pragma unroll
  for (int q = 0; q < RBW; q++) {
#pragma unroll
    for (int r = 0; r < 4; r++) Wl[16 * (wid + 4 * q) + kg + 4 * r + lr * LDW] = acc[q][r];
    acc[q] = (d4){0.0, 0.0, 0.0, 0.0};
  }
  __syncthreads();
  for (int kb = 16 * wid; kb < H; kb += 4 * KU) {
    double b[KU], a[KU][RBW];
#pragma unroll
    for (int u = 0; u < KU; u++) {
      if (kb + 4 * u < H) {
        b[u] = Wl[kb + 4 * u + kg + lr * LDW];
#pragma unroll
        for (int q = 0; q < RBW; q++) a[u][q] = gld(Ri11 + (16 * (wid + 4 * q) + lr) + (int64_t)(kb + 4 * u + kg) * ldi);
      }
    }
#pragma unroll
    for (int u = 0; u < KU; u++)
      if (kb + 4 * u < H) {
#pragma unroll
        for (int q = 0; q < RBW; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][q], b[u], acc[q], 0, 0, 0);
      }
  }
#pragma unroll
  for (int q = 0; q < RBW; q++)
#pragma unroll
    for (int r = 0; r < 4; r++) gst(Out + 16 * (wid + 4 * q) + kg + 4 * r + (int64_t)lr * ldi, -acc[q][r]);
}

// <= 256 registers per lane (212 used): the workgroup must fit next to the waves of a running bulk update (2 x 120 registers per SIMD
// for the bf16 update, 240 for the fp64 one, of 512) - the first version took 458 and every launch waited for a CU to drain
// completely (2.7 ms per launch, profiles/r04_experiments.log).
__global__ void __launch_bounds__(LTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) chain64_coop_kernel(const Chain64 g) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  // LDS (68.5 KiB since round 5; 82.3 before): B0, B1 and a small scratch block.  The packed Dinv_i that phase S reads lives IN B1:
  // B1 (X_b of the workers; S of workgroup 0's leaf) is only used in phase U and in the leaf, the packed inverse only in phase S, and
  // a workgroup-wide barrier separates the two in both directions (chain_arrive between S and U; the end-of-step meeting, resp. the
  // barrier behind trtri_lds in workgroup 0's leaf, before the next inverse is packed).  (Tried for co-residence with a one-per-CU
  // bulk workgroup of 88 KiB: co-residence itself turned out to be the problem - see launch_tn_dma in gemm.hip - the smaller
  // footprint stayed.)
  double* B0 = lds;                       // X_a; T of the leaf
  double* B1 = lds + LMAX * LLD;          // X_b; S of the leaf
  double* Dp = B1;                        // Dinv_i, packed upper: (k, p), k <= p, at p(p+1)/2 + k  (aliases B1, see above)
  double* Sc = lds + 2 * LMAX * LLD;      // scratch: flags, the leaf's dinv / rowbuf
  static_assert(LMAX * (LMAX + 1) / 2 <= LMAX * LLD, "the packed inverse must fit into B1");
  __builtin_amdgcn_s_setprio(3);
  const int G = (int)gridDim.x, w = (int)blockIdx.x;
  const int nblk = g.nblk;
  double* const R = g.R; const int64_t ldr = g.ldr;
  int& bad = *reinterpret_cast<int*>(Sc);
  double* dinv = Sc + 2; double* rowbuf = dinv + LMAX;
  int& barrier_ok = *reinterpret_cast<int*>(Sc + 2 + LMAX + 256);             // (all LDS stays in the dynamic region)
  int epoch = 0, epoch1 = 0;             // arrivals expected at the end-of-step meetings (ctr[0]) / the mid-step ones (ctr[2])
  bool alive = true;
  long long* tr = (g.trace && w < 64 && threadIdx.x == 0) ? g.trace + (int64_t)w * 32 * 8 : nullptr;
#define CHAIN_STAMP(step, j) do { if (tr) tr[(step) * 8 + (j)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
  // (the leaf's inner stamps of workgroup 0 go to the row of workgroup 63, which a launch of <= 63 workgroups leaves free)
#define LEAF_STAMP(step, j) do { if (tr && G < 64) (tr + 63 * 32 * 8)[(step) * 8 + (j)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)

  // ---- prologue: the state word of the slot, then save (primary launch) or restore (recovery launch) the block
  const int GU = g.recover ? 3 : 1;        // what a workgroup of this launch writes into the state word when it gives up
  bool inject = false;
  {
    int* sflag = reinterpret_cast<int*>(Sc) + 1;
    if (threadIdx.x == 0) {
      const int st = __hip_atomic_load(g.ctr + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int inj = 0;
      if (!g.recover && g.fallbacks) inj = __hip_atomic_load(g.fallbacks + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0;
      *sflag = st | (inj << 8);
    }
    __syncthreads();
    const int sv = *sflag;
    __syncthreads();
    inject = (sv >> 8) != 0;
    const int st = sv & 255;
    // primary launch that finds 1: a peer has already given up (I was dispatched late) - nothing to do but be counted out.
    // recovery launch: 0 = the primary launch went through - return (the slot's words are untouched); 1 / 2 = it gave up - take
    // over (2: my peer has already started); 3 = my peer has given up in the recovery itself - be counted out.
    if (g.recover && st == 0) return;
    if (g.recover && st == 1 && threadIdx.x == 0) atomicCAS(g.ctr + 3, 1, 2);
    if ((!g.recover && st == 1) || (g.recover && st == 3)) alive = false;
    // A workgroup of the PRIMARY launch that is dispatched after a peer has given up (st == 1) still saves its share: nobody got past
    // the first meeting without it, so its blocks are untouched, and the recovery launch restores EVERY block from the backup - a share
    // that was never saved would come back as whatever the buffer held before (round 5: seen as soon as the backup buffer was a fresh
    // allocation per plan instead of one kept per stream handle; with the test hook the peers give up within microseconds).
    const bool save_late = !g.recover && st == 1;
    if ((alive || save_late) && g.backup) {
      // block q of the packed upper enumeration (column by column of blocks): workgroup 0 takes block (0, 0) - the only one that is
      // modified before the first meeting (by its own leaf) - the workers share the rest; the meeting orders save / restore before use
      const int nq = nblk * (nblk + 1) / 2;
      for (int q = (w == 0) ? 0 : w; q < nq; q += (w == 0) ? nq : (G - 1)) {
        int a = q, b = 0;
        while (a >= b + 1) { a -= b + 1; b++; }
        double* blk = R + (int64_t)a * 64 + (int64_t)b * 64 * ldr;
        double* sav = g.backup + (int64_t)q * 64 * 64;
        for (int e = threadIdx.x; e < 64 * 64; e += LTHREADS) {
          if (g.recover) gst(blk + (e & 63) + (int64_t)(e >> 6) * ldr, sav[e]);
          else sav[e] = gld(blk + (e & 63) + (int64_t)(e >> 6) * ldr);
        }
      }
      if (g.recover && w == 0 && threadIdx.x == 0) {
        if (g.info) atomicCAS(g.info, -64, 0);
        if (g.fallbacks) atomicAdd(g.fallbacks, 1);
      }
    }
    if (inject && w == 0 && threadIdx.x == 0) atomicSub(g.fallbacks + 1, 1);
  }
  // step -1 = the leaf of block 0 alone
  for (int i = -1; alive && i + 1 < nblk; i++) {
    const int r = nblk - 1 - i;
    // the lane's indices are recomputed from an opaque copy of threadIdx.x every step: hoisted out of this (long) loop they cost the
    // kernel ~ 100 registers of loop-invariant offsets and masks, i.e. scratch spills under the 168-register cap
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int lane = t & 63, wid = t >> 6, lr = lane & 15, kg = lane >> 4;
    CHAIN_STAMP(i + 1, 0);
    if (i >= 0) {
      // ---- phase S: the block row of step i, X_a = Dinv_i^T A_ia, in place (A_ia has no other reader).  Workgroup 0 owns a = i + 1
      // (Dinv_i is still in its LDS from the leaf), the workers share the rest.
      const int nmine = (w == 0) ? 1 : (r - 1 > w - 1 ? (r - 1 - (w - 1) + (G - 2)) / (G - 1) : 0);
      if (w > 0 && nmine > 0) {
        const double* Dinv = g.Ri + (int64_t)i * 64 * (g.ldi + 1);
        __syncthreads();
        for (int e0 = 0; e0 < 64 * 64; e0 += 4 * LTHREADS) {
          double v[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int e = e0 + u * LTHREADS + t, ii = e & 63, jj = e >> 6;
            v[u] = (ii <= jj) ? gld(Dinv + ii + (int64_t)jj * g.ldi) : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int e = e0 + u * LTHREADS + t, ii = e & 63, jj = e >> 6;
            if (ii <= jj) Dp[jj * (jj + 1) / 2 + ii] = v[u];
          }
        }
      }
      // workgroup 0 has its trailing block (i+1, i+1) on the way while it solves
      double cin[4][4];
      if (w == 0) {
        const double* C = R + (int64_t)(i + 1) * 64 * (ldr + 1);
#pragma unroll
        for (int sb = 0; sb < 4; sb++) {
          const int id = wid + 4 * sb, bi = id & 3, bj = id >> 2;
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int row = bi * 16 + lr, col = bj * 16 + kg + 4 * rr;
            cin[sb][rr] = (row <= col) ? gld(C + row + (int64_t)col * ldr) : 0.0;
          }
        }
      }
      for (int m = 0; m < nmine; m++) {
        const int a = (w == 0) ? i + 1 : i + 2 + (w - 1) + m * (G - 1);
        double* Aia = R + (int64_t)i * 64 + (int64_t)a * 64 * ldr;
        __syncthreads();
        for (int e0 = 0; e0 < 64 * 64; e0 += 4 * LTHREADS) {
          double v[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { const int e = e0 + u * LTHREADS + t; v[u] = gld(Aia + (e & 63) + (int64_t)(e >> 6) * ldr); }
#pragma unroll
          for (int u = 0; u < 4; u++) { const int e = e0 + u * LTHREADS + t; SM(B0, e & 63, e >> 6) = v[u]; }
        }
        __syncthreads();
        // wave `wid` owns the 16 x 16 blocks (bi, bj) = ((wid + s) & 3, s): every wave meets each K range 16 (bi + 1) once
        d4 acc[4];
#pragma unroll
        for (int sb = 0; sb < 4; sb++) {
          const int bi = (wid + sb) & 3, bj = sb;
          d4 c = {0.0, 0.0, 0.0, 0.0};
          const int pcol = bi * 16 + lr;
          const double* dcol = Dp + pcol * (pcol + 1) / 2;
          for (int k0 = 0; k0 < 16 * (bi + 1); k0 += 16) {
            double av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int k = k0 + 4 * u + kg;
              av[u] = (k <= pcol) ? dcol[k] : 0.0;                       // Dinv[k][pcol]
              bv[u] = SM(B0, k, bj * 16 + lr);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], c, 0, 0, 0);    // rows kg + 4r, cols lr
          }
          acc[sb] = c;
        }
        __syncthreads();                 // every read of the unsolved block is done
#pragma unroll
        for (int sb = 0; sb < 4; sb++) {
          const int bi = (wid + sb) & 3, bj = sb;
#pragma unroll
          for (int rr = 0; rr < 4; rr++) SM(B0, bi * 16 + kg + 4 * rr, bj * 16 + lr) = acc[sb][rr];
        }
        __syncthreads();
        for (int e = t; e < 64 * 64; e += LTHREADS) gst(Aia + (e & 63) + (int64_t)(e >> 6) * ldr, SM(B0, e & 63, e >> 6));
      }
      CHAIN_STAMP(i + 1, 1);
      epoch1 += G;
      chain_arrive(g.ctr + 2, g.fence);
      // (workgroup 0 only needs its own X_{i+1} for what follows: it says that it is there and goes on.  This meeting counts on its
      // OWN word: on a shared counter workgroup 0's early arrival at the end-of-step meeting would stand in for a worker that has
      // not stored its solved block yet - seen as wrong factors when eight processes time-slice one GPU)
      if (w > 0 && !chain_wait(g.ctr + 2, epoch1, g.fence, &barrier_ok, g.ctr + 3, GU)) { chain_give_up(g.ctr + 3, g.info, GU); alive = false; break; }
      CHAIN_STAMP(i + 1, 2);
      // ---- phase U: C_ab -= X_a^T X_b on the trailing blocks (upper part on the diagonal ones).  Workgroup 0: block (i+1, i+1), kept
      // in LDS for the leaf; workers: pairs w, w + G - 1, ... of the triangular enumeration.
      const int npairs = r * (r + 1) / 2;
      const int cnt = (w == 0) ? 1 : (npairs > w ? (npairs - w + (G - 2)) / (G - 1) : 0);
      for (int m = 0; m < cnt; m++) {
        int q = (w == 0) ? 0 : w + m * (G - 1), bjt = 0;
        while (q >= bjt + 1) { q -= bjt + 1; bjt++; }
        const int a = i + 1 + q, b = i + 1 + bjt;
        const bool diag = (a == b);
        const double* Xa = R + (int64_t)i * 64 + (int64_t)a * 64 * ldr;
        const double* Xb = R + (int64_t)i * 64 + (int64_t)b * 64 * ldr;
        double* C = R + (int64_t)a * 64 + (int64_t)b * 64 * ldr;
        if (w > 0) {
#pragma unroll
          for (int sb = 0; sb < 4; sb++) {
            const int id = wid + 4 * sb, bi = id & 3, bj = id >> 2;
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
              const int row = bi * 16 + lr, col = bj * 16 + kg + 4 * rr;
              cin[sb][rr] = (!diag || row <= col) ? gld(C + row + (int64_t)col * ldr) : 0.0;
            }
          }
        }
        if (w > 0) {                                                  // (workgroup 0 still holds X_{i+1} in B0)
          __syncthreads();
          for (int e0 = 0; e0 < 64 * 64; e0 += 4 * LTHREADS) {
            double v1[4], v2[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int e = e0 + u * LTHREADS + t, ii = e & 63, jj = e >> 6;
              v1[u] = gld(Xa + ii + (int64_t)jj * ldr);
              v2[u] = diag ? 0.0 : gld(Xb + ii + (int64_t)jj * ldr);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int e = e0 + u * LTHREADS + t, ii = e & 63, jj = e >> 6;
              SM(B0, ii, jj) = v1[u];
              if (!diag) SM(B1, ii, jj) = v2[u];
            }
          }
        } else {
          for (int e = t; e < 64 * LLD; e += LTHREADS) B1[e] = 0.0;    // becomes S of the leaf
        }
        __syncthreads();
        const double* XB = diag ? B0 : B1;
#pragma unroll
        for (int sb = 0; sb < 4; sb++) {
          const int id = wid + 4 * sb, bi = id & 3, bj = id >> 2;     // bi: row block of C (from X_a), bj: column block (from X_b)
          if (diag && bi > bj) continue;
          d4 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
          for (int k0 = 0; k0 < 64; k0 += 4) {
            const double colv = SM(XB, k0 + kg, bj * 16 + lr);
            const double rowv = SM(B0, k0 + kg, bi * 16 + lr);
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(colv, rowv, c, 0, 0, 0);   // result: (col = kg + 4r, row = lr)
          }
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int row = bi * 16 + lr, col = bj * 16 + kg + 4 * rr;
            if (!diag || row <= col) {
              if (w == 0) SM(B1, row, col) = cin[sb][rr] - c[rr];
              else gst(C + row + (int64_t)col * ldr, cin[sb][rr] - c[rr]);
            }
          }
        }
      }
      CHAIN_STAMP(i + 1, 3);
    }
    if (w == 0) {
      // ---- leaf of block i + 1: S = B1 (upper part, rest zero), T = B0 -> R_bb in place, Dinv_b with a zero-filled lower part
      const int bk = i + 1;
      double* C = R + (int64_t)bk * 64 * (ldr + 1);
      double* Dn = g.Ri + (int64_t)bk * 64 * (g.ldi + 1);
      __syncthreads();                                               // every read of B0 (X_a) is done, S = B1 is complete
      if (i < 0) {
        for (int e = t; e < 64 * LLD; e += LTHREADS) B1[e] = 0.0;
        __syncthreads();
        for (int e0 = 0; e0 < 64 * 64; e0 += 4 * LTHREADS) {
          double v[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int e = e0 + u * LTHREADS + t, ii = e & 63, jj = e >> 6;
            v[u] = (ii <= jj) ? gld(C + ii + (int64_t)jj * ldr) : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 4; u++) { const int e = e0 + u * LTHREADS + t; SM(B1, e & 63, e >> 6) = v[u]; }
        }
      }
      for (int e = t; e < 64 * LLD; e += LTHREADS) B0[e] = 0.0;
      if (t == 0) bad = 0;
      __syncthreads();
      LEAF_STAMP(i + 1, 0);
      potrf_lds(B1, dinv, rowbuf, 64, &bad, t, (tr && G < 62) ? tr + 62 * 32 * 8 + ((i + 1) & 15) * 16 : nullptr);
      LEAF_STAMP(i + 1, 1);
      for (int e = t; e < 64 * 64; e += LTHREADS) {
        const int ii = e & 63, jj = e >> 6;
        if (ii <= jj) gst(C + ii + (int64_t)jj * ldr, SM(B1, ii, jj));
      }
      LEAF_STAMP(i + 1, 2);
      trtri_lds(B1, B0, dinv, 64, t);
      __syncthreads();
      LEAF_STAMP(i + 1, 3);
      if (t == 0 && bad != 0 && g.info) atomicCAS(g.info, 0, g.info_base + bk * 64 + bad);
      __syncthreads();                                               // (`bad` lives in the region Dp is about to reuse)
      for (int e = t; e < 64 * 64; e += LTHREADS) {
        const int ii = e & 63, jj = e >> 6;
        const double v = (ii <= jj) ? SM(B0, ii, jj) : 0.0;
        gst(Dn + ii + (int64_t)jj * g.ldi, v);
        if (ii <= jj) Dp[jj * (jj + 1) / 2 + ii] = v;                // Dinv_{i+1} stays here for workgroup 0's solve of the next step
      }
    }
    CHAIN_STAMP(i + 1, 4);
    epoch += G;
    if (inject && i < 0) { chain_give_up(g.ctr + 3, g.info, GU); alive = false; break; }     // (test hook: as if a peer never arrived)
    if (!chain_barrier(g.ctr, epoch, g.fence, &barrier_ok, g.ctr + 3, GU)) { chain_give_up(g.ctr + 3, g.info, GU); alive = false; break; }
    CHAIN_STAMP(i + 1, 5);
  }
  // ---- the inverse, level by level: pairs of h x h diagonal blocks, one task per 16-column strip (cf. trinv_merge_kernel)
  CHAIN_STAMP(nblk, 0);
  for (int h = 64, lvl = 1; alive && h <= g.hmax && 2 * h <= 64 * nblk; h *= 2, lvl++) {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int strips = h / 16, ntask = (64 * nblk / (2 * h)) * strips;
    for (int task = w; task < ntask; task += G) {
      const int z = task / strips, sidx = task - z * strips;
      const int64_t o = (int64_t)z * 2 * h;
      const double* R12 = R + o + (o + h) * ldr;
      const double* Ri22 = g.Ri + (o + h) + (o + h + 16 * sidx) * g.ldi;
      const double* Ri11 = g.Ri + o + o * g.ldi;
      double* Out = g.Ri + o + (o + h + 16 * sidx) * g.ldi;
      __syncthreads();                                               // the strip buffer of the previous task is free
      if (h == 64) chain_merge_task<1, 8>(R12, ldr, Ri11, Ri22, Out, g.ldi, sidx, B0, t);
      else if (h == 128) chain_merge_task<2, 8>(R12, ldr, Ri11, Ri22, Out, g.ldi, sidx, B0, t);
      else chain_merge_task<4, 8>(R12, ldr, Ri11, Ri22, Out, g.ldi, sidx, B0, t);
    }
    CHAIN_STAMP(nblk, lvl);
    epoch += G;
    if (2 * h <= g.hmax && 4 * h <= 64 * nblk) {                     // (nothing follows the last level)
      if (!chain_barrier(g.ctr, epoch, g.fence, &barrier_ok, g.ctr + 3, GU)) { chain_give_up(g.ctr + 3, g.info, GU); alive = false; }
    }
  }
#undef CHAIN_STAMP
#undef LEAF_STAMP
  // the last workgroup through resets the counters for the next launch that is handed this slot
  if (threadIdx.x == 0) {
    const int done = __hip_atomic_fetch_add(g.ctr + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == G - 1) {
      __hip_atomic_store(g.ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(g.ctr + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(g.ctr + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (a recovery launch ends the episode whatever happened in it - a second give-up leaves info = -64 for the caller;
      //  a primary launch without a backup has no recovery launch behind it and clears its own mark)
      if (g.recover || !g.backup) __hip_atomic_store(g.ctr + 3, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

// Factor phase of blocked_cholinv (cholinv.hip) in one launch of `wgs` resident workgroups (chain64_coop_kernel): R (n = 64 nblk)
// factored in place, the 64 x 64 diagonal blocks of Ri = their inverses, the off-diagonal blocks up to pairs of hmax x hmax.
// ctr: four ints (end-of-step meetings, exit count, mid-step meetings, spare), zero before the first use (the kernel leaves them zero).
constexpr size_t CHAIN_LDS_BYTES = (2 * LMAX * LLD + 2 + LMAX + 256 + 2) * sizeof(double);      // B0, B1 (+ the packed inverse inside it), scratch

// Workgroups of chain64_coop_kernel the current device can hold at once (occupancy of the kernel x compute units), cached per device.
// A launch with more workgroups than that could never have them all resident: its meetings would only end by the give-up path.
int cap_chain64_coop_max_resident() {
  static int cached[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 32;
  if (cached[dev] > 0) return cached[dev];
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(chain64_coop_kernel), LTHREADS, CHAIN_LDS_BYTES) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu <= 0 || cus <= 0) {
    (void)hipGetLastError();
    return 32;
  }
  cached[dev] = per_cu * cus;
  return cached[dev];
}

int cap_chain64_coop(double* R, int64_t ldr, double* Ri, int64_t ldi, int nblk, int* info, int info_base, int* ctr, int wgs, int fence,
                     int hmax, hipStream_t stream, long long* trace, double* backup, int* fallbacks) {
  if (nblk <= 0) return CAP_OK;
  if (nblk > 31) trace = nullptr;
  if (wgs < 2 || hmax < 0 || hmax > 256 || (hmax & (hmax - 1))) return CAP_ERR_ARG;
  const int useful = std::max(1, (nblk - 1) * nblk / 2);           // one workgroup per trailing block of the first step
  wgs = std::max(2, std::min(wgs, useful));
  Chain64 g{R, ldr, Ri, ldi, nblk, info, info_base, ctr, fence, hmax, trace, backup, fallbacks, 0};
  // access notes: R's upper triangle in place, the square of Ri (diagonal 64-blocks are written whole), the stream's own counter words
  // and backup, the device-wide fall-back counters and the pivot report (atomics)
  auto note = [&]() {
    if (!cap_acc_on()) return;
    const int64_t n = (int64_t)nblk * 64;
    cap_acc_rw(R, ldr, n, n, 1); cap_acc_rw(Ri, ldi, n, n);
    cap_acc_rw(ctr, 4, 4, 1, 0, 4);
    if (backup) cap_acc_rw(backup, 0, (int64_t)nblk * (nblk + 1) / 2 * 64 * 64, 1);
    if (fallbacks) cap_acc_atomic(fallbacks, 4, 4);
    if (info) cap_acc_atomic(info, 1, 4);
  };
  note();
  hipLaunchKernelGGL(chain64_coop_kernel, dim3((unsigned)wgs), dim3(LTHREADS), CHAIN_LDS_BYTES, stream, g);
  CAP_HIP(hipGetLastError());
  if (backup) {
    // the recovery launch: two workgroups that return at once unless a workgroup of the launch above gave up (ctr[3] == 1)
    g.recover = 1; g.trace = nullptr;
    note();
    hipLaunchKernelGGL(chain64_coop_kernel, dim3(2), dim3(LTHREADS), CHAIN_LDS_BYTES, stream, g);
    CAP_HIP(hipGetLastError());
  }
  return CAP_OK;
}

// Ri12 = -Ri11 (R12 Ri22) for npairs aligned pairs of h x h diagonal blocks (h = 64, 128 or 256), one launch
int cap_trinv_merge(const double* R, int64_t ldr, double* Ri, int64_t ldi, int64_t h, int npairs, hipStream_t stream) {
  if (npairs <= 0) return CAP_OK;
  const dim3 grid((unsigned)(h / 16), (unsigned)npairs);
  const size_t lds_bytes = (size_t)16 * (h + 1) * sizeof(double);
  if (cap_acc_on() && (h == 64 || h == 128 || h == 256))
    for (int64_t z = 0; z < npairs; z++) {
      const int64_t o = z * 2 * h;
      cap_acc_r(R + o + (o + h) * ldr, ldr, h, h); cap_acc_r(Ri + o + o * ldi, ldi, h, h, 1); cap_acc_r(Ri + (o + h) + (o + h) * ldi, ldi, h, h, 1);
      cap_acc_w(Ri + o + (o + h) * ldi, ldi, h, h);
    }
  if (h == 64) hipLaunchKernelGGL(trinv_merge_kernel<1>, grid, dim3(LTHREADS), lds_bytes, stream, R, ldr, Ri, ldi);
  else if (h == 128) hipLaunchKernelGGL(trinv_merge_kernel<2>, grid, dim3(LTHREADS), lds_bytes, stream, R, ldr, Ri, ldi);
  else if (h == 256) hipLaunchKernelGGL(trinv_merge_kernel<4>, grid, dim3(LTHREADS), lds_bytes, stream, R, ldr, Ri, ldi);
  else return CAP_ERR_UNSUPPORTED;
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

int cap_panel64_solve_update(double* R, int64_t ldr, const double* Dinv, int64_t ldi, int i, int nblk, double* Xs,
                             hipStream_t stream, double* Dnext, int64_t ldn, int* info, int info_base, const double* cj_src,
                             double* cj_dst, int64_t cj_ld, int cj_cols, int direct) {
  const int r = nblk - 1 - i;
  if (r <= 0) return CAP_OK;
  const size_t lds_bytes = (2 * LMAX * LLD + LMAX * (LMAX + 1) / 2) * sizeof(double);
  static_assert(LMAX * (LMAX + 1) / 2 >= 2 + LMAX + 256, "the folded leaf keeps bad / dinv / rowbuf in the packed-Dinv region");
  const Panel64Fold f{Dnext, ldn, info, info_base, cj_src, cj_dst, cj_ld, cj_cols, direct};
  if (cap_acc_on()) {
    const int64_t o = (int64_t)(i + 1) * 64, w = (int64_t)r * 64;
    cap_acc_r(Dinv, ldi, 64, 64, 1);
    cap_acc(direct ? CAP_ACC_RW : CAP_ACC_R, R + (int64_t)i * 64 + o * ldr, ldr, 64, w);      // block row i right of the diagonal block
    cap_acc_rw(R + o + o * ldr, ldr, w, w, 1);                                                  // trailing blocks (a, b), i < a <= b
    if (!direct) cap_acc_w(Xs, 0, w * 64, 1);
    if (Dnext) { cap_acc_w(Dnext, ldn, 64, 64); if (info) cap_acc_atomic(info, 1, 4); }
    if (cj_cols > 0) { cap_acc_r(cj_src, 0, (int64_t)64 * cj_cols, 1); cap_acc_w(cj_dst, cj_ld, 64, cj_cols); }
  }
  hipLaunchKernelGGL(panel64_solve_update_kernel, dim3(r * (r + 1) / 2), dim3(LTHREADS), lds_bytes, stream, R, ldr, Dinv, ldi, i, nblk,
                     Xs, f);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

constexpr size_t LEAF_LDS_BYTES = (2 * LMAX * LLD + 2 + LMAX + 256) * sizeof(double);

int cap_leaf_cholinv(double* A, int64_t lda, double* Rinv, int64_t ldr, int n, int zero_lower, int* info,
                     int info_base, hipStream_t stream, const double* cjob_src, double* cjob_dst, int64_t cjob_ld,
                     int cjob_cols) {
  if (n <= 0) return CAP_OK;
  if (n > LMAX) return CAP_ERR_ARG;
  if (cap_acc_on()) {
    cap_acc_rw(A, lda, n, n, 1);
    if (Rinv) cap_acc_w(Rinv, ldr, n, n, zero_lower ? 0 : 1);
    if (info) cap_acc_atomic(info, 1, 4);
    if (cjob_cols > 0) { cap_acc_r(cjob_src, 0, (int64_t)64 * cjob_cols, 1); cap_acc_w(cjob_dst, cjob_ld, 64, cjob_cols); }
  }
  hipLaunchKernelGGL(leaf_cholinv_kernel, dim3(1), dim3(LTHREADS), LEAF_LDS_BYTES, stream, A, lda, Rinv, ldr, n,
                     zero_lower, info, info_base, cjob_src, cjob_dst, cjob_ld, cjob_cols);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

int cap_leaf_trtri(const double* R, int64_t ldr, double* Rinv, int64_t ldi, int n, hipStream_t stream) {
  if (n <= 0) return CAP_OK;
  if (n > LMAX) return CAP_ERR_ARG;
  cap_acc_r(R, ldr, n, n, 1); cap_acc_w(Rinv, ldi, n, n, 1);
  hipLaunchKernelGGL(leaf_trtri_kernel, dim3(1), dim3(LTHREADS), LEAF_LDS_BYTES, stream, R, ldr, Rinv, ldi, n);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}
