/* A plain CBLAS / LAPACKE program - nothing in it knows about this library: host arrays, cblas_dgemm, LAPACKE_dpotrf, LAPACKE_dtrtri,
 * cblas_dtrmm, cblas_dsyrk with the argument lists of MKL's mkl.h (the reference's seam: blas/interface.hpp:54,74,92,
 * lapack/interface.hpp:39,54).  Linked with -lcapital_amd_cblas instead of a CPU BLAS, every call below runs on the MI355X
 * (include/capital_amd_cblas.h).
 *   gcc -std=c99 examples/cblas_offload_demo.c -Iinclude/for_upstream -Lcapital_amd/lib -lcapital_amd_cblas -lm -o demo && LD_LIBRARY_PATH=capital_amd/lib ./demo 1500
 * Prints: n, the Cholesky residual ||R^T R - A||_F / ||A||_F, ||R^-1 R - I||_F / sqrt(n), the calls served.               */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "mkl.h"                                   /* any mkl.h / cblas.h + lapacke.h; here include/for_upstream/mkl.h */

void capcb_counters(long long* calls, long long* bytes_in, long long* bytes_out);   /* the one line that knows: how much was staged */

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1000, k = n / 2 + 3, ld = n + 1;       /* a leading dimension that is not the window */
  double* G = malloc(sizeof(double) * (size_t)k * n);                              /* k x n */
  double* A = malloc(sizeof(double) * (size_t)ld * n);                             /* A = G^T G + n I, upper triangle (DSYRK) */
  double* R = malloc(sizeof(double) * (size_t)ld * n);
  double* W = malloc(sizeof(double) * (size_t)ld * n);
  double* Ri = malloc(sizeof(double) * (size_t)ld * n);
  unsigned s = 12345u;
  for (size_t i = 0; i < (size_t)k * n; i++) { s = s * 1664525u + 1013904223u; G[i] = (double)(s >> 8) / 16777216.0 - 0.5; }
  for (int j = 0; j < n; j++) for (int i = 0; i < ld; i++) A[(size_t)j * ld + i] = (i == j) ? (double)n : (i < j || i >= n ? 0.0 : -777.0);   /* junk below the diagonal */
  cblas_dsyrk(CblasColMajor, CblasUpper, CblasTrans, n, k, 1.0, G, k, 1.0, A, ld);
  for (size_t i = 0; i < (size_t)ld * n; i++) R[i] = A[i];
  const int info = LAPACKE_dpotrf(LAPACK_COL_MAJOR, 'U', n, R, ld);
  if (info) { printf("LAPACKE_dpotrf info = %d\n", info); return 1; }
  /* W = R^T R by DTRMM on a copy of (the upper triangle of) R: W <- R, then W <- R^T W */
  for (int j = 0; j < n; j++) for (int i = 0; i < ld; i++) W[(size_t)j * ld + i] = (i <= j && i < n) ? R[(size_t)j * ld + i] : 0.0;
  cblas_dtrmm(CblasColMajor, CblasLeft, CblasUpper, CblasTrans, CblasNonUnit, n, n, 1.0, R, ld, W, ld);
  double num = 0, den = 0;
  for (int j = 0; j < n; j++) for (int i = 0; i <= j; i++) { const double a = A[(size_t)j * ld + i], d = W[(size_t)j * ld + i] - a; num += d * d; den += a * a; }
  for (int j = 0; j < n; j++) for (int i = j + 1; i < ld; i++) if (R[(size_t)j * ld + i] != (i < n ? -777.0 : 0.0)) { printf("memory outside the upper triangle was touched\n"); return 1; }
  /* R^-1 by DTRTRI, R^-1 R by DGEMM */
  for (size_t i = 0; i < (size_t)ld * n; i++) Ri[i] = R[i];
  if (LAPACKE_dtrtri(LAPACK_COL_MAJOR, 'U', 'N', n, Ri, ld)) { printf("LAPACKE_dtrtri failed\n"); return 1; }
  for (int j = 0; j < n; j++) for (int i = j + 1; i < n; i++) { Ri[(size_t)j * ld + i] = 0.0; R[(size_t)j * ld + i] = 0.0; }
  cblas_dgemm(CblasColMajor, CblasNoTrans, CblasNoTrans, n, n, n, 1.0, Ri, ld, R, ld, 0.0, W, ld);
  double inv = 0;
  for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) { const double d = W[(size_t)j * ld + i] - (i == j ? 1.0 : 0.0); inv += d * d; }
  long long calls = 0, in = 0, out = 0;
  capcb_counters(&calls, &in, &out);
  printf("n = %d residual = %.3e inverse = %.3e calls = %lld staged_MB = %.1f\n", n, sqrt(num / den), sqrt(inv / n), calls, (double)(in + out) / 1048576.0);
  free(G); free(A); free(R); free(W); free(Ri);
  return 0;
}
